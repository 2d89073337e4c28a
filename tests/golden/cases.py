"""Golden-fixture case table shared by make_golden.py (generator, reference side) and the tests (checker side).

Each case: name -> dict(cfg=<make_config kwargs>, n=<samples>, regime=<synth regime>, wseed=<weight seed>,
                        rseed=<rotation seed>, direction='forward'|'inverse', fisher=<synth.fisher_A kind or None>)
Inputs and weights are regenerated from rotationnormflow_amd.synth on both sides; fixtures hold outputs only.
"""
SYMSOL = dict(condition=1, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)

CASES = {
    # BASELINE configs[0]: 8-layer MobiusAffine, unconditional
    "c1_default":      dict(cfg=dict(layers=8), n=2048, regime="default", wseed=1, rseed=42, direction="forward", fisher=None),
    "c1_trained":      dict(cfg=dict(layers=8), n=2048, regime="trained", wseed=2, rseed=43, direction="forward", fisher=None),
    "c1_trained_inv":  dict(cfg=dict(layers=8), n=1024, regime="trained", wseed=2, rseed=44, direction="inverse", fisher=None),
    # configs[1]/[2]: 24-layer MobiusAffine + matrix-Fisher base
    "c2_default":      dict(cfg=dict(layers=24), n=2048, regime="default", wseed=3, rseed=45, direction="forward", fisher="diag531"),
    "c2_trained":      dict(cfg=dict(layers=24), n=2048, regime="trained", wseed=4, rseed=46, direction="forward", fisher="tilted"),
    # configs[3]: conditional SYMSOL-I structure, F=256
    "c4_default":      dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n=1024, regime="default", wseed=5, rseed=47, direction="forward", fisher=None),
    "c4_trained":      dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n=1024, regime="trained", wseed=6, rseed=48, direction="forward", fisher=None),
    "c4_trained_inv":  dict(cfg=dict(layers=6, feature_dim=256, **SYMSOL), n=512, regime="trained", wseed=6, rseed=49, direction="inverse", fisher=None),
    # configs[4]: 42-layer Moebius-only inverse (conditional F=512, and its unconditional variant)
    "c5_default_inv":  dict(cfg=dict(layers=42, condition=1, feature_dim=512, rot="None", frequent_permute=1, last_affine=0, first_affine=0),
                            n=512, regime="default", wseed=7, rseed=50, direction="inverse", fisher="diag531"),
    "c5u_trained_inv": dict(cfg=dict(layers=42, condition=0, rot="None", last_affine=0, first_affine=0),
                            n=1024, regime="trained", wseed=8, rseed=51, direction="inverse", fisher=None),
    "c5u_trained_fwd": dict(cfg=dict(layers=42, condition=0, rot="None", last_affine=0, first_affine=0),
                            n=1024, regime="trained", wseed=8, rseed=52, direction="forward", fisher=None),
    # registry / schedule edge cases
    "cond16_regular":  dict(cfg=dict(layers=4, condition=1, feature_dim=None, rot="16Trans"), n=1024, regime="trained", wseed=9, rseed=53, direction="forward", fisher=None),
    "cond16_reg_inv":  dict(cfg=dict(layers=3, condition=1, feature_dim=40, rot="16Trans", frequent_permute=1), n=512, regime="trained", wseed=10, rseed=54, direction="inverse", fisher=None),
    "k16_freqperm":    dict(cfg=dict(layers=5, segments=16, frequent_permute=1), n=1024, regime="trained", wseed=11, rseed=55, direction="forward", fisher=None),
    "k8_nofirst":      dict(cfg=dict(layers=7, segments=8, first_affine=0), n=1024, regime="trained", wseed=12, rseed=56, direction="forward", fisher=None),
    "k128_inv":        dict(cfg=dict(layers=3, segments=128), n=512, regime="trained", wseed=13, rseed=57, direction="inverse", fisher=None),
    # segment counts that are not multiples of 8 (flow/mobiusflow.py:7-14 takes any `segments`): the kernels pad the last fc_last tile
    "k20_fwd":         dict(cfg=dict(layers=4, segments=20), n=1024, regime="trained", wseed=31, rseed=91, direction="forward", fisher=None),
    "k20_inv":         dict(cfg=dict(layers=4, segments=20), n=512, regime="trained", wseed=31, rseed=92, direction="inverse", fisher=None),
    "k75_cond_inv":    dict(cfg=dict(layers=2, segments=75, condition=1, feature_dim=24, rot="None", frequent_permute=1, last_affine=0, first_affine=0),
                            n=512, regime="trained", wseed=32, rseed=93, direction="inverse", fisher=None),
    # unconditional LU parameterisation of the 4x4 affine, and the SVD-rotation layer (registry rows a15 / a18)
    "lu16_uncond":     dict(cfg=dict(layers=3, rot="16Trans", lu=1), n=1024, regime="trained", wseed=15, rseed=59, direction="forward", fisher=None),
    "lu16_uncond_inv": dict(cfg=dict(layers=3, rot="16Trans", lu=1), n=512, regime="trained", wseed=15, rseed=60, direction="inverse", fisher=None),
    "rot16_uncond":    dict(cfg=dict(layers=3, rot="16Rot"), n=1024, regime="trained", wseed=16, rseed=61, direction="forward", fisher=None),
    "rot16_uncond_inv": dict(cfg=dict(layers=3, rot="16Rot"), n=512, regime="trained", wseed=16, rseed=62, direction="inverse", fisher=None),
    "lu16un_cond":     dict(cfg=dict(layers=3, condition=1, feature_dim=16, rot="16UnTrans", lu=1), n=512, regime="trained", wseed=17, rseed=63,
                            direction="forward", fisher=None),
    "rot16un_cond":    dict(cfg=dict(layers=3, condition=1, feature_dim=16, rot="16UnRot"), n=512, regime="trained", wseed=18, rseed=64,
                            direction="forward", fisher=None),
    # 3x3 / 6x6 ablation layers, unconditional (README.md:151-155; flow/squeezetrans.py:176-361, flow/rottrans.py:72-165)
    "gs9_uncond":      dict(cfg=dict(layers=3, rot="9TransLSmith"), n=1024, regime="trained", wseed=19, rseed=65, direction="forward", fisher=None),
    "gs9_uncond_inv":  dict(cfg=dict(layers=3, rot="9TransLSmith"), n=512, regime="trained", wseed=19, rseed=66, direction="inverse", fisher=None),
    "gs9lu_uncond":    dict(cfg=dict(layers=3, rot="9TransLSmith", lu=1), n=512, regime="trained", wseed=20, rseed=67, direction="forward", fisher=None),
    "gs36_uncond":     dict(cfg=dict(layers=3, rot="36Trans"), n=1024, regime="trained", wseed=21, rseed=68, direction="forward", fisher=None),
    "gs36_uncond_inv": dict(cfg=dict(layers=3, rot="36Trans"), n=512, regime="trained", wseed=21, rseed=69, direction="inverse", fisher=None),
    "svdl9_uncond":    dict(cfg=dict(layers=3, rot="9TransLSVD"), n=512, regime="trained", wseed=22, rseed=70, direction="forward", fisher=None),
    "svdl9_uncond_inv": dict(cfg=dict(layers=3, rot="9TransLSVD"), n=512, regime="trained", wseed=22, rseed=71, direction="inverse", fisher=None),
    "svdr9_uncond":    dict(cfg=dict(layers=3, rot="9TransRSVD"), n=512, regime="trained", wseed=23, rseed=72, direction="forward", fisher=None),
    "smithr9_uncond":  dict(cfg=dict(layers=3, rot="9TransRSmith"), n=512, regime="trained", wseed=24, rseed=73, direction="forward", fisher=None),
    "smithr9_uncond_inv": dict(cfg=dict(layers=3, rot="9TransRSmith"), n=512, regime="trained", wseed=24, rseed=74, direction="inverse", fisher=None),
    # README.md:153-154 ablations on SYMSOL: affine only (--dist noflow), and unconditional 4x4 with --lu 1 behind a conditional first layer is
    # not constructible here (Condition16TransLU is batch-coupled in the reference: torch.diag of a [N,4] tensor, squeezetrans.py:127)
    "noflow_affine":   dict(cfg=dict(layers=6, dist="noflow", feature_dim=32, **SYMSOL), n=512, regime="trained", wseed=25, rseed=75, direction="forward", fisher=None),
    "noflow_affine_inv": dict(cfg=dict(layers=6, dist="noflow", feature_dim=32, **SYMSOL), n=512, regime="trained", wseed=25, rseed=76, direction="inverse", fisher=None),
    # conditional 3x3 ablation layers (per-sample matrix I + MLP(feature)): Gram-Schmidt with tangent log-det, Smith and polar rotations
    "cgs9_cond":       dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransLSmith"), n=512, regime="trained", wseed=26, rseed=77, direction="forward", fisher=None),
    "cgs9_cond_inv":   dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransLSmith"), n=512, regime="trained", wseed=26, rseed=78, direction="inverse", fisher=None),
    "csmithr9_cond":   dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransRSmith", last_affine=1), n=512, regime="trained", wseed=27, rseed=79, direction="forward", fisher=None),
    "csmithr9_cond_inv": dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransRSmith"), n=512, regime="trained", wseed=27, rseed=80, direction="inverse", fisher=None),
    "csvdl9_cond":     dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransLSVD"), n=512, regime="trained", wseed=28, rseed=81, direction="forward", fisher=None),
    "csvdl9_cond_inv": dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransLSVD"), n=512, regime="trained", wseed=28, rseed=82, direction="inverse", fisher=None),
    "csvdr9_cond":     dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransRSVD", frequent_permute=1), n=512, regime="trained", wseed=29, rseed=83, direction="forward", fisher=None),
    "csvdr9_cond_inv": dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="9TransRSVD"), n=512, regime="trained", wseed=29, rseed=84, direction="inverse", fisher=None),
    "cgs36_cond":      dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="36Trans"), n=512, regime="trained", wseed=30, rseed=85, direction="forward", fisher=None),
    "cgs36_cond_inv":  dict(cfg=dict(layers=3, condition=1, feature_dim=24, rot="36Trans", last_affine=1), n=512, regime="trained", wseed=30, rseed=86, direction="inverse", fisher=None),
    "embed_cond":      dict(cfg=dict(layers=3, condition=1, feature_dim=24, embedding=1, embedding_dim=8, rot="16UnTrans", last_affine=1), n=512,
                            regime="default", wseed=14, rseed=58, direction="forward", fisher=None),
}

# the conditional LU / SVD-rotation registry rows (flow/squeezetrans.py:94-144,264-277; flow/rottrans.py:37-66): per-sample matrices
# built by the reference with batched torch ops (batch-coupled torch.diag; U^T V of a batched SVD)
CASES.update({
    "clu16_cond":      dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16Trans", lu=1), n=512, regime="trained", wseed=51, rseed=151, direction="forward", fisher=None),
    "clu16_cond_inv":  dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16Trans", lu=1), n=512, regime="trained", wseed=51, rseed=152, direction="inverse", fisher=None),
    "clu16_first":     dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", lu=1, last_affine=1, first_affine=0, frequent_permute=1),
                            n=512, regime="trained", wseed=52, rseed=153, direction="forward", fisher=None),
    "clu9_cond":       dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="9TransLSmith", lu=1), n=512, regime="trained", wseed=53, rseed=154, direction="forward", fisher=None),
    "clu9_cond_inv":   dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="9TransLSmith", lu=1), n=512, regime="trained", wseed=53, rseed=155, direction="inverse", fisher=None),
    "crot16_cond":     dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16Rot"), n=512, regime="trained", wseed=54, rseed=156, direction="forward", fisher=None),
    "crot16_cond_inv": dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16Rot"), n=512, regime="trained", wseed=54, rseed=157, direction="inverse", fisher=None),
    "crot16_first":    dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnRot", last_affine=1, first_affine=0), n=512, regime="trained", wseed=55, rseed=158, direction="forward", fisher=None),
})


# Round 3: a third, adversarial weight regime ("imbalanced": synth.REGIMES -- random points of every MLP's
# ReLU-rescaling orbit over 12 binades, features x 30), and the BASELINE inverse configs at full depth in the trained regime.
C5 = dict(layers=42, condition=1, feature_dim=512, rot="None", frequent_permute=1, last_affine=0, first_affine=0)
C5U = dict(layers=42, condition=0, rot="None", last_affine=0, first_affine=0)
CASES.update({
    "c1_imbal":        dict(cfg=dict(layers=8), n=2048, regime="imbalanced", wseed=61, rseed=161, direction="forward", fisher=None),
    "c1_imbal_inv":    dict(cfg=dict(layers=8), n=1024, regime="imbalanced", wseed=61, rseed=162, direction="inverse", fisher=None),
    "c2_imbal":        dict(cfg=dict(layers=24), n=4096, regime="imbalanced", wseed=62, rseed=163, direction="forward", fisher="tilted"),
    "c4_imbal":        dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n=1024, regime="imbalanced", wseed=63, rseed=164, direction="forward", fisher=None),
    "c4_imbal_inv":    dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n=512, regime="imbalanced", wseed=63, rseed=165, direction="inverse", fisher=None),
    "c5u_imbal_inv":   dict(cfg=C5U, n=1024, regime="imbalanced", wseed=64, rseed=166, direction="inverse", fisher=None),
    "c5u_imbal_fwd":   dict(cfg=C5U, n=1024, regime="imbalanced", wseed=64, rseed=167, direction="forward", fisher=None),
    "c5_imbal_inv":    dict(cfg=C5, n=256, regime="imbalanced", wseed=65, rseed=168, direction="inverse", fisher="diag531"),
    "c4_trained_inv24": dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n=512, regime="trained", wseed=6, rseed=169, direction="inverse", fisher=None),
    "c5_trained_inv":  dict(cfg=C5, n=512, regime="trained", wseed=66, rseed=170, direction="inverse", fisher="diag531"),
    "c5_trained_fwd":  dict(cfg=C5, n=512, regime="trained", wseed=66, rseed=171, direction="forward", fisher=None),
    # more than 128 segments (flow/mobiusflow.py:7-14 takes any `segments`): the inverse keeps 64 segments per lane in registers and streams
    # the rest through a per-wave stash (round 3; round 2 refused K > 128); K = 196: 25 fc_last tiles, the last one half padding
    "k192_inv":        dict(cfg=dict(layers=2, segments=192), n=256, regime="trained", wseed=67, rseed=172, direction="inverse", fisher=None),
    "k196_cond_inv":   dict(cfg=dict(layers=2, segments=196, condition=1, feature_dim=24, rot="None", frequent_permute=1, last_affine=0, first_affine=0),
                            n=256, regime="trained", wseed=68, rseed=173, direction="inverse", fisher=None),
    "k196_cond_fwd":   dict(cfg=dict(layers=2, segments=196, condition=1, feature_dim=24, rot="None", frequent_permute=1, last_affine=0, first_affine=0),
                            n=256, regime="trained", wseed=68, rseed=174, direction="forward", fisher=None),
})


# Round 4: inverse passes with MORE THAN 64 SEGMENTS through the layer kinds of the extended instantiation -- conditional 3x3 Gram-Schmidt and
# 6x6 layers, a side layer (the batch-coupled conditional LU on the 3x3 layer) -- which round 3 refused (flow/mobiusflow.py:7-14 takes any
# `segments` with any `rot`)
CASES.update({
    "k96_cgs9_inv":    dict(cfg=dict(layers=2, segments=96, condition=1, feature_dim=24, rot="9TransLSmith"), n=256, regime="trained", wseed=71, rseed=181, direction="inverse", fisher=None),
    "k80_cgs36_inv":   dict(cfg=dict(layers=2, segments=80, condition=1, feature_dim=24, rot="36Trans", last_affine=1), n=256, regime="trained", wseed=72, rseed=182, direction="inverse", fisher=None),
    "k72_clu9_inv":    dict(cfg=dict(layers=2, segments=72, condition=1, feature_dim=24, rot="9TransLSmith", lu=1), n=256, regime="trained", wseed=73, rseed=183, direction="inverse", fisher=None),
    "k136_csvdl9_inv": dict(cfg=dict(layers=2, segments=136, condition=1, feature_dim=24, rot="9TransLSVD"), n=256, regime="trained", wseed=74, rseed=184, direction="inverse", fisher=None),
})


# Gradients through Flow.inverse (BinFind.backward, flow/mobiusflow.py:247-273): the reference's own autograd in fp64, loss =
# sum(a * ldj) + sum(B * R_out) with seeded a [n], B [n,3,3] (tests/golden/make_golden.py run_inverse_grad_case).
GRAD_CASES = {
    "invgrad_uncond":      dict(cfg=dict(layers=3, segments=16), n=192, regime="trained", wseed=41, rseed=141),
    "invgrad_mobius_only": dict(cfg=dict(layers=3, segments=8, rot="None", first_affine=0), n=160, regime="trained", wseed=42, rseed=142),
    "invgrad_cond":        dict(cfg=dict(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1,
                                         last_affine=1, first_affine=0), n=160, regime="trained", wseed=43, rseed=143),
    # segment counts beyond 64 (round 3: the 16-rotation backward kernel, csrc/train_block16.h, holds up to 512): gradients through
    # Flow.forward -- the training direction -- and through Flow.inverse
    "k96_train":           dict(cfg=dict(layers=2, segments=96), n=100, regime="trained", wseed=44, rseed=144, direction="forward"),
    "k200_cond_train":     dict(cfg=dict(layers=1, segments=200, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1,
                                         last_affine=1, first_affine=0), n=72, regime="trained", wseed=45, rseed=145, direction="forward"),
    "k96_invgrad":         dict(cfg=dict(layers=2, segments=96, rot="None", first_affine=0), n=80, regime="trained", wseed=46, rseed=146),
}
