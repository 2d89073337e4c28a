"""Case table of tests/golden/make_trained.py (generator, reference side) and tests/test_gpu_trained.py / test_oracle_trained.py (checkers).

TRAINED: name -> flow config + the synthetic ``raw``-style target + the optimisation recipe.  TRAJ: short Adam trajectories from recipe weights.
"""
SYMSOL = dict(condition=1, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)

TRAINED = {
    # BASELINE configs[0] structure (settings/raw.yml with --layers 8): two sharp matrix-Fisher modes (kappa 20: ~9 degrees per axis)
    "trained_c1": dict(cfg=dict(layers=8), n_modes=2, kappa=20.0, mode_seed=301, data_seed=302, init_seed=303,
                       n_train=16384, n_test=2048, n_inverse=512, steps=3000, batch=256, lr=1e-3),
    # 4-layer conditional flow, SYMSOL structure (Condition16Trans first, Uncondition16Trans between), F = 32: three feature classes, each a
    # two-mode target
    "trained_cond4": dict(cfg=dict(layers=4, feature_dim=32, **SYMSOL), n_classes=3, kappa=20.0, mode_seed=311, data_seed=312, init_seed=313,
                          n_train=16384, n_test=1024, n_inverse=512, steps=2500, batch=256, lr=1e-3),
    # BASELINE configs[1] structure at full depth (settings/raw.yml: 24 layer pairs, K = 64): three sharp modes; 48 layers of compounding
    # fc_last / softplus saturation
    "trained_c2": dict(cfg=dict(layers=24), n_modes=3, kappa=20.0, mode_seed=331, data_seed=332, init_seed=333,
                       n_train=32768, n_test=2048, n_inverse=256, steps=3000, batch=256, lr=5e-4),
    # BASELINE configs[3] structure (settings/symsol.yml with --layers 24 --feature_dim 256): four feature classes, each a two-mode target;
    # the features of class 0 arrive UN-NORMALISED (x30) to stress the pack-time equalisation of the conditional layers
    "trained_c4": dict(cfg=dict(layers=24, feature_dim=256, **SYMSOL), n_classes=4, kappa=20.0, mode_seed=341, data_seed=342, init_seed=343,
                       n_train=16384, n_test=1024, n_inverse=256, steps=1500, batch=256, lr=5e-4, class_scale={0: 30.0}),
}

TRAJ = {
    "traj_c1": dict(cfg=dict(layers=3, segments=32), n_modes=2, kappa=20.0, mode_seed=301, data_seed=322, wseed=323, regime="default",
                    steps=20, batch=256, lr=1e-3),
}
