"""GPU: the default split-precision ("f16x2") arithmetic is scale-free.

The f16x2 kernels hold operands as fp16 hi + UNSCALED fp16 lo pairs, whose resolution is absolute (2^-25).  A ReLU network computes the
same function anywhere on its rescaling orbit (hidden layer 1 x s, layer 2 x s, layer 3 x s^-2, ...), so the packers first move every
conditioner MLP to one canonical point of that orbit (csrc/equalize.h) and only then split.  These tests take the golden cases, move
their weights along the orbit, and demand the SAME gates against the UNCHANGED reference fixtures."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import _lib, make_config, runtime, synth
from tests.gpu_helpers import product_flow
from tests.helpers import load_case

pytestmark = pytest.mark.gpu


def relu_rescale(weights, c1, c2, c0=1.0):
    """Function-preserving move of every ConditionalTransform in a state dict (flow/condition.py:24-30): x0, x3 * c0, x1 * c1, x2 * c2."""
    w = {k: v.copy() for k, v in weights.items()}
    for pre in sorted(k[: -len(".fc_first.weight")] for k in w if k.endswith(".fc_first.weight")):
        f = np.float32
        w[pre + ".fc_first.weight"] *= f(c0); w[pre + ".fc_first.bias"] *= f(c0)
        w[pre + ".layers.1.weight"] *= f(c1 / c0); w[pre + ".layers.1.bias"] *= f(c1)
        w[pre + ".layers.3.weight"] *= f(c2 / c1); w[pre + ".layers.3.bias"] *= f(c2)
        w[pre + ".layers.5.weight"] *= f(c0 / c2); w[pre + ".layers.5.bias"] *= f(c0)
        w[pre + ".fc_last.weight"] *= f(1.0 / c0)
    return w


def _run(cfg, w, R, feat, direction="forward", calibrate=False):
    fl = product_flow(cfg, w)
    Rd = torch.from_numpy(R).cuda()
    fd = None if feat is None else torch.from_numpy(feat).cuda()
    if calibrate:
        fl.calibrate_feature_scale(fd)
    with torch.no_grad():
        Rt, ldj = fl(Rd, fd) if direction == "forward" else fl.inverse(Rd, fd)
    torch.cuda.synchronize()
    return fl, Rt.cpu().numpy().astype(np.float64), ldj.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
@pytest.mark.parametrize("s", [2.0 ** -8, 2.0 ** -4, 2.0 ** 6])
@pytest.mark.parametrize("name", ["c2_trained", "c4_trained"])
def test_forward_gates_hold_on_the_relu_rescaling_orbit(name, s, precision):
    """The verdict's counter-example (layer 1 x s, layer 2 x s, layer 3 x s^-2): the gates of test_forward_matches_reference_golden against
    the unchanged ldj64 / rot64 fixtures, natively (no fp32 re-run), and for f16x2 bit-identical to the unscaled weights' result -- the
    packer lands on the same canonical record."""
    old = runtime.get_precision()
    runtime.set_precision(precision)
    try:
        cfg, w, R, feat, fx, spec = load_case(name)
        fl, Rt, ldj = _run(cfg, relu_rescale(w, s, s * s), R, feat)
        assert fl._packed(torch.device("cuda", torch.cuda.current_device())).precision == precision
        assert not runtime.fallback_fired(torch.device("cuda", torch.cuda.current_device()))
        noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
        err = np.abs(ldj - fx["ldj64"])
        assert abs(ldj.mean() - fx["ldj64"].mean()) < 1e-5
        assert err.mean() <= 2 * noise.mean() + 2e-6
        assert err.max() <= 4 * noise.max() + 2e-5
        rnoise = np.abs(fx["rot32"].astype(np.float64) - fx["rot64"]).max()
        assert np.abs(Rt - fx["rot64"]).max() <= 4 * rnoise + 1e-5
        if precision == "f16x2":
            _, Rt1, ldj1 = _run(cfg, w, R, feat)
            assert np.array_equal(ldj, ldj1) and np.array_equal(Rt, Rt1)
    finally:
        runtime.set_precision(old)


def test_without_equalisation_the_audit_sends_the_rescaled_flow_to_fp32():
    """What the safety net does when the equalisation is switched off: the s = 2^-8 flow is refused by the pack-time audit, packed for the
    exact-fp32 kernels, and still passes the gates."""
    L = _lib.lib()
    old = L.rnf_set_equalize(0)
    try:
        cfg, w, R, feat, fx, spec = load_case("c2_trained")
        fl, Rt, ldj = _run(cfg, relu_rescale(w, 2.0 ** -8, 2.0 ** -16), R, feat)
        assert fl._packed(torch.device("cuda", torch.cuda.current_device())).precision == "fp32"
        noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
        err = np.abs(ldj - fx["ldj64"])
        assert abs(ldj.mean() - fx["ldj64"].mean()) < 1e-5 and err.max() <= 4 * noise.max() + 2e-5
    finally:
        L.rnf_set_equalize(old)


def test_large_features_stay_native():
    """|feature| up to ~150 (un-normalised backbone outputs): inside the fp16 range of the hi / scaled-lo pairs, no fp32 re-run, and the
    result follows the fp64 oracle."""
    cfg = make_config(layers=4, segments=16, condition=1, feature_dim=40, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=21, regime="trained")
    for k in w:                                      # a network trained on such features has correspondingly small input weights
        if k.endswith("fc_first.weight"):
            v = w[k].copy(); v[:, (3 if ".conditioner." in k else 0):] /= np.float32(40.0); w[k] = v
    n = 2048
    R = synth.uniform_rotations(n, seed=22)
    feat = (synth.features(n, 40, seed=23) * np.float32(40.0)).astype(np.float32)
    assert np.abs(feat).max() > 100
    fl, Rt, ldj = _run(cfg, w, R, feat, calibrate=True)          # Flow.calibrate_feature_scale: the one explicit call features of this scale need
    assert not runtime.fallback_fired(torch.device("cuda", torch.cuda.current_device()))
    Rw, lw = orc.flow_forward(cfg, w, R, feat, dtype=torch.float64)
    _, l32 = orc.flow_forward(cfg, w, R, feat, dtype=torch.float32)
    noise = np.abs(l32.double().numpy() - lw.numpy())
    err = np.abs(ldj - lw.numpy())
    assert err.mean() <= 2 * noise.mean() + 2e-6 and err.max() <= 4 * noise.max() + 2e-5, (err.mean(), err.max(), noise.mean(), noise.max())
    # the equalisation was calibrated on this batch (explicitly): the packed flow knows its features have a mean square of ~1600
    assert 1000.0 < fl._packed(torch.device("cuda", torch.cuda.current_device())).feature_ms < 2500.0


def test_training_forward_with_a_huge_segment_weight_is_finite_and_differentiable():
    """the training passes run unguarded (no fp32 fallback images), so their softplus must be overflow-safe on its own.
    fc_last bias of one segment at 120: loss and gradients finite and equal to oracle autograd."""
    cfg = make_config(layers=2, segments=16)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=31, regime="trained")
    for k in w:
        if k.endswith("conditioner.fc_last.bias"):
            b = w[k].copy(); b[3] = 120.0; w[k] = b
    fl = product_flow(cfg, w).train()
    R = synth.uniform_rotations(96, seed=32)
    _, ldj = fl(torch.from_numpy(R).cuda())
    loss = (-ldj).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    p = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in w.items()}
    _, ldj_o = orc.flow_forward(cfg, p, torch.from_numpy(R).double(), None, dtype=torch.float64, grad=True)
    want = dict(zip(p, torch.autograd.grad((-ldj_o).mean(), list(p.values()))))
    assert abs(float(loss) - float((-ldj_o).mean())) < 1e-5
    for k, prm in fl.named_parameters():
        g = prm.grad.cpu().double()
        assert torch.isfinite(g).all(), k
        assert float((g - want[k]).abs().max()) <= 2e-4 * max(float(want[k].abs().max()), 1e-3), k


@pytest.mark.parametrize("mode", ["train", "replica"])
def test_side_layer_flow_evaluates_from_live_parameters(mode):
    """a flow with a ConditionRot / Condition16TransLU layer under no_grad in train() mode (the nn.Module default) or as an
    nn.DataParallel replica packs on the device from the live parameters; that path must hand the side layers to the launcher too."""
    cfg = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16Rot")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=54, regime="trained")
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(512, seed=156)).cuda()
    f = torch.from_numpy(synth.features(512, 24, seed=1156)).cuda()
    with torch.no_grad():
        Rt0, ldj0 = fl(R, f)
        if mode == "train":
            fl.train()
        else:
            fl._is_replica = True
        Rt1, ldj1 = fl(R, f)
        res = fl.log_prob(R, f)
    torch.cuda.synchronize()
    assert torch.isfinite(ldj1).all()
    assert float((ldj1 - ldj0).abs().max()) < 1e-4 and float((Rt1 - Rt0).abs().max()) < 1e-4
    assert float((res["logp"] - ldj0).abs().max()) < 1e-4


def test_device_packed_inference_is_guarded():
    """train()-mode inference packs on the device; the blob now carries the exact-fp32 images too, so the range guard (and with it the
    LEAN kernel) work there as in eval mode."""
    L = _lib.lib()
    eq, au = L.rnf_set_equalize(0), L.rnf_set_pack_audit(0)
    try:
        from tests.test_gpu_guard import _weights
        cfg = make_config(layers=3, segments=16)
        fl = product_flow(cfg, _weights(cfg, 5, True)).train()
        R = torch.from_numpy(synth.uniform_rotations(3000, seed=11)).cuda()
        with torch.no_grad():
            Rt, ldj = fl(R)
        assert runtime.fallback_fired(R.device)
        assert torch.isfinite(ldj).all() and torch.isfinite(Rt).all()
    finally:
        L.rnf_set_equalize(eq)
        L.rnf_set_pack_audit(au)


@pytest.mark.parametrize("name", ["c4_trained", "c4_imbal", "cond16_regular", "embed_cond"])
def test_fused_projection_kernel_matches_the_two_kernel_path(name):
    """rnf_set_fused(1): the feature projection inside the stack kernel (features resident in registers, projected one layer ahead into an
    L2-resident stash) -- same arithmetic as the pre-pass; gates vs the fixture too."""
    L = _lib.lib()
    cfg, w, R, feat, fx, spec = load_case(name)
    _, Rt0, ldj0 = _run(cfg, w, R, feat)
    old = L.rnf_set_fused(1)
    try:
        _, Rt1, ldj1 = _run(cfg, w, R, feat)
        n_big = 70000                                                  # more than one tile per workgroup: the tile-boundary path of the pipeline
        Rb = synth.uniform_rotations(n_big, seed=5)
        fb = synth.features(n_big, feat.shape[1], seed=6) * np.float32(synth.feature_scale(spec["regime"]))
        _, Rt3, ldj3 = _run(cfg, w, Rb, fb)
    finally:
        L.rnf_set_fused(old)
    _, Rt2, ldj2 = _run(cfg, w, Rb, fb)
    # same arithmetic up to the order of the residual add (the 8-wave kernels keep x0 in registers) and, for the fixture-sized launch, the
    # softplus form of the general instantiation: ulp-level differences that 24 layers stretch like the reference's own fp32 noise
    assert np.abs(ldj3 - ldj2).max() < 2e-4 and np.abs(ldj3 - ldj2).mean() < 3e-6 and np.abs(Rt3 - Rt2).max() < 2e-4
    assert np.abs(ldj1 - ldj0).max() < 2e-4 and np.abs(ldj1 - ldj0).mean() < 3e-6 and np.abs(Rt1 - Rt0).max() < 2e-4
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    err = np.abs(ldj1 - fx["ldj64"])
    assert err.mean() <= 2 * noise.mean() + 2e-6 and err.max() <= 4 * noise.max() + 2e-5


def test_evaluations_do_not_synchronise_the_host():
    """after the first call (which packs the parameters) an evaluation must not wait for the device anywhere --
    ConditionRot's per-sample SVD and the matrix-Fisher sampler's proper SVD used to go through host LAPACK and a device->host copy.
    torch's sync debug mode raises on every synchronising call."""
    from rotationnormflow_amd.utils.fisher import MatrixFisherN
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16Rot")           # ConditionRot layers
    fl = product_flow(cfg, synth.fill_state_dict(orc.state_shapes(cfg), seed=54, regime="trained"))
    R = torch.from_numpy(synth.uniform_rotations(2048, seed=156)).cuda()
    f = torch.from_numpy(synth.features(2048, 24, seed=1156)).cuda()
    cfg5 = make_config("C5", layers=3)
    fl5 = product_flow(cfg5, synth.fill_state_dict(orc.state_shapes(cfg5), seed=55, regime="trained"))
    f5 = torch.from_numpy(synth.features(2048, 512, seed=7)).cuda()
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")).to(dev))
    with torch.no_grad():
        fl(R, f); fl.inverse(R, f); fl5.inverse(R, f5); base._sample(16)        # first calls: packing, workspace allocation
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            a = fl(R, f)
            b = fl.inverse(R, f)
            z = base._sample(2048).reshape(-1, 3, 3)                             # eval.py:327-347: base samples, their density, inverse pass
            lp = base._log_prob(z)
            c = fl5.inverse(z, f5)
            d = fl5.log_prob(R, f5, base=base)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.isfinite(a[1]).all() and torch.isfinite(b[1]).all() and torch.isfinite(c[1]).all() and torch.isfinite(lp).all()
    assert torch.isfinite(d["logp"]).all()


def test_feature_square_sum_reduces_in_bounded_chunks():
    """the calibration reduction must not make an fp64 copy of the whole feature batch (round 5's vector_norm(dtype=float64)
    did: +2x the batch in transient HBM, 4 GB for C5's features).  256 MB of fp32 features: the peak grows by less than the batch itself."""
    n = 64 * 1024 * 1024
    f = torch.randn(n // 512, 512, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    sq = runtime.feature_square_sum(f)
    torch.cuda.synchronize()
    grown = torch.cuda.max_memory_allocated() - before
    assert grown < f.numel() * 4, grown                                     # (one 2^24-entry fp64 chunk: 128 MB)
    ref = sum(float(c.double().square().sum()) for c in f.reshape(-1).split(1 << 22))
    assert abs(float(sq[0]) - ref) <= 1e-9 * ref and float(sq[1]) == n
    assert abs(runtime.feature_mean_square(f) - 1.0) < 0.05
