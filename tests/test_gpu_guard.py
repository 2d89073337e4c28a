"""GPU: the range guard of the split-precision kernels.  An activation (or feature) beyond the fp16 range turns a sample's log-det into
NaN inside the f16x2 kernels; the call is then re-run, on the device and without a host synchronisation, on the exact-fp32 kernels
(include/rnf_hip.h desc columns 6 / 7; flow/condition.py:24-30 knows no such range limit)."""
import numpy as np
import pytest
import torch

import rotationnormflow_amd as rnf
from oracle import flow_oracle as orc
from rotationnormflow_amd import make_config, runtime, synth
from rotationnormflow_amd.utils.fisher import MatrixFisherN
from tests.gpu_helpers import product_flow

pytestmark = pytest.mark.gpu


@pytest.fixture
def unequalised():
    """The packers normally move every MLP to the canonical point of its ReLU-rescaling orbit (csrc/equalize.h), which also takes the
    hidden activations of the blown-up weights below back into range: these tests of the RUN-TIME guard switch that (and the pack-time
    audit's refusal) off."""
    from rotationnormflow_amd import _lib
    L = _lib.lib()
    eq, au = L.rnf_set_equalize(0), L.rnf_set_pack_audit(0)
    yield
    L.rnf_set_equalize(eq)
    L.rnf_set_pack_audit(au)


def _weights(cfg, seed, blow_up):
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=seed, regime="trained")
    if blow_up:                       # weights stay far inside the fp16 range, the hidden activations of every Moebius layer do not
        for k in w:
            if k.endswith("conditioner.fc_first.weight") or k.endswith("conditioner.fc_first.bias"):
                w[k] = w[k] * np.float32(40.0)
            if k.endswith("conditioner.layers.1.weight"):
                w[k] = w[k] * np.float32(2000.0)
            if k.endswith("conditioner.layers.3.weight") or k.endswith("conditioner.layers.5.weight"):
                w[k] = w[k] * np.float32(1.0e-2)
    return w


@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_activation_overflow_falls_back_to_fp32(direction, unequalised):
    cfg = make_config(layers=3, segments=16)
    w = _weights(cfg, 5, True)
    assert max(float(np.abs(v).max()) for v in w.values()) < 6.0e4
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(3000, seed=11)).cuda()
    assert rnf.get_precision() == "f16x2"
    with torch.no_grad():
        Rt, ldj = fl(R) if direction == "forward" else fl.inverse(R)
    assert fl._packed(R.device).precision == "f16x2"            # the weights themselves packed fine
    assert runtime.fallback_fired(R.device)
    assert torch.isfinite(ldj).all() and torch.isfinite(Rt).all()
    # the re-run IS the strict path (round 6: the bf16x3 kernels; RNF_FALLBACK=fp32 keeps the exact fp32-input MFMA of rounds 2 - 5):
    # bit-identical to a call that asks for it
    rnf.set_precision(runtime._fallback_precision)
    try:
        with torch.no_grad():
            Rt32, ldj32 = fl(R) if direction == "forward" else fl.inverse(R)
    finally:
        rnf.set_precision("f16x2")
    assert torch.equal(ldj, ldj32) and torch.equal(Rt, Rt32)
    # and it follows the oracle (fp64)
    fn = orc.flow_forward if direction == "forward" else orc.flow_inverse
    Rw, lw = fn(cfg, w, R.cpu().numpy(), None, dtype=torch.float64)
    err = np.abs(ldj.cpu().double().numpy() - lw.numpy())
    assert np.median(err) < 1e-4 and np.quantile(err, 0.99) < 2e-2, (np.median(err), err.max())


def test_no_fallback_in_range_and_fused_sum_uses_the_rerun(unequalised):
    cfg = make_config(layers=3, segments=16)
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")))
    R = torch.from_numpy(synth.uniform_rotations(2500, seed=12)).cuda()
    fl = product_flow(cfg, _weights(cfg, 6, False))
    with torch.no_grad():
        res = fl.log_prob(R, base=base)
    assert not runtime.fallback_fired(R.device)
    assert abs(float(res["sum"][0]) - float(res["logp"].double().sum())) < 1e-6 * abs(float(res["sum"][0]))
    fl = product_flow(cfg, _weights(cfg, 6, True))
    with torch.no_grad():
        res = fl.log_prob(R, base=base)
    assert runtime.fallback_fired(R.device)
    assert torch.isfinite(res["logp"]).all()
    assert abs(float(res["sum"][0]) - float(res["logp"].double().sum())) < 1e-6 * abs(float(res["sum"][0]))
    assert float(res["sum"][1]) == 2500.0


def test_feature_overflow_falls_back_to_fp32():
    cfg = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=8, regime="trained")
    fl = product_flow(cfg, w)
    n = 1500
    R = torch.from_numpy(synth.uniform_rotations(n, seed=13)).cuda()
    feat = synth.features(n, 24, seed=14)
    feat[7, 3] = 1.0e5                                          # one feature entry outside the fp16 range
    fd = torch.from_numpy(feat).cuda()
    with torch.no_grad():
        Rt, ldj = fl(R, fd)
    assert runtime.fallback_fired(R.device)
    assert torch.isfinite(ldj).all() and torch.isfinite(Rt).all()
    _, lw = orc.flow_forward(cfg, w, R.cpu().numpy(), feat, dtype=torch.float64)
    err = np.abs(ldj.cpu().double().numpy() - lw.numpy())
    assert np.quantile(err, 0.99) < 1e-3, err.max()


def test_side_layer_conditioner_overflow_is_reported_one_call_later():
    """Training a flow with a side layer (Condition16TransLU): a weight of the layer's conditioner beyond the fp16 range is reported by the
    device packer through a status word read one call later (no step waits for the device), as for the flow's own blob."""
    import contextlib
    import io
    from rotationnormflow_amd import runtime, synth
    from rotationnormflow_amd.configs import make_config
    from rotationnormflow_amd.flow.flow import Flow
    if runtime.get_precision() != "f16x2":
        pytest.skip("only the split-precision kernels have a range limit")
    cfg = make_config(layers=1, segments=8, condition=1, feature_dim=24, lu=1)
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    fl = fl.cuda().train()
    side = [m for m in fl.modules() if type(m).__name__ == "ConditionLU"][0]
    with torch.no_grad():
        side.w_l_net.layers[1].weight[0, 0] = 1.0e30           # (1e5 is no longer outside the range: the packer's equalisation rescales the unit)
    R = torch.from_numpy(synth.uniform_rotations(64, seed=1)).cuda()
    f = torch.from_numpy(synth.features(64, 24, seed=2)).cuda()
    from rotationnormflow_amd import autograd
    try:
        with pytest.raises(runtime.HalfRangeError):       # surfaces at the next conditioner call whose predecessor's status word has landed
            fl(R, f)
            torch.cuda.synchronize()
            fl(R, f)
    finally:
        torch.cuda.synchronize()
        autograd._pending_mlp_flags.clear()               # status words of this test's other calls must not surface in a later test


@pytest.mark.parametrize("n", [1500, 70000])
def test_segment_weight_beyond_the_one_piece_softplus(n):
    """softplus(s) for s = 120 and s = 400 (the reference's softplus returns s itself beyond its threshold of 20; weights the reference's own
    training produced reach s = 250, tests/golden/trained_cond4).  Round 3's lean one-piece log2(1 + 2^(s log2 e)) overflowed there and the
    whole launch was re-run on the exact-fp32 kernels; since round 4 the lean form takes the median of (log2(1 + 2^x), x, 127)
    (so3_math.h softplus2_lean) and every launch size runs the same lean family: no re-run, the result follows the oracle."""
    cfg = make_config(layers=2, segments=16)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=9, regime="trained")
    for k in w:
        if k.endswith("conditioner.fc_last.bias"):
            b = w[k].copy()
            b[3] = 120.0                                           # the raw weight of segment 3: softplus(120) = 120 in the reference
            b[7] = 400.0                                           # 2^(400 log2 e) is inf in fp32
            w[k] = b
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(n, seed=13)).cuda()
    with torch.no_grad():
        Rt, ldj = fl(R)
    assert not runtime.fallback_fired(R.device)
    assert torch.isfinite(ldj).all() and torch.isfinite(Rt).all()
    m = 2000
    Rw, lw = orc.flow_forward(cfg, w, R[:m].cpu().numpy(), None, dtype=torch.float64)
    assert np.abs(ldj[:m].cpu().double().numpy() - lw.numpy()).max() < 1e-3
    assert np.abs(Rt[:m].cpu().double().numpy() - Rw.numpy()).max() < 1e-4


@pytest.mark.parametrize("n", [1500, 70000])
def test_all_segment_weights_tiny(n):
    """Every raw segment weight at -12: softplus ~ 6e-6.  The one-piece form of the lean kernels would round fl(1 + e) to ~7 bits of e: their
    layer finish flags the tiny weight SUM and the exact-fp32 kernels re-run the launch -- at every launch size (round 4: the lean family is
    chosen by the flow's structure, not by the batch size)."""
    cfg = make_config(layers=2, segments=16)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=10, regime="trained")
    for k in w:
        if k.endswith("conditioner.fc_last.weight"):
            v = w[k].copy(); v[:16] *= np.float32(0.05); w[k] = v
        if k.endswith("conditioner.fc_last.bias"):
            b = w[k].copy(); b[:16] = -12.0; w[k] = b
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(n, seed=14)).cuda()
    with torch.no_grad():
        Rt, ldj = fl(R)
    assert runtime.fallback_fired(R.device)
    m = 2000
    Rw, lw = orc.flow_forward(cfg, w, R[:m].cpu().numpy(), None, dtype=torch.float64)
    err = np.abs(ldj[:m].cpu().double().numpy() - lw.numpy())
    assert err.max() < 5e-5 and err.mean() < 5e-6, (err.max(), err.mean())


@pytest.mark.parametrize("mode", ["weights", "first-batch"])
def test_repeated_guard_firing_warns_and_recalibrates(mode):
    """a conditional flow whose features are far larger than the scale its images were packed for trips the x0 guard and every
    launch is silently re-run on the exact-fp32 kernels.  The runtime notices (asynchronously -- no call waits for the device) and warns once.
    "first-batch" calibration (rounds 3 - 5): it then re-calibrates on the batch at hand and the guard is quiet again.  "weights" (the
    default since round 6: nothing is measured or re-packed behind the caller's back) the warning names the call, the results stay right on
    the fp32 re-runs, and Flow.calibrate_feature_scale(features) makes the guard quiet."""
    import time
    import warnings
    if runtime.get_precision() != "f16x2":
        pytest.skip("only the split-precision kernels are guarded")
    old_mode = runtime.get_feature_calibration()
    runtime.set_feature_calibration(mode)
    try:
        cfg = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
        w = synth.fill_state_dict(orc.state_shapes(cfg), seed=12, regime="trained")
        fl = product_flow(cfg, w)
        n = 4096
        R = torch.from_numpy(synth.uniform_rotations(n, seed=15)).cuda()
        f_small = synth.features(n, 24, seed=16)
        f_big = (f_small * 3000.0).astype(np.float32)               # x0 lands far beyond kX0Guard = 64 for images packed for unit-scale features
        with torch.no_grad():
            fl.log_prob(R, torch.from_numpy(f_small).cuda())
            assert not runtime.fallback_fired(R.device)
            fb = torch.from_numpy(f_big).cuda()
            want, _ = orc.log_prob(cfg, w, R[:256].cpu().numpy(), f_big[:256], None, torch.float64)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                fired = []
                for it in range(12):
                    got = fl.log_prob(R, fb)["logp"]
                    fired.append(runtime.fallback_fired(R.device))   # (synchronises: the watch words of earlier calls have landed)
                    time.sleep(0.01)
                assert (got[:256].cpu().double() - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))
                assert fired[0] and fired[1]                             # stale scale: the guard fires ...
                msgs = [str(c.message) for c in caught if "range guard" in str(c.message)]
                assert len(msgs) == 1
                if mode == "first-batch":
                    assert not any(fired[-3:]), fired                    # ... until the runtime re-calibrated on the large features
                else:
                    assert all(fired), fired                             # ... and keeps firing: nothing is re-packed behind the caller's back
                    assert "calibrate_feature_scale" in msgs[0]
                    assert fl.calibrate_feature_scale(fb) > 1e6
                    got = fl.log_prob(R, fb)["logp"]
                    assert not runtime.fallback_fired(R.device)
                    assert (got[:256].cpu().double() - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))
    finally:
        runtime.set_feature_calibration(old_mode)


def _rows_in_a_fresh_process(first_scale):
    """Child of test_conditional_rows_do_not_depend_on_the_first_batch_a_process_sees: load trained_c4.pth through the harness, evaluate a batch
    whose features are scaled by `first_scale` FIRST, then the fixture's own rows; prints the sha256 of those rows' log-dets."""
    import hashlib
    import sys
    from rotationnormflow_amd import harness
    from tests.trained_helpers import load_trained
    cfg, ckpt, w, fx, spec = load_trained("trained_c4")
    flow = harness.build_flow_from_checkpoint(cfg, ckpt)
    R = torch.from_numpy(fx["test_rot"].astype(np.float32)).cuda()
    f = torch.from_numpy(fx["test_feat"].astype(np.float32)).cuda()
    with torch.no_grad():
        flow(R, f * float(first_scale))
        _, ldj = flow(R, f)
    sys.stdout.write("ROWS " + hashlib.sha256(ldj.cpu().numpy().tobytes()).hexdigest() + "\n")


def test_conditional_rows_do_not_depend_on_the_first_batch_a_process_sees():
    """two FRESH processes that see different first batches (features x 1 and x 1/50) produce bit-identical rows for
    trained_c4.pth without calling set_feature_scale -- the packed images are a function of the checkpoint (weights + its feature-scale
    sidecar tests/golden/trained_c4.pth.rnf.json), not of the data a process happens to see first.  (Rounds 3 - 5 calibrated on the first
    batch: the x 1/50 process packed other images.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for scale in ("1.0", "0.02"):
        out = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {root!r}); from tests.test_gpu_guard import _rows_in_a_fresh_process as f; f({scale})"],
                             capture_output=True, text=True, cwd=root, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("ROWS ")][-1])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_a_non_finite_feature_row_stays_with_its_own_image_on_the_shared_row_kernels(direction):
    """Shared feature rows (feature_repeat = Q) enter x0 through a matrix step on the ROWS kernels, where the NaN record of one image meets the
    0 indicator of its wave-mates from the next image (NaN x 0 = NaN).  Those launches run guarded: the guard fires, the exact-fp32 re-run reads
    the records per lane, and only the rotations of the bad image come out NaN (as the reference's would: torch.relu propagates NaN; the
    kernels flag a NaN x0 and poison the sample's outputs) -- every other image as in a clean run (fp32 arithmetic)."""
    cfg = make_config(layers=4, segments=16, condition=1, feature_dim=40, rot="16UnTrans", last_affine=1, frequent_permute=1)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=8, regime="trained")
    fl = product_flow(cfg, w)
    B, Q = 6, 37                                                  # 37: every wave straddles two images somewhere
    R = torch.from_numpy(synth.uniform_rotations(B * Q, seed=9)).cuda()
    f = torch.from_numpy(synth.features(B, 40, seed=10)).cuda()
    fl.set_feature_scale(1.0)                                     # (a NaN row must not enter the calibration)
    run = (lambda ff: fl(R, ff, feature_repeat=Q)[1]) if direction == "forward" else (lambda ff: fl.inverse(R, ff, feature_repeat=Q)[1])
    with torch.no_grad():
        clean = run(f)
        assert not runtime.fallback_fired(R.device)
        bad = f.clone()
        bad[2] = float("nan")
        got = run(bad)
        assert runtime.fallback_fired(R.device)
    mine = torch.zeros(B * Q, dtype=torch.bool, device="cuda")
    mine[2 * Q: 3 * Q] = True
    assert torch.isnan(got[mine]).all() and torch.isfinite(got[~mine]).all()
    tol = 2e-4 if direction == "forward" else 6 * np.pi / 2 ** 14
    assert (got[~mine] - clean[~mine]).abs().max().item() < tol
