"""GPU (-m gpu): the HIP path through the C ABI against the golden vectors of the reference and against the oracle.

Tolerances (fp32 path, bar from BASELINE.json north_star: "within 1e-5 fp32", "matching reference NLL to 1e-5"):
  * mean log-det over the batch (the mean-NLL statistic): |hip - ref_fp64| < 1e-5
  * per sample: the reference's own fp32 run deviates from its fp64 run by `noise` (stored in the fixture); the HIP
    result must be as close to the fp64 truth as that: mean err <= 2*mean noise + 2e-6, max err <= 4*max noise + 2e-5
  * inverse: the 15-step bisection resolves theta to pi/2^15 ~ 1e-4, decisions can flip under rounding (SURVEY section 7
    "Bisection parity"), so the inverse is compared against the reference's own fp32-vs-fp64 spread, x3.
"""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import _lib, make_config, runtime, synth
from rotationnormflow_amd.utils.fisher import MatrixFisherN
from tests.golden.cases import CASES
from tests.gpu_helpers import product_flow, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x2", "fp32", "bf16x3"], autouse=True)
def precision(request):
    """Every parity test runs for all three arithmetics of the conditioner GEMMs: the default split-precision fp16 MFMA path, the exact
    fp32-input MFMA path and (round 6) the strict bf16x3 path.  Same tolerances for all."""
    old = runtime.get_precision()
    runtime.set_precision(request.param)
    yield request.param
    runtime.set_precision(old)

FORWARD = [n for n, s in CASES.items() if s["direction"] == "forward"]
INVERSE = [n for n, s in CASES.items() if s["direction"] == "inverse"]          # incl. K = 128 (two staged halves) and K % 8 != 0


def test_native_library_is_loaded():
    L = _lib.lib()
    assert L.rnf_abi_version() == _lib.ABI_VERSION
    assert torch.cuda.is_available()


def test_conditioner_mfma_chain_matches_oracle():
    from rotationnormflow_amd.flow.condition import ConditionalTransform
    torch.manual_seed(0)
    K = 64
    m = ConditionalTransform(3, 4 * K)
    with torch.no_grad():
        m.fc_last.weight.mul_(5.0)
    y = torch.from_numpy(synth.uniform_rotations(1000, seed=1)[:, :, 0].copy())
    with torch.no_grad():
        got = m.cuda()(y.cuda()).cpu().double().numpy()
    p = {"c." + k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    want = orc.conditioner(y.double(), p, "c").numpy()
    assert np.abs(got - want).max() < 2e-5


@pytest.mark.parametrize("name", FORWARD)
def test_forward_matches_reference_golden(name, precision):
    fl, Rt, ldj, fx, spec, _ = run_case(name)
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    if "rot16" in name:
        # UnconditionRot / ConditionRot: U^T V depends on the SVD's arbitrary column signs, the reference's own fp32 and fp64 runs give
        # different rotations (tests/test_oracle_golden.py); the product runs ONE device SVD routine in fp32 (csrc/svd4_lapack.h) like the fp32 reference.
        err32 = np.abs(ldj - fx["ldj32"])
        if name.startswith("crot16"):
            # per-sample SVDs: where two singular values of a sample's matrix nearly coincide, U^T V is ill conditioned and the tiny
            # difference between the HIP conditioner's output and torch's Linear chain is amplified for that sample
            assert np.median(err32) < 2e-5 and np.quantile(err32, 0.95) < 1e-3, (np.median(err32), np.quantile(err32, 0.95))
            assert np.median(np.abs(Rt - fx["rot32"]).reshape(len(ldj), -1).max(1)) < 2e-5
        else:
            assert err32.max() < 5e-5 and abs(ldj.mean() - fx["ldj32"].astype(np.float64).mean()) < 1e-5
            assert np.abs(Rt - fx["rot32"]).max() < 5e-5
        return
    if name.startswith("clu16"):
        # Condition16TransLU: the reference's upper factor carries the batch-coupled diagonal on EVERY row (flow/squeezetrans.py:127),
        # the per-sample matrices are close to singular (log-dets down to -30) and the reference's own fp32 run is 1e-3 .. 1 away from
        # its fp64 run; the gate is that spread
        # (means over the best 99 % of the samples: the untrimmed means are set by the two or three samples whose matrices are closest to
        # singular -- errors of 0.4 .. 1.3 in the reference's own fp32 run -- and move by a factor of three with any change of rounding)
        err = np.abs(ldj - fx["ldj64"])
        trim = lambda a: np.sort(a)[: int(0.99 * len(a))].mean()
        assert trim(err) <= 3 * trim(noise) + 1e-5 and np.quantile(err, 0.99) <= 4 * np.quantile(noise, 0.99) + 1e-4
        return
    err = np.abs(ldj - fx["ldj64"])
    assert abs(ldj.mean() - fx["ldj64"].mean()) < 1e-5
    assert err.mean() <= 2 * noise.mean() + 2e-6
    assert err.max() <= 4 * noise.max() + 2e-5
    rnoise = np.abs(fx["rot32"].astype(np.float64) - fx["rot64"]).max()
    assert np.abs(Rt - fx["rot64"]).max() <= 4 * rnoise + 1e-5
    # outputs are rotations.  (Stacks WITHOUT Moebius layers -- the `--dist noflow` ablation -- go through the constant layers' 10x10 tables
    # only, which are linear in the rotation entries and do not re-orthonormalise: rounding of consecutive ill-conditioned matrices adds up to
    # 1.3e-5 there; every Moebius layer rebuilds two columns from unit vectors, so flows with them stay at 1e-6.)
    otol = 3e-5 if name.startswith("noflow") else 1e-5
    assert np.abs(np.einsum("nij,nkj->nik", Rt, Rt) - np.eye(3)).max() < otol
    assert np.abs(np.linalg.det(Rt) - 1).max() < otol
    # SURVEY 8(c), the per-sample form of the gate: |build - ref64| <= max(1e-5, 2 |ref32 - ref64|) sample by sample, and the 99th percentile
    # against the reference's fp32 run no worse than that run's own distance from fp64.  An INDEPENDENT fp32 evaluation of the same formulas
    # draws its own rounding noise, so a sample whose reference noise happens to be small can exceed twice that noise without being wrong:
    # measured pass fractions (tools/parity_stats.py --all-forward, profiles/r5/parity_stats.jsonl) are 0.991 - 1.000 for the default
    # split-precision arithmetic (it is CLOSER to fp64 than the reference's fp32 run) and 0.916 - 1.000 for the exact-fp32 kernels; the worst
    # excess over the per-sample bound stays below the reference's own worst noise.
    per = np.maximum(1e-5, 2 * noise)
    frac, excess = float(np.mean(err <= per)), float(np.max(err - per))
    p99_32 = float(np.quantile(np.abs(ldj - fx["ldj32"].astype(np.float64)), 0.99))
    p99_ref = float(np.quantile(noise, 0.99))
    if precision in ("f16x2", "bf16x3"):
        assert frac >= 0.99 and excess <= noise.max() + 1e-5, (frac, excess)       # (round 6: 0.985 -> 0.99; measured minimum 0.991, c4_imbal)
        assert p99_32 <= p99_ref + 1e-5, (p99_32, p99_ref)
    else:
        assert frac >= 0.88 and excess <= 2 * noise.max() + 1e-5, (frac, excess)
        assert p99_32 <= 1.5 * p99_ref + 1e-5, (p99_32, p99_ref)


@pytest.mark.parametrize("name", INVERSE)
def test_inverse_matches_reference_golden(name):
    fl, Rt, ldj, fx, spec, _ = run_case(name)
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    ref_l, ref_R = fx["ldj64"], fx["rot64"]
    if name.startswith("crot16") or name.startswith("clu16"):     # see the forward test: conventions of the SVD / near-singular matrices
        ref = fx["ldj32"].astype(np.float64) if name.startswith("crot16") else fx["ldj64"]
        err = np.abs(ldj - ref)
        if name.startswith("crot16"):
            assert np.median(err) < 2e-4 and np.quantile(err, 0.9) < 5e-3, (np.median(err), np.quantile(err, 0.9))
        else:
            assert err.mean() <= 3 * noise.mean() + 1e-4 and np.quantile(err, 0.99) <= 4 * np.quantile(noise, 0.99) + 1e-3
        return
    if "rot16" in name:                           # UnconditionRot: dtype-unstable in the reference itself, see the forward test
        ref_l, ref_R = fx["ldj32"].astype(np.float64), fx["rot32"].astype(np.float64)
        noise = np.full_like(noise, 2e-6)
    err = np.abs(ldj - ref_l)
    rnoise = np.abs(fx["rot32"].astype(np.float64) - ref_R).reshape(len(err), -1).max(1) + (0 if ref_R is fx["rot64"] else 2e-6)
    rerr = np.abs(Rt - ref_R).reshape(len(err), -1).max(1)
    # bulk: as close to the fp64 truth as the reference's own fp32 run
    assert err.mean() <= 3 * noise.mean() + 1e-5
    assert rerr.mean() <= 3 * rnoise.mean() + 1e-5
    # tail: a sample whose root sits within rounding error of a boundary of the bisection grid lands one cell (pi / 2^14 = 1.9e-4 rad)
    # away in EITHER implementation -- the reference's own fp32 and fp64 runs disagree on such samples too (tools/inverse_stats.py,
    # profiles/r2/inverse_stats.jsonl: 0.6 - 1.2 cells for 2 - 8 layer stacks, the reference's fp32 run 0.7 - 2.2) and in a deep stack
    # the shifts of several layers add up (42 layers: 6.6 cells here, 8.5 in the reference's fp32 run).  So: never more than 2 cells (one
    # flipped cell, stretched by the layers behind it: 1.3 observed) beyond the reference's own spread, no more samples off by half a cell than the reference has (+0.5 %), and the log-det within
    # 6 cells' worth (|d ldj / d theta| stays below ~6 on these weights) of it.
    cell = np.pi / 2 ** 14
    # (42-layer imbalanced stack, r3: one sample of 1024 sits 16 cells off where the reference's own fp32 run reaches 10 -- both
    # arithmetics, same sample; a maximum over ~1000 draws of a cascade of flipped cells is gated at twice the reference's own maximum)
    assert rerr.max() <= 2.0 * cell + (2.0 if "imbal" in name else 1.0) * rnoise.max()
    assert err.max() <= 6 * cell + noise.max()
    assert np.mean(rerr > 0.5 * cell) <= max(0.01, np.mean(rnoise > 0.5 * cell)) + 0.005


def _device_rotations(fl, fd):
    """{layer index: [n,4,4] fp64} -- U^T V of every ConditionRot layer as the DEVICE SVD routine (csrc/svd4_lapack.h) produced it for these
    features: the very matrices the stack kernel applied (build_side_buffer calls the same entry on the same inputs)."""
    from rotationnormflow_amd.flow.rottrans import ConditionRot
    out = {}
    for i, layer in enumerate(fl.layers):
        if isinstance(layer, ConditionRot):
            out[i] = layer._rnf_side(fd.to(torch.float32), grad=False).reshape(-1, 4, 4).cpu().double()
    return out


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("crot16")])
def test_condition_rot_every_sample_against_the_oracle_on_the_same_svd_factors(name):
    """Round 6: the percentile gates of the crot16 fixtures compare two SVD ROUTINES (the device restatement of LAPACK vs the
    reference's torch.svd) and leave 5 - 10 % of the samples unbounded.  Here the fp64 oracle applies the device routine's own U^T V
    (rot_override), so every other operation of every sample -- conditioners, quaternion maps, Moebius layers, bisection -- is gated at fp32
    rounding with NO sample exempt: a wrong sign or transposition in any of them would be an O(1) error."""
    from tests.helpers import load_case
    fl, Rt, ldj, fx, spec, (Rd, fd) = run_case(name)
    cfg, w, R, feat, _, _ = load_case(name)
    rots = _device_rotations(fl, fd)
    assert rots and all(float((r @ r.transpose(-1, -2) - torch.eye(4, dtype=torch.float64)).abs().max()) < 1e-5 for r in rots.values())
    fn = orc.flow_forward if spec["direction"] == "forward" else orc.flow_inverse
    want_R, want_l = fn(cfg, w, R, feat, dtype=torch.float64, rot_override=rots)
    err = np.abs(ldj - want_l.numpy())
    rerr = np.abs(Rt - want_R.numpy()).reshape(len(ldj), -1).max(1)
    if spec["direction"] == "forward":
        assert err.max() < 1e-4 and err.mean() < 1e-5, (err.max(), err.mean())
        assert rerr.max() < 1e-4 and rerr.mean() < 5e-6, (rerr.max(), rerr.mean())
        assert abs(ldj.mean() - want_l.numpy().mean()) < 1e-5
    else:       # the bisection grid: every sample within 2 cells (+ fp32 rounding), the log-det within 6 cells' worth of slope
        cell = np.pi / 2 ** 14
        assert rerr.max() <= 2.0 * cell + 1e-4, rerr.max() / cell
        assert err.max() <= 6 * cell + 1e-4 and err.mean() <= 0.5 * cell, (err.max() / cell, err.mean() / cell)
        assert np.mean(rerr > 0.5 * cell) <= 0.02


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("clu16")])
def test_condition_lu_every_sample_relative_to_its_condition_number(name):
    """Round 6: Condition16TransLU's per-sample matrices are close to singular for a few samples (log-dets down to -30), so
    the fixture gates trim the worst 1 %.  Here NO sample is exempt: its error is bounded by the amplification its own matrices allow,
    |ldj - ldj64| <= c * sum_layers cond(M_layer) * 2^-23 + 2e-5 with cond from the fp64 oracle's matrices (log|det M| - 4 log|M q| moves by
    ~ cond(M) times the relative rounding of M's entries, which come out of an fp32 conditioner)."""
    from tests.helpers import load_case
    fl, Rt, ldj, fx, spec, (Rd, fd) = run_case(name)
    cfg, w, R, feat, _, _ = load_case(name)
    p = {k: torch.from_numpy(np.asarray(v)).double() for k, v in w.items()}
    f64 = torch.from_numpy(feat).double()
    kappa = np.zeros(len(ldj))
    for i, kind in enumerate(orc.layer_kinds(cfg)):
        if kind == "clu16":
            M = orc.cond_lu_matrix(f64, p, f"layers.{i}.net", 4)
            kappa += torch.linalg.cond(M).numpy()
    assert kappa.min() >= 1.0
    err = np.abs(ldj - fx["ldj64"])
    # c: measured (profiles/r6/clu16_cond_stats.txt, worst sample of 512, (err - 2e-5) / (cond 2^-23)): forward 3.7 - 10.1 for the build, 4.5 -
    # 11.6 for the REFERENCE's own fp32 run; inverse pass (the per-sample matrix is inverted first: one more factor of the conditioning) 35 - 40
    # against the reference's 21.  Gates at ~1.5x the larger of the two.
    c = 64.0 if spec["direction"] == "inverse" else 16.0
    bound = c * kappa * 2.0 ** -23 + 2e-5
    worst = float(np.max(err / bound))
    assert worst <= 1.0, (worst, int(np.argmax(err / bound)), float(kappa.max()), float(err.max()))
    # ... and the reference's own fp32 run needs the same kind of slack: the bound is not looser than the reference's arithmetic by more than ~4x
    noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
    assert np.max(noise / bound) > 0.25


@pytest.mark.parametrize("name", ["c2_default", "c2_trained"])
def test_fused_log_prob_and_nll_sum(name):
    fl, Rt, ldj, fx, spec, (Rd, fd) = run_case(name)
    A = torch.from_numpy(synth.fisher_A(spec["fisher"]))
    base = MatrixFisherN(A)
    with torch.no_grad():
        res = fl.log_prob(Rd, fd, base=base, return_rotation=True)
        sep = base._log_prob(torch.from_numpy(Rt).float().cuda())
    torch.cuda.synchronize()
    lp = res["logp"].cpu().double().numpy()
    want = fx["ldj64"] + fx["fisher64"]
    noise = np.abs((fx["ldj32"].astype(np.float64) + fx["fisher32"]) - want)
    err = np.abs(lp - want)
    assert abs(lp.mean() - want.mean()) < 1e-5                      # mean NLL
    assert err.mean() <= 2 * noise.mean() + 2e-6
    assert err.max() <= 4 * noise.max() + 2e-5
    s = res["sum"].cpu().numpy()
    assert s[1] == lp.shape[0]
    assert abs(s[0] - lp.sum()) < 1e-6 * max(1.0, abs(lp.sum()))      # fp64 accumulation of the fp32 per-sample values
    assert np.abs(sep.cpu().double().numpy() + ldj - lp).max() < 2e-5  # separate base kernel agrees with the fused one
    assert np.abs(res["rotation"].cpu().double().numpy() - Rt).max() == 0.0


def test_roundtrip_and_unit_mass_at_scale():
    """Size-independent properties at a BASELINE-sized batch: inverse(forward(R)) ~ R, ldj_inv ~ -ldj_fwd, and
    mean(exp(ldj)) ~ 1 over Haar samples (the reference's own de-facto checks)."""
    n = 1 << 18
    cfg = orc.make_config(layers=24)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=77, regime="default")
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(n, seed=5)).cuda()
    with torch.no_grad():
        Rt, ldj = fl(R)
        Rb, ldjb = fl.inverse(Rt)
    torch.cuda.synchronize()
    assert torch.isfinite(ldj).all() and torch.isfinite(Rt).all()
    assert (Rb - R).abs().max().item() < 2e-3
    assert (Rb - R).abs().mean().item() < 4e-4
    assert (ldj + ldjb).abs().mean().item() < 1e-3
    mass = torch.exp(ldj.double()).mean().item()
    assert abs(mass - 1.0) < 0.02


def test_ragged_and_empty_batches():
    cfg = orc.make_config(layers=2)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=5, regime="trained")
    fl = product_flow(cfg, w)
    R = synth.uniform_rotations(777, seed=6)
    want_R, want_l = orc.flow_forward(cfg, w, R, None, torch.float64)
    for n in (0, 1, 31, 33, 255, 257, 777):
        with torch.no_grad():
            Rt, ldj = fl(torch.from_numpy(R[:n]).cuda())
        assert Rt.shape == (n, 3, 3) and ldj.shape == (n,)
        if n:
            assert (ldj.cpu().double() - want_l[:n]).abs().max().item() < 3e-5
            assert (Rt.cpu().double() - want_R[:n]).abs().max().item() < 3e-5


def test_errors_are_loud():
    cfg = orc.make_config(layers=1, condition=1, feature_dim=16)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=5)
    fl = product_flow(cfg, w)
    R = torch.from_numpy(synth.uniform_rotations(8, seed=6)).cuda()
    with torch.no_grad():
        with pytest.raises(AssertionError):
            fl(R)                                                   # conditional flow without a feature (mobiusflow.py:48-49)
        with pytest.raises(RuntimeError):
            fl(R.cpu(), torch.zeros(8, 16))                         # no CPU fallback
    Rb, lb = fl.inverse(R, torch.zeros(8, 16, device="cuda"))       # grad mode on, parameters require grad: differentiable (round 2)
    assert lb.requires_grad
    Rc, lc = fl(R, torch.zeros(2, 16, device="cuda"), feature_repeat=4)   # shared feature rows under autograd: expanded, differentiable (round 3)
    assert lc.requires_grad


@pytest.mark.parametrize("direction", ["forward", "inverse"])
@pytest.mark.parametrize("Q", [500, 37, 32, 7])
def test_shared_feature_rows_equal_materialised_repeat_and_the_oracle(direction, Q):
    """feature_repeat=Q (pose estimation: one image feature against Q query rotations, agent.py:238-263) must give what the reference's
    pattern gives -- every feature row repeated Q times -- including runs that straddle 32-rotation wave tiles (Q = 37, 500).  Round 5: rows of
    >= 32 rotations run on the SAME kernel family as the materialised repeat (conditional-lean forward / 8-wave inverse; ROWS instantiations,
    the per-row record enters x0 as one more exact-fp32 fc_first matrix step), shorter rows (Q = 7) on the extended instantiation.  Against the
    materialised repeat: fp32 rounding (forward; the inverse in bisection cells); against the fp64 oracle on the repeated features: the
    gates of the fixture tests, with the oracle's own fp32 run as the noise."""
    cfg = orc.make_config(layers=4, segments=16, condition=1, feature_dim=40, rot="16UnTrans", last_affine=1, frequent_permute=1)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=8, regime="trained")
    fl = product_flow(cfg, w)
    B = 9
    Rn = synth.uniform_rotations(B * Q, seed=9)
    fn = synth.features(B, 40, seed=10)
    R = torch.from_numpy(Rn).cuda()
    f = torch.from_numpy(fn).cuda()
    frep = f[:, None, :].expand(B, Q, 40).reshape(B * Q, 40).contiguous()
    with torch.no_grad():
        if direction == "forward":
            a = fl(R, f, feature_repeat=Q)
            b = fl(R, frep)
        else:
            a = fl.inverse(R, f, feature_repeat=Q)
            b = fl.inverse(R, frep)
    dR, dl = (a[0] - b[0]).abs().reshape(B * Q, -1).max(1)[0], (a[1] - b[1]).abs()
    if direction == "forward":
        assert dR.max().item() < 2e-5 and dl.max().item() < 5e-5
    else:
        # the inverse lands on the bisection grid (cells of pi / 2^14): the row form adds G to x0 in another order than the tile form (one more
        # fc_first matrix step instead of the accumulator start), and a root within rounding of a cell boundary may take the other cell
        cell = np.pi / 2 ** 14
        # (3 cells: two arithmetics that each flip a cell in consecutive layers -- 2.5 observed on one of 4500 samples)
        assert dR.max().item() <= 3.0 * cell and dl.max().item() <= 6 * cell and (dR > 0.5 * cell).float().mean().item() <= 0.01
        assert dR.median().item() < 2e-6 and dl.median().item() < 5e-6
    frep_n = np.repeat(fn, Q, axis=0)
    run = orc.flow_forward if direction == "forward" else orc.flow_inverse
    r64, l64 = run(cfg, w, Rn, frep_n, dtype=torch.float64)
    r32, l32 = run(cfg, w, Rn, frep_n, dtype=torch.float32)
    noise = (l32.double() - l64).abs().numpy()
    err = np.abs(a[1].cpu().double().numpy() - l64.numpy())
    if direction == "forward":
        assert err.mean() <= 2 * noise.mean() + 2e-6 and err.max() <= 4 * noise.max() + 2e-5, (err.mean(), err.max(), noise.mean(), noise.max())
        assert (a[0].cpu().double() - r64).abs().max().item() < 1e-4
    else:                                                          # the gates of test_inverse_matches_reference_golden (bisection cells)
        cell = np.pi / 2 ** 14
        rnoise = (r32.double() - r64).abs().reshape(len(err), -1).max(1)[0].numpy()
        rerr = (a[0].cpu().double() - r64).abs().reshape(len(err), -1).max(1)[0].numpy()
        assert err.mean() <= 3 * noise.mean() + 1e-5 and rerr.mean() <= 3 * rnoise.mean() + 1e-5
        assert rerr.max() <= 2.0 * cell + rnoise.max() and err.max() <= 6 * cell + noise.max()
    with torch.no_grad(), pytest.raises(ValueError):
        fl(R[:-1], f, feature_repeat=Q)


def test_fused_log_prob_with_shared_feature_rows():
    """Density of a few images on a grid of rotations (eval.py:444-462): flow.log_prob(grid, feature, feature_repeat=Q) against the
    materialised repeat."""
    cfg = orc.make_config(layers=3, segments=16, condition=1, feature_dim=24, rot="16UnTrans", last_affine=1)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=12, regime="trained")
    fl = product_flow(cfg, w)
    B, Q = 5, 333
    R = torch.from_numpy(synth.uniform_rotations(B * Q, seed=13)).cuda()
    f = torch.from_numpy(synth.features(B, 24, seed=14)).cuda()
    with torch.no_grad():
        a = fl.log_prob(R, f, feature_repeat=Q)
        b = fl.log_prob(R, f[:, None, :].expand(B, Q, 24).reshape(B * Q, 24).contiguous())
    assert (a["logp"] - b["logp"]).abs().max().item() < 5e-5
    assert abs(float(a["sum"][0] - b["sum"][0])) < 1e-3 and float(a["sum"][1]) == B * Q


def test_condition_lu_is_batch_coupled_like_the_reference():
    """ConditionLU's torch.diag over the batch dimension (flow/squeezetrans.py:127): the same rows in another batch order give other
    results for every sample -- reproduced, and equal to the oracle's restatement of that expression on the permuted batch."""
    from tests.helpers import load_case
    cfg, w, R, feat, fx, spec = load_case("clu9_cond")
    fl = product_flow(cfg, w)
    perm = np.random.default_rng(0).permutation(len(R))
    with torch.no_grad():
        _, a = fl(torch.from_numpy(R).cuda(), torch.from_numpy(feat).cuda())
        _, b = fl(torch.from_numpy(R[perm]).cuda(), torch.from_numpy(feat[perm]).cuda())
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert np.abs(b - a[perm]).mean() > 1e-2                              # NOT a per-sample function
    want = orc.flow_forward(cfg, w, R[perm], feat[perm], dtype=torch.float64)[1].numpy()
    assert np.abs(b - want).mean() < 2e-5
    # sharded evaluation of such a flow is refused (every rank would see other leading rows)
    from rotationnormflow_amd import dist
    with pytest.raises(NotImplementedError, match="batch"):
        dist.sharded_mean_nll(dist.flow_evaluator(fl), torch.from_numpy(R).cuda(), torch.from_numpy(feat).cuda(), rank=0, world=2)


def test_shared_feature_rows_are_differentiable():
    """eval.py:464-480 (nll_grad): d log p / d(query rotation) with every image feature repeated over its queries (agent.py:240-244).
    feature_repeat under autograd = the materialised repeat: same values, same gradients, and the feature gradient is summed per row."""
    cfg = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=71, regime="trained")
    fl = product_flow(cfg, w)
    Q, n_img = 16, 12
    R = torch.from_numpy(synth.uniform_rotations(Q * n_img, seed=72)).cuda()
    f = torch.from_numpy(synth.features(n_img, 24, seed=73)).cuda()
    Ra, fa = R.clone().requires_grad_(True), f.clone().requires_grad_(True)
    _, la = fl(Ra, fa, feature_repeat=Q)
    la.sum().backward()
    Rb, fb = R.clone().requires_grad_(True), f.clone().requires_grad_(True)
    _, lb = fl(Rb, fb.repeat_interleave(Q, dim=0))
    lb.sum().backward()
    assert torch.allclose(la, lb, atol=1e-6) and torch.allclose(Ra.grad, Rb.grad, atol=1e-5) and torch.allclose(fa.grad, fb.grad, atol=1e-4)
    assert fa.grad.shape == (n_img, 24) and float(fa.grad.abs().max()) > 0
    # side layers (ConditionRot) accept shared rows too: expanded on the host side of the launch
    cfg2 = make_config(layers=2, segments=16, condition=1, feature_dim=24, rot="16Rot")
    fl2 = product_flow(cfg2, synth.fill_state_dict(orc.state_shapes(cfg2), seed=74, regime="trained"))
    with torch.no_grad():
        _, l1 = fl2(R, f, feature_repeat=Q)
        _, l2 = fl2(R, f.repeat_interleave(Q, dim=0))
        l3 = fl2.log_prob(R, f, feature_repeat=Q)["logp"]
    assert torch.allclose(l1, l2, atol=1e-6) and torch.allclose(l3, l2, atol=1e-5)


def test_shared_feature_rows_inverse_with_more_than_64_segments():
    """Round 4: the pose-estimation pattern (one image feature against Q query rotations pushed through Flow.inverse, agent.py:238-263) for
    K = 96 segments -- round 3 refused shared feature rows on the K > 64 inverse.  Equal to the materialised repeat (the reference's
    feature.repeat, agent.py:240-244) and to the oracle."""
    cfg = make_config(layers=2, segments=96, condition=1, feature_dim=24, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=81, regime="trained")
    fl = product_flow(cfg, w)
    B, Q = 4, 96
    R = synth.uniform_rotations(B * Q, seed=82)
    f = synth.features(B, 24, seed=83)
    Rd, fd = torch.from_numpy(R).cuda(), torch.from_numpy(f).cuda()
    with torch.no_grad():
        xa, la = fl.inverse(Rd, fd, feature_repeat=Q)
        xb, lb = fl.inverse(Rd, fd.repeat_interleave(Q, dim=0))
    assert torch.allclose(la, lb, atol=2e-5) and torch.allclose(xa, xb, atol=2e-5)
    wR, wl = orc.flow_inverse(cfg, w, R, np.repeat(f, Q, axis=0), dtype=torch.float64)
    cell = np.pi / 2 ** 14
    assert (xa.cpu().double() - wR).abs().reshape(B * Q, -1).max(1).values.median().item() < cell
    assert (la.cpu().double() - wl).abs().median().item() < 2e-4


def test_reference_named_functions_calculate_16_and_9():
    """flow/squeezetrans.py's module-level calculate_16 / calculate_9 under their own names: a matrix per rotation (and one for all) through
    the stack kernel's side-matrix path, against the oracle's affine16 / gs9 (pinned to the reference), values and gradients."""
    from rotationnormflow_amd.flow import squeezetrans as st
    n = 1000
    R = torch.from_numpy(synth.uniform_rotations(n, seed=5))
    g = torch.Generator().manual_seed(9)
    M16 = torch.eye(4) + 0.3 * torch.randn(n, 4, 4, generator=g)
    M9 = torch.eye(3) + 0.3 * torch.randn(n, 3, 3, generator=g)
    for fn, ofn, M in ((st.calculate_16, orc.affine16, M16), (st.calculate_9, orc.gs9, M9)):
        Rt, ldj = fn(M.cuda(), R.cuda())
        want_R, want_l = ofn(M.double(), R.double())
        assert (Rt.cpu().double() - want_R).abs().max().item() < 2e-5
        assert (ldj.cpu().double() - want_l).abs().max().item() < 2e-5
        Rt1, ldj1 = fn(M[:1].cuda(), R.cuda())                      # one matrix for every rotation
        w1R, w1l = ofn(M[:1].double().expand(n, *M.shape[1:]), R.double())
        assert (Rt1.cpu().double() - w1R).abs().max().item() < 2e-5 and (ldj1.cpu().double() - w1l).abs().max().item() < 2e-5
        Mg = M[:64].cuda().requires_grad_(True)                     # differentiable w.r.t. the matrices, like the layers built on it
        Rt2, ldj2 = fn(Mg, R[:64].cuda())
        (ldj2.sum() + (Rt2 * Rt2.detach().roll(1, 0)).sum()).backward()
        Mo = M[:64].double().requires_grad_(True)
        oR, ol = ofn(Mo, R[:64].double())
        (ol.sum() + (oR * oR.detach().roll(1, 0)).sum()).backward()
        assert (Mg.grad.cpu().double() - Mo.grad).abs().max().item() < 2e-4 * max(1.0, Mo.grad.abs().max().item())
    assert torch.allclose(st.my_det_4_4(M16.cuda()), torch.linalg.det(M16.cuda()), atol=1e-5)
    assert torch.allclose(st.my_det_3_3(M9.cuda()), torch.linalg.det(M9.cuda()), atol=1e-5)
    with pytest.raises(ValueError):
        st.calculate_16(M16[:7].cuda(), R.cuda())
