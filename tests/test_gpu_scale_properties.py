"""GPU (-m gpu): size-independent properties at BASELINE.json's full batch sizes (2^20 per GPU; 2^22 for the sharded config).
The oracle cannot run these sizes in seconds, so the checks are algebraic: per-sample independence (a rotation's density does
not depend on which batch, chunk or rank-shard it travels in), additivity of the fp64 NLL accumulation, and determinism."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import make_config, synth
from rotationnormflow_amd.dist import shard_bounds
from rotationnormflow_amd.utils.fisher import MatrixFisherN
from tests.gpu_helpers import product_flow

pytestmark = pytest.mark.gpu


def _c2():
    cfg = make_config("C2")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=2024, regime="trained")
    return cfg, w, product_flow(cfg, w)


def test_fisher24_full_batch_independence_additivity_determinism():
    cfg, w, fl = _c2()
    n = 1 << 20
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).cuda()
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")))
    with torch.no_grad():
        full = fl.log_prob(R, base=base)
        again = fl.log_prob(R, base=base)
    lp = full["logp"]
    assert torch.isfinite(lp).all()
    assert torch.equal(lp, again["logp"]) and torch.equal(full["sum"], again["sum"])             # bitwise deterministic
    s = full["sum"].cpu().numpy()
    assert s[1] == n and abs(s[0] - lp.double().sum().item()) < 1e-9 * abs(s[0])                  # fp64 accumulation
    # 8-way contiguous sharding (the multi-GPU partition): every shard reproduces its rows bit for bit, sums add up
    tot = 0.0
    for r in range(8):
        lo, hi = shard_bounds(n, r, 8)
        with torch.no_grad():
            part = fl.log_prob(R[lo:hi], base=base)
        assert torch.equal(part["logp"], lp[lo:hi])
        tot += part["sum"][0].item()
    assert abs(tot - s[0]) < 1e-9 * abs(s[0])
    # ragged re-batching: rows evaluated in a different position / tile / lane pair are unchanged
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(1))[: 100_003].cuda()
    with torch.no_grad():
        sub = fl.log_prob(R[idx].contiguous(), base=base)["logp"]
    assert torch.equal(sub, lp[idx])
    # small launches run the 4-wave instantiation of the stack kernel, mid-sized ones the 8-wave one, the full batch the 16-wave one -- all of
    # the SAME kernel family (round 4: the family -- lean / conditional-lean / general, which differ in their softplus form -- is fixed by the
    # flow's structure and never by the launch size, csrc/rnf_api.hip launch_stack): bit for bit the same rows
    with torch.no_grad():
        small = fl.log_prob(R[:4096].contiguous(), base=base)["logp"]
        mid = fl.log_prob(R[:40000].contiguous(), base=base)["logp"]
    assert torch.equal(small, lp[:4096])
    assert torch.equal(mid, lp[:40000])
    # spot check against the oracle (fp64) on a few hundred of those rows
    pick = idx[:512].cpu()
    want, _ = orc.log_prob(cfg, w, R[pick.cuda()].cpu().numpy(), None, synth.fisher_A("diag531"), torch.float64)
    assert abs(float(sub[:512].double().mean().cpu()) - float(want.mean())) < 2e-5


def test_cone_sized_batch_2pow22_mean_nll_matches_shard_average():
    """BASELINE configs[2]: 2^22 rotations as 8 shards of 2^19; the all-reduced statistic is the sum of the shard statistics."""
    cfg, w, fl = _c2()
    n = 1 << 22
    R = torch.from_numpy(synth.uniform_rotations(n, seed=7)).cuda()
    with torch.no_grad():
        whole = fl.log_prob(R)["sum"].cpu().numpy()
        parts = [fl.log_prob(R[slice(*shard_bounds(n, r, 8))])["sum"].cpu().numpy() for r in range(8)]
    assert whole[1] == n and sum(p[1] for p in parts) == n
    assert abs(sum(p[0] for p in parts) - whole[0]) < 1e-9 * abs(whole[0])
    assert np.isfinite(whole[0])


def test_conditional_chunking_is_invisible():
    """C4 structure above the 2^18-sample workspace chunk: chunk boundaries do not change any row."""
    cfg = make_config("C4", layers=4)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=5, regime="trained")
    fl = product_flow(cfg, w)
    n = (1 << 18) + 12345
    R = torch.from_numpy(synth.uniform_rotations(n, seed=3)).cuda()
    F = torch.from_numpy(synth.features(n, 256, seed=4)).cuda()
    with torch.no_grad():
        a = fl.log_prob(R, F)["logp"]
        lo = (1 << 18) - 1000
        b = fl.log_prob(R[lo:].contiguous(), F[lo:].contiguous())["logp"]
    # chunk boundaries, chunk sizes and the workgroup width a chunk's size selects are invisible: bit-identical rows
    assert torch.equal(a[lo:], b)


STRUCTURES = {
    "mobius_affine": dict(layers=3),
    "no_first_affine": dict(layers=3, first_affine=0),
    "last_affine": dict(layers=3, last_affine=1),
    "mobius_only": dict(layers=4, rot="None"),
    "affine_only": dict(layers=3, dist="noflow"),
    "k16_freq": dict(layers=3, segments=16, frequent_permute=1),
    "k8": dict(layers=2, segments=8),
    "lu": dict(layers=2, lu=1),
    "cond16_first": dict(layers=3, condition=1, feature_dim=16, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0),
}


@pytest.mark.parametrize("name", list(STRUCTURES))
@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_launch_shape_is_invisible(name, direction):
    """The library picks 4-, 8- or 16-wave workgroups, chunks and staging by launch size; a rotation's result must not depend on it.
    Rows are evaluated once in a launch just above each dispatch threshold and once in small pieces, for layer stacks that exercise
    every staging corner (affine blocks staged in LDS or not, consecutive constant layers, one fc_last tile, conditional first layer);
    a few hundred rows are also checked against the oracle."""
    cfg = make_config(None, **STRUCTURES[name])
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=11, regime="trained")
    fl = product_flow(cfg, w)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    F = cfg.feature_dim if cfg.condition else None
    for n in (cus * 128 + 33, cus * 256 + 31):                               # just into the 8-wave and the 16-wave forward launches
        R = torch.from_numpy(synth.uniform_rotations(n, seed=n % 97)).cuda()
        feat = torch.from_numpy(synth.features(n, F, seed=5)).cuda() if F else None
        run = (lambda r, f: fl(r, f)) if direction == "forward" else (lambda r, f: fl.inverse(r, f))
        with torch.no_grad():
            Rt, ldj = run(R, feat)
            lo = n - 777
            Rs, ls = run(R[lo:].contiguous(), None if feat is None else feat[lo:].contiguous())       # 777 rows: the 4-wave shape
        assert torch.isfinite(ldj).all()
        assert torch.equal(ls, ldj[lo:]) and torch.equal(Rs, Rt[lo:])
    pick = slice(n - 256, n)
    fn = orc.flow_forward if direction == "forward" else orc.flow_inverse
    wR, wl = fn(cfg, w, R[pick].cpu().numpy(), None if feat is None else feat[pick].cpu().numpy(), torch.float64)
    tol = 2e-4 if direction == "inverse" else 5e-5                           # inverse: one bisection cell is pi / 2^14
    assert (ldj[pick].cpu().double() - wl).abs().mean().item() < tol


def test_c4_full_size_independence_chunks_and_oracle():
    """BASELINE configs[3] at its full size: SYMSOL-I structure (Condition16Trans + 24 Moebius (3+256 inputs) + 23 Uncondition16Trans),
    2^20 rotations with their own 256-d feature rows.  Four workspace chunks of 2^18: rows must not depend on the chunk, the position in
    the batch or the launch shape they travel in; the fp64 sum is the sum of the rows; 256 rows are checked against the oracle (fp64)."""
    cfg = make_config("C4")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=2024, regime="trained")
    fl = product_flow(cfg, w)
    n = 1 << 20
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).cuda()
    F = torch.from_numpy(synth.features(n, 256, seed=43)).cuda()
    with torch.no_grad():
        full = fl.log_prob(R, F)
        again = fl.log_prob(R, F)
    lp = full["logp"]
    assert torch.isfinite(lp).all()
    assert torch.equal(lp, again["logp"]) and torch.equal(full["sum"], again["sum"])
    s = full["sum"].cpu().numpy()
    assert s[1] == n and abs(s[0] - lp.double().sum().item()) < 1e-9 * abs(s[0])
    # a window that straddles the second chunk boundary, evaluated on its own (other chunk offsets, other tiles)
    lo, hi = (1 << 19) - 70_001, (1 << 19) + 50_003
    with torch.no_grad():
        win = fl.log_prob(R[lo:hi].contiguous(), F[lo:hi].contiguous())["logp"]
    assert torch.equal(win, lp[lo:hi])
    # scattered rows in a small launch (8-wave instantiation of the same conditional-lean family): bit for bit
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:30_011].cuda()
    with torch.no_grad():
        sub = fl.log_prob(R[idx].contiguous(), F[idx].contiguous())["logp"]
    assert torch.equal(sub, lp[idx])
    pick = idx[:256]
    want, _ = orc.log_prob(cfg, w, R[pick].cpu().numpy(), F[pick].cpu().numpy(), None, torch.float64)
    got = lp[pick].cpu().double()
    assert abs(float(got.mean()) - float(want.mean())) < 1e-5
    assert float((got - want).abs().max()) < 2e-4


def test_c5_full_size_sample_inverse_roundtrip_and_oracle():
    """BASELINE configs[4] at its full size: 2^20 base samples from MatrixFisherN(diag(5,3,1)) drawn on the device, pushed through the
    inverse of the 42-layer Moebius-only conditional flow (F = 512 precomputed features), eval.py:327-347.  Property checks: finite
    rotations, forward(inverse(z)) == z up to the bisection cells of 42 layers, the forward log-det is minus the inverse one; 256 rows
    against the oracle's 15-step bisection (fp64)."""
    cfg = make_config("C5")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=2024, regime="trained")
    fl = product_flow(cfg, w)
    n = 1 << 20
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")).cuda())
    torch.manual_seed(7)
    z = base._sample(n).reshape(-1, 3, 3)
    F = torch.from_numpy(synth.features(n, 512, seed=44)).cuda()
    with torch.no_grad():
        x, ldj_inv = fl.inverse(z, F)
    assert torch.isfinite(x).all() and torch.isfinite(ldj_inv).all()
    assert (torch.linalg.det(x[:4096]) - 1).abs().max().item() < 1e-4
    with torch.no_grad():
        back, ldj_fwd = fl(x, F)
    rt = (back - z).abs().reshape(n, -1).max(1).values                     # 42 layers of pi / 2^14 cells, amplified by the later layers:
    assert rt.max().item() < 5e-2                                          # gated below against the reference's own round trip
    # rows do not depend on the chunk or position they travel in
    lo, hi = (1 << 18) - 5_001, (1 << 18) + 7_003
    with torch.no_grad():
        xs, ls = fl.inverse(z[lo:hi].contiguous(), F[lo:hi].contiguous())
    assert torch.equal(xs, x[lo:hi]) and torch.equal(ls, ldj_inv[lo:hi])
    # oracle spot check (the reference's batch-global 15-step bisection, fp64)
    pick = torch.arange(0, n, n // 256)[:256].cuda()
    wR, wl = orc.flow_inverse(cfg, w, z[pick].cpu().numpy(), F[pick].cpu().numpy(), dtype=torch.float64)
    rerr = (x[pick].cpu().double() - wR).abs().reshape(256, -1).max(1).values
    lerr = (ldj_inv[pick].cpu().double() - wl).abs()
    cell = np.pi / 2 ** 14
    assert rerr.median().item() < 0.5 * cell and rerr.max().item() < 12 * cell     # (the reference's own fp32 run: 8.5 cells at 42 layers)
    assert lerr.median().item() < 1e-4 and lerr.max().item() < 2e-2
    # round trip: no worse than what the reference's algorithm achieves on the same rows (its bisection leaves half a cell per layer too)
    bR, bl = orc.flow_forward(cfg, w, wR.numpy(), F[pick].cpu().numpy(), dtype=torch.float64)
    rt_ref = (bR - z[pick].cpu().double()).abs().reshape(256, -1).max(1).values
    assert rt[pick].mean().item() <= 1.5 * rt_ref.mean().item() + 1e-5
    assert rt.mean().item() <= 2.0 * rt_ref.mean().item() + 1e-5
    ld_ref = (bl + wl).abs().mean().item()
    assert (ldj_fwd + ldj_inv).abs().mean().item() <= 2.0 * ld_ref + 1e-5


def test_c3_global_batch_equals_the_sum_of_its_eight_shards():
    """BASELINE configs[2] at its full size: 2^22 rotations, split contiguously over 8 ranks (2^19 each, `dist.shard_bounds`) with one
    all-reduce of {sum log p, count}.  On one GPU: every shard reproduces its rows of the single 2^22 evaluation bit for bit (all launches run
    the same 16-wave instantiation), so the reduced mean NLL of the 8-GPU run equals the 1-GPU run's."""
    cfg = make_config("C3")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=2024, regime="trained")
    fl = product_flow(cfg, w)
    n = 1 << 22
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).cuda()
    with torch.no_grad():
        full = fl.log_prob(R)
    lp, s = full["logp"], full["sum"].cpu().numpy()
    assert torch.isfinite(lp).all() and s[1] == n
    tot = np.zeros(2)
    for r in range(8):
        lo, hi = shard_bounds(n, r, 8)
        assert hi - lo == 1 << 19
        with torch.no_grad():
            part = fl.log_prob(R[lo:hi])
        assert torch.equal(part["logp"], lp[lo:hi])
        tot += part["sum"].cpu().numpy()
    assert tot[1] == n and abs(tot[0] - s[0]) < 1e-9 * abs(s[0])
    assert abs(-tot[0] / tot[1] - (-s[0] / s[1])) < 1e-12 * abs(s[0] / s[1]) + 1e-12


@pytest.mark.parametrize("log2n", [14, 11])
def test_c3_small_global_batch_as_eight_shards_is_bit_equal(log2n):
    """a strong-scaled batch whose shards fall below the 16-wave launch size (2^14 rotations -> 2^11 per rank: the 4-wave
    instantiation; the whole batch on one GPU: 4- or 8-wave) must still give the 1-GPU rows bit for bit, and the same reduced mean NLL."""
    cfg = make_config("C3")
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=2024, regime="trained")
    fl = product_flow(cfg, w)
    n = 1 << log2n
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).cuda()
    with torch.no_grad():
        full = fl.log_prob(R)
    lp, s = full["logp"], full["sum"].cpu().numpy()
    tot = np.zeros(2)
    for r in range(8):
        lo, hi = shard_bounds(n, r, 8)
        with torch.no_grad():
            part = fl.log_prob(R[lo:hi])
        assert torch.equal(part["logp"], lp[lo:hi])
        tot += part["sum"].cpu().numpy()
    assert tot[1] == n and abs(tot[0] / tot[1] - s[0] / s[1]) < 1e-12 * abs(s[0] / s[1])
    # and the rows are those of a 2^20 evaluation (16-wave launches) that contains them
    big = torch.from_numpy(synth.uniform_rotations(1 << 20, seed=43)).cuda()
    big[5000:5000 + n] = R
    with torch.no_grad():
        assert torch.equal(fl.log_prob(big)["logp"][5000:5000 + n], lp)


@pytest.mark.parametrize("direction", ["forward", "inverse"])
@pytest.mark.parametrize("Q", [512, 500, 37])
def test_shared_rows_results_do_not_depend_on_the_shard(direction, Q):
    """Shared feature rows (feature_repeat = Q, the ROWS kernels): a rotation's result must not depend on the launch it travels in either.
    Shards cut at image boundaries start at multiples of Q, which for Q = 500 or 37 are NOT multiples of the 32-rotation wave tile: the same
    rotation then sits in another wave, next to other neighbours, in a wave that straddles two images or does not -- the row record enters x0
    as ONE form of matrix step whatever the wave looks like (csrc/flow_kernels.h GFragRows), so the rows are bit-equal."""
    cfg = make_config(layers=6, feature_dim=256, condition=1, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=91, regime="trained")
    fl = product_flow(cfg, w)
    B = 24
    R = torch.from_numpy(synth.uniform_rotations(B * Q, seed=92)).cuda()
    f = torch.from_numpy(synth.features(B, 256, seed=93)).cuda()
    run = (lambda r, ff: fl(r, ff, feature_repeat=Q)) if direction == "forward" else (lambda r, ff: fl.inverse(r, ff, feature_repeat=Q))
    with torch.no_grad():
        full = run(R, f)
        for lo, hi in ((0, 3), (3, 4), (4, 11), (11, 24)):                       # image ranges: starts at 3 Q, 4 Q, 11 Q rotations
            part = run(R[lo * Q: hi * Q].contiguous(), f[lo:hi].contiguous())
            assert torch.equal(part[0], full[0][lo * Q: hi * Q]) and torch.equal(part[1], full[1][lo * Q: hi * Q]), (lo, hi)


def test_shared_rows_wide_launch_equals_narrow_shards():
    """The same across workgroup widths: 160 images x 512 queries run the 16-wave ROWS kernel, a 3-image shard the 4-wave one, a 40-image shard
    the 8-wave one -- bit-equal rows (one kernel family, one arithmetic)."""
    cfg = make_config(layers=6, feature_dim=256, condition=1, rot="16UnTrans", frequent_permute=1, last_affine=1, first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=91, regime="trained")
    fl = product_flow(cfg, w)
    B, Q = 160, 512
    R = torch.from_numpy(synth.uniform_rotations(B * Q, seed=94)).cuda()
    f = torch.from_numpy(synth.features(B, 256, seed=95)).cuda()
    with torch.no_grad():
        full = fl.log_prob(R, f, feature_repeat=Q)["logp"]
        for lo, hi in ((7, 10), (100, 140)):
            part = fl.log_prob(R[lo * Q: hi * Q].contiguous(), f[lo:hi].contiguous(), feature_repeat=Q)["logp"]
            assert torch.equal(part, full[lo * Q: hi * Q]), (lo, hi)
