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
    # small launches run the 4-wave instantiation of the stack kernel, mid-sized ones the 8-wave one, the launches above the 16-wave
    # one: same arithmetic
    with torch.no_grad():
        small = fl.log_prob(R[:4096].contiguous(), base=base)["logp"]
        mid = fl.log_prob(R[:40000].contiguous(), base=base)["logp"]
    assert (small - lp[:4096]).abs().max().item() < 2e-6
    assert (mid - lp[:40000]).abs().max().item() < 2e-6
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
        assert (ls - ldj[lo:]).abs().max().item() < 3e-6 and (Rs - Rt[lo:]).abs().max().item() < 3e-6
    pick = slice(n - 256, n)
    fn = orc.flow_forward if direction == "forward" else orc.flow_inverse
    wR, wl = fn(cfg, w, R[pick].cpu().numpy(), None if feat is None else feat[pick].cpu().numpy(), torch.float64)
    tol = 2e-4 if direction == "inverse" else 5e-5                           # inverse: one bisection cell is pi / 2^14
    assert (ldj[pick].cpu().double() - wl).abs().mean().item() < tol
