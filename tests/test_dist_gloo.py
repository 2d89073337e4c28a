"""CPU, world_size=2, gloo: the N>1 path (shard -> local {sum, count} -> all-reduce -> mean NLL) gives the same mean NLL
as one process over the whole batch.  The per-shard evaluator here is the oracle (the HIP evaluator needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import flow_oracle as orc
from rotationnormflow_amd import dist as rdist
from rotationnormflow_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = orc.make_config(layers=2)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=3, regime="trained")
    A = synth.fisher_A("diag531")
    R = torch.from_numpy(synth.uniform_rotations(n, seed=11))

    def evaluate(rot, feat):
        lp, _ = orc.log_prob(cfg, w, rot, None, A, torch.float32)
        return torch.tensor([lp.double().sum().item(), float(rot.shape[0])], dtype=torch.float64)

    nll, tot = rdist.sharded_mean_nll(evaluate, R, None, rank, world)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([float(nll), float(tot[0]), float(tot[1])]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 1001])
def test_two_rank_mean_nll_matches_single_process(tmp_path, n):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)                       # every rank ends with the same reduced statistic
    cfg = orc.make_config(layers=2)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=3, regime="trained")
    lp, nll = orc.log_prob(cfg, w, synth.uniform_rotations(n, seed=11), None, synth.fisher_A("diag531"), torch.float32)
    assert r0[2] == n
    assert abs(r0[0] - nll) < 1e-6


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 8, 1000, 1001, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [rdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo <= hi for lo, hi in spans)


def _grad_blob_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from rotationnormflow_amd import dist as rdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        blob = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        rdist.all_reduce_mean_(blob)
        q.put((rank, blob.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_gradient_blob_all_reduce_mean_world2():
    """Training exchange: ONE all-reduce averages the whole gradient blob over the ranks (rotationnormflow_amd.dist.all_reduce_mean_)."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + 7
    procs = [ctx.Process(target=_grad_blob_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    want = np.arange(1000, dtype=np.float32) * 1.5
    assert np.allclose(got[0], want) and np.allclose(got[1], want)


def _calib_worker(rank, world, port, out_dir):
    import contextlib
    import io
    import torch.distributed as dist
    from rotationnormflow_amd import make_config
    from rotationnormflow_amd.flow.flow import Flow
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            fl = Flow(make_config(None, layers=2, condition=1, feature_dim=16, rot="16UnTrans", last_affine=1, first_affine=0, frequent_permute=1))
        # shards of very different scale: per-process calibration would disagree by a factor of 16
        feat = torch.from_numpy(synth.features(1001, 16, seed=9))
        feat[:501] *= 4.0
        lo, hi = rdist.shard_bounds(1001, rank, world)
        ms = rdist.calibrate_feature_scale(fl, feat[lo:hi])
        again = rdist.calibrate_feature_scale(fl, feat[lo:hi] * 100.0)            # already fixed: no second measurement
        np.save(os.path.join(out_dir, f"ms{rank}.npy"), np.array([ms, again, fl._feature_ms_fixed]))
    finally:
        dist.destroy_process_group()


def test_feature_calibration_is_one_value_for_all_ranks(tmp_path):
    """the ranks of a sharded evaluation agree on ONE feature mean square (all-reduce of {sum f^2, count}), equal to the
    value a single process measures on the whole batch -- so that every rank packs the same images as the 1-GPU run."""
    from rotationnormflow_amd import runtime
    world = 2
    mp.spawn(_calib_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m0, m1 = np.load(tmp_path / "ms0.npy"), np.load(tmp_path / "ms1.npy")
    assert np.array_equal(m0, m1) and m0[0] == m0[1] == m0[2]
    feat = torch.from_numpy(synth.features(1001, 16, seed=9))
    feat[:501] *= 4.0
    assert m0[0] == runtime.feature_mean_square(feat)                              # the single-process measurement, bit for bit
    assert abs(np.log2(m0[0]) * 16 - round(np.log2(m0[0]) * 16)) < 1e-9            # quantised to 1/16 binade
    assert 0.9 < m0[0] / float(feat.double().square().mean()) < 1.1


def test_shared_feature_rows_must_divide_the_batch():
    """one helper, one error, for every path that expands shared feature rows."""
    from rotationnormflow_amd import runtime
    f = torch.arange(12.0).reshape(3, 4)
    out = runtime.expand_shared_rows(f, 6, 2)
    assert out.shape == (6, 4) and torch.equal(out[0], out[1]) and torch.equal(out[4], f[2])
    with pytest.raises(ValueError, match="not a multiple of feature_repeat"):
        runtime.expand_shared_rows(f, 7, 2)
