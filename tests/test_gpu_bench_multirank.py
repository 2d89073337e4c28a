"""GPU: the N > 1 control flow of bench.py (torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE, barrier + max-over-ranks timing, one all-reduce
of {sum log p, count}, rank 0 prints ONE JSON line) on a single-GPU box: two ranks share cuda:0 and the collective runs over gloo
(RNF_BENCH_SHARED_GPU=1, a switch the driver never sets).  The real multi-GPU run uses RCCL; only the transport differs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_json_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RNF_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-log2", "15"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 2 * (1 << 15)
    assert rec["value"] > 0 and 10.0 < rec["mean_nll"] < 18.0          # 2 x 2^15 uniform rotations under MF(diag(5,3,1)) after the flow
    assert "cpu_baseline" not in rec                                   # rank-0-at-N=1 only
