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
# C4q / C5q: the reference's own conditional evaluation shape (shared feature rows); C2t / C4t: reference-trained checkpoints
ALL_SECONDARY = ["C1", "C3", "C4", "C5", "C5u", "C4q", "C5q", "C2t", "C4t"]


def _run(cmd, env, timeout=400):
    """One retry when a rank hangs in rendezvous (seen once in ~40 runs on the pool: two ranks on a cold box; the hang dump of
    RNF_BENCH_HANG_DUMP ends the stuck attempt).  A genuine failure fails twice."""
    out = None
    for attempt in range(2):
        try:
            out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        except subprocess.TimeoutExpired as e:             # pragma: no cover
            out = subprocess.CompletedProcess(cmd, 124, e.stdout or "", (e.stderr or "") + "\n[timeout]")
        if out.returncode == 0:
            break
    return out


def _compact(out):
    """The driver-facing record: the LAST stdout line, one JSON object under 3000 bytes (BENCH_r04: a 40 KB line overflowed the driver's
    stdout tail and the headline was lost), the only line that starts with "{"."""
    lines = out.stdout.splitlines()
    braces = [ln for ln in lines if ln.startswith("{")]
    assert len(braces) == 1 and lines[-1] == braces[0], out.stdout[-3000:]
    assert len(braces[0]) < 3000, len(braces[0])
    rec = json.loads(braces[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config"):
        assert k in rec, k
    assert "frac" in rec["roofline"] and rec["roofline"]["bound"] in ("valu", "mfma") and rec["roofline"]["unit"] == "TFLOP/s"
    return rec


def _full(out):
    """The full record: the `BENCH_FULL ` line (also written to bench_full.json)."""
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("BENCH_FULL ")]
    assert len(lines) == 1
    return json.loads(lines[0][len("BENCH_FULL "):])


def test_two_ranks_one_json_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RNF_BENCH_SHARED_GPU="1", RNF_BENCH_HANG_DUMP="150")     # a hung rank dumps its stacks and exits
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-log2", "15"]
    out = _run(cmd, env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = _compact(out)
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["rccl_ranks"] == 2
    assert rec["config"]["global_batch"] == 2 * (1 << 15)
    assert rec["value"] > 0 and 10.0 < rec["mean_nll"] < 18.0          # 2 x 2^15 uniform rotations under MF(diag(5,3,1)) after the flow
    # rank 0 adds a SHORT CPU baseline and the parity block of its shard after the process group is gone (round 5: the N > 1 line has the
    # same keys as the N = 1 line)
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port" and rec["parity"]["mean_abs_err_of_the_mean"] < 1e-5


def test_self_launch_from_gpus_flag():
    """`python bench.py --gpus 2` with no torchrun around it: the script starts its two ranks itself (child torch.distributed.run, before
    the parent touches the GPU) and relays the one JSON line; here both ranks share the one GPU of the box over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RNF_BENCH_SHARED_GPU"] = "1"
    env["RNF_BENCH_HANG_DUMP"] = "150"
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-log2", "15",
                "--no-secondary"], env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = _compact(out)
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["backend"] == "gloo"
    assert rec["config"]["global_batch"] == 2 * (1 << 15)


def test_more_gpus_than_the_box_has_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RNF_BENCH_SHARED_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s) are visible" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    # a rank count that contradicts the launcher is refused too
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env2, cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_single_gpu_line_carries_both_arithmetics_and_every_config_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch-log2", "16", "--no-cpu-baseline",
                          "--no-configs", "--no-pmc"], cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    short, rec = _compact(out), _full(out)
    assert rec["roofline"]["bound"] == "valu" and "frac" in rec["roofline"] and rec["secondary"]["roofline"]["bound"] == "mfma"
    assert rec["secondary"]["value"] > 0 and abs(rec["secondary"]["mean_nll"] - rec["mean_nll"]) < 1e-4
    assert rec["value_fp32_exact"] == rec["secondary"]["value"] and "configs" not in rec
    assert abs(short["value"] - rec["value"]) <= 1e-4 * rec["value"] and abs(short["value_fp32_exact"] - rec["value_fp32_exact"]) <= 1e-4 * rec["value"]
    # round 6: the strict arithmetic (bf16x3) on the same line: faster than the exact fp32-input MFMA, same statistic, its own roofline
    st = rec["strict"]
    assert rec["value_strict"] == st["value"] > rec["value_fp32_exact"] and abs(st["mean_nll"] - rec["mean_nll"]) < 1e-4
    assert st["roofline"]["bound"] == "mfma" and 0.0 < st["roofline"]["frac"] < st["roofline"]["frac_executed"] < 1.0
    assert abs(short["value_strict"] - rec["value_strict"]) <= 1e-4 * rec["value_strict"] and short["strict"]["dtype"] == "bf16x3"
    assert short["full_record"] == "bench_full.json"
    with open(os.path.join(ROOT, "bench_full.json")) as fh:
        assert json.load(fh)["value"] == rec["value"]
    for cfg in ("C1", "C4", "C5", "C5u", "C4q", "C5q", "C4t"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "1", "--warmup", "1", "--batch-log2", "14",
                              "--no-cpu-baseline", "--no-secondary", "--no-pmc"], cwd=ROOT, capture_output=True, text=True, timeout=400)
        assert out.returncode == 0, (cfg, out.stderr[-2000:])
        rec = _compact(out)
        assert rec["value"] > 0 and rec["config"]["workload"].startswith(cfg)


def test_rccl_leg_executes_under_a_launcher_on_one_gpu():
    """torchrun --nproc-per-node 1: the rank initialises the nccl (= RCCL) process group on its GPU and the barrier and the all-reduce of
    {sum log p, count} go through RCCL (one-rank communicator) -- the collective leg of the N-GPU run, executed on the hardware at hand."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RNF_BENCH_SHARED_GPU")}
    env["RNF_BENCH_HANG_DUMP"] = "200"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch-log2", "16", "--no-cpu-baseline", "--no-secondary"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = _compact(out)
    assert rec["backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["n_gpus"] == 1 and rec["value"] > 0


def test_c3_strong_scaling_splits_one_global_batch():
    """BASELINE configs[2] ("cone", 2^22 rotations over the ranks): --config C3 is STRONG scaling -- the same global batch whatever N, rank r
    takes rows [r N / G, (r + 1) N / G), so the mean NLL of the two-rank run (shared-GPU rig) equals the one-rank run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RNF_BENCH_HANG_DUMP"] = "200"
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C3", "--steps", "2", "--warmup", "1", "--batch-log2", "17", "--no-secondary",
            "--no-cpu-baseline", "--no-pmc"]
    one = _run(base + ["--gpus", "1"], env)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = _full(one)
    two = _run(base + ["--gpus", "2"], dict(env, RNF_BENCH_SHARED_GPU="1"))
    assert two.returncode == 0, two.stderr[-2000:]
    r2 = _full(two)
    assert _compact(one)["scaling"] == _compact(two)["scaling"] == "strong"
    for r, g in ((r1, 1), (r2, 2)):
        assert r["scaling"] == "strong" and r["n_gpus"] == g and r["rccl_ranks"] == g
        assert r["config"]["global_batch"] == 1 << 17 and r["config"]["rotations_per_gpu"] == (1 << 17) // g
        assert r["config"]["workload"].startswith("C3")
    # two shards of 2^16 rows run the 8-wave kernel, the single 2^17 launch the 16-wave one of the SAME lean family (round 4): the same rows
    # bit for bit, so the two reduced means differ only by the order of the fp64 additions
    assert abs(r1["mean_nll"] - r2["mean_nll"]) < 1e-12 * max(1.0, abs(r1["mean_nll"]))


def test_default_line_has_the_reference_noise_beside_its_parity():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch-log2", "16", "--no-configs", "--no-pmc"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    short, rec = _compact(out), _full(out)
    assert short["cpu_baseline"]["value"] > 0 and short["cpu_baseline"]["cores"] >= 1 and short["parity"]["reference_fp32_max"] > 0
    p = rec["parity"]
    assert p["mean_abs_err_of_the_mean"] < 1e-5 and p["max_abs_err"] <= 4 * p["reference_fp32"]["max_abs_err"] + 2e-5
    assert rec["secondary"]["parity"]["mean_abs_err_of_the_mean"] < 1e-5
    assert rec["cpu_baseline"]["kind"] == "port" and rec["vs_baseline"] is None and rec["vs_cpu_baseline"] > 100


def test_full_line_carries_every_config_with_live_counters():
    """The default invocation (here at 2^15 rotations so that it takes seconds): ONE JSON line with the C2 headline, the exact-fp32 co-headline
    and a `configs` object for C1, C3, C4, C5, C5u -- each with both arithmetics, a parity block, a CPU baseline and a roofline whose HBM
    traffic and VALU / matrix-pipe fractions come from rocprofv3 --pmc passes run by this very process (children, before it touches the GPU)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch-log2", "15"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    short, rec = _compact(out), _full(out)
    assert rec["config"]["workload"].startswith("C2") and rec["value"] > 0 and rec["value_fp32_exact"] > 0
    assert "error" not in rec["pmc"], rec["pmc"]
    # the strict leg with its own live counters and parity block (round 6)
    sr = rec["strict"]["roofline"]
    assert rec["value_strict"] > 0 and sr["traffic"] > 0 and 0.0 < sr["matrix_pipe_frac"] <= 1.0 and "bf16" in sr["kernel"] or "2," in sr["kernel"]
    assert rec["strict"]["parity"]["max_abs_err"] < 5e-4 and short["strict"]["parity_max"] == pytest.approx(rec["strict"]["parity"]["max_abs_err"], rel=1e-3)
    assert sorted(rec["configs"]) == sorted(ALL_SECONDARY + ["train"])
    # the compact line: the headline with its live roofline and CPU baseline, and one small entry per other workload
    assert short["roofline"]["traffic"] > 0 and 0.0 < short["roofline"]["valu_issue_frac"] <= 1.0 and short["cpu_baseline"]["value"] > 0
    assert sorted(short["configs"]) == sorted(rec["configs"])
    for name in ALL_SECONDARY:
        assert short["configs"][name]["value"] > 0 and "parity_max" in short["configs"][name], name
        if name not in ("C3", "C2t", "C4t"):                      # (bench.THIN_CONFIGS: value / time / parity only in the compact line)
            assert short["configs"][name]["traffic_x"] > 0 and 0.0 < short["configs"][name]["valu_issue_frac"] <= 1.0, name
    assert short["configs"]["train"]["ms_per_iteration"] > 0
    for name in ("C2t", "C4t"):                                   # reference-trained weights: both directions timed, the range guard quiet
        assert rec["configs"][name]["weights"].startswith("trained_c") and rec["configs"][name]["ms_per_step_inverse"] > 0
        assert rec["configs"][name]["fallback_fired"] is False and short["configs"][name]["ms_per_step_inverse"] > 0
    tr = rec["configs"].pop("train")
    assert 0.0 < tr["ms_per_iteration"] < tr["ms_per_iteration_per_tensor_parameters"] and tr["ms_per_iteration_hip_graph"] > 0.0
    for name, c in [("C2", rec)] + sorted(rec["configs"].items()):
        r = c["roofline"]
        assert r["traffic"] is not None and r["traffic"] > 0, (name, r.get("traffic_source"))
        assert 0.0 < r["valu_issue_frac"] <= 1.0 and 0.0 <= r["matrix_pipe_frac"] <= 1.0, name
        assert r["kernels"] and "flow_stack_kernel" in r["kernel"] or "featproj" in r["kernel"], name
        if name not in ("C2t", "C4t"):                            # (the trained-weights entries run the default arithmetic only)
            assert c["secondary"]["roofline"]["traffic"] is not None, name
        assert c["parity"]["mean_abs_err_of_the_mean"] < 2e-5 and "reference_fp32" in c["parity"], name
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0, name
    # the conditional configs' steps are made of a projection pre-pass and the stack kernel
    names = " ".join(k["name"] for k in rec["configs"]["C4"]["roofline"]["kernels"])
    assert "featproj" in names and "flow_stack_kernel" in names


def _self_launch(args, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RNF_BENCH_SHARED_GPU="1", RNF_BENCH_HANG_DUMP="400", OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    return _run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env, timeout=timeout)


def test_eight_ranks_weak_and_strong_equal_the_single_rank_statistic():
    """world 8 on the shared-GPU rig (the driver's N = 8 launch shape: `bench.py --gpus 8`, ranks started by the script itself on a free
    port, one JSON line from rank 0).  C2, weak: 8 x 2^14 rotations.  C3, strong: ONE global batch split 8 ways (`rotations_per_gpu` =
    global / 8) whose reduced mean NLL equals the one-rank evaluation of the same batch to fp64 addition order -- every shard reproduces its
    rows bit for bit whatever launch shape its size selects (tests/test_gpu_scale_properties.py)."""
    weak = _self_launch(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch-log2", "14", "--no-secondary"])
    assert weak.returncode == 0, weak.stderr[-2000:]
    r = _compact(weak)
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["scaling"] == "weak" and r["config"]["global_batch"] == 8 << 14
    assert r["config"]["workload"].startswith("C2") and "configs" not in r and r["cpu_baseline"]["value"] > 0
    base = ["--config", "C3", "--steps", "2", "--warmup", "1", "--batch-log2", "17", "--no-secondary", "--no-cpu-baseline", "--no-pmc"]
    one = _self_launch(base + ["--gpus", "1"], {"RNF_BENCH_SHARED_GPU": "0"})
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = _full(one)
    eight = _self_launch(base + ["--gpus", "8"])
    assert eight.returncode == 0, eight.stderr[-2000:]
    r8 = _full(eight)
    assert _compact(eight)["rccl_ranks"] == 8
    assert r8["scaling"] == "strong" and r8["rccl_ranks"] == 8 and r8["config"]["global_batch"] == 1 << 17
    assert r8["config"]["rotations_per_gpu"] == (1 << 17) // 8 and r1["config"]["rotations_per_gpu"] == 1 << 17
    assert abs(r1["mean_nll"] - r8["mean_nll"]) <= 1e-12 * abs(r1["mean_nll"])
    # the STRONG-scaling line as the driver would launch it (CPU baseline and counters on): every contract key, `roofline`,
    # `cpu_baseline` (its sample as numbers, "cores of threads_available") and the rank count read back from the communicator, under 3 KB
    strong = _self_launch(["--config", "C3", "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch-log2", "17", "--no-secondary"])
    assert strong.returncode == 0, strong.stderr[-2000:]
    c = _compact(strong)
    assert c["scaling"] == "strong" and c["n_gpus"] == 8 and c["rccl_ranks"] == 8 and c["config"]["workload"].startswith("C3")
    cb = c["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["n"] > 0 and cb["best_s"] > 0 and 1 <= cb["cores"] <= cb["threads_available"]
    assert c["roofline"]["frac"] > 0 and abs(c["mean_nll"] - r1["mean_nll"]) <= 1e-12 * abs(r1["mean_nll"])


def test_a_dying_rank_ends_the_run_with_an_error_not_a_hang():
    """One of four ranks exits before the rendezvous: the launcher tears the others down and `bench.py --gpus 4` returns non-zero without a
    JSON line, well inside the timeout."""
    import time
    t0 = time.time()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RNF_BENCH_SHARED_GPU="1", RNF_BENCH_DIE_RANK="2", RNF_BENCH_HANG_DUMP="300")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--batch-log2", "12",
                          "--no-secondary"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 400
