#!/usr/bin/env python3
"""Benchmark of the hot path: rotation log-prob evaluations per second + mean NLL.

Workload (BASELINE.json configs[1], "fisher24"): 24-layer MobiusAffine flow (48 layers: [Moebius, Uncondition16Trans] x 24,
K = 64 segments) + matrix-Fisher base, forward log_prob only, 2^20 uniform-SO(3) rotations per GPU, fp32.
A "step" = one fused density evaluation of the whole per-GPU batch (inputs resident in HBM) + the mean-NLL reduction
(on N > 1 GPUs: one RCCL all-reduce of {sum log p, count}).  Weak scaling: every rank evaluates its own 2^20 shard.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-log2 20] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_ROTATION = 24 * 57_728          # conditioner GEMMs only, exact 2*MAC: 1,385,472
BYTES_PER_ROTATION = 40                  # read 36 B rotation + write 4 B log-prob (log-prob-only form)
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md chip table: f32-input MFMA = f32 VALU peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md chip table: BF16/FP16 MFMA dense
PEAK_HBM_GBPS = 8000.0
CLOCK_SETTLE_LAUNCHES = 16                # untimed launches in front of the warm-up steps (GPU clock ramp), reported in the JSON line


def build_flow(device):
    from rotationnormflow_amd import make_config, synth
    from rotationnormflow_amd.flow.flow import Flow
    cfg = make_config("C2")
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    weights = synth.fill_state_dict(shapes, seed=2024, regime="trained")
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return cfg, weights, fl.to(device).eval()


def host_threads():
    """Threads the baseline may really use: the scheduler affinity of this process (cgroup-limited boxes report far
    more in os.cpu_count()), capped so small ATen ops do not drown in oversubscription."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline(cfg, weights, A, budget_s=12.0):
    """The oracle (torch CPU restatement of the reference path, parity-pinned to it) timed on this box's host cores,
    on a bounded sample sized from a short probe so the default run stays within minutes."""
    from oracle import flow_oracle as orc          # measured as the BASELINE only; never used by the product
    from rotationnormflow_amd import synth
    threads = host_threads()
    torch.set_num_threads(threads)
    probe_n = 1024
    R = synth.uniform_rotations(probe_n, seed=1)
    orc.log_prob(cfg, weights, R[:256], None, A, torch.float32)              # warm-up
    t0 = time.perf_counter()
    orc.log_prob(cfg, weights, R, None, A, torch.float32)
    probe_rate = probe_n / (time.perf_counter() - t0)
    n = int(min(max(probe_rate * budget_s, 2048), 131072)) // 1024 * 1024
    R = synth.uniform_rotations(n, seed=2)
    chunk = 16384                                                            # bound the [N,K,3,3] temporaries
    t0 = time.perf_counter()
    for s in range(0, n, chunk):
        orc.log_prob(cfg, weights, R[s:s + chunk], None, A, torch.float32)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="rotations/s", cores=threads, kind="port",
                sample=f"{n} rotations of the same fisher24 workload (24-layer flow + matrix-Fisher base), fp32, torch-CPU oracle, "
                       f"{threads} threads, one pass of {dt:.1f} s")


def pmc_traffic(precision, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this workload (FETCH_SIZE and WRITE_SIZE
    are collected in their own runs, tools/profile_round.sh; gfx950 correction applied by tools/pmc_summary.py).  None when the run is
    not the profiled configuration."""
    path = os.path.join(ROOT, "profiles", "r1", f"pmc_flow_stack_kernel_{precision}_end.json")
    if n != 1 << 20 or not os.path.exists(path):
        return None, None
    with open(path) as fh:
        d = json.load(fh)
    return d.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)


def pmc_busy(precision, n):
    """VALU / matrix-pipe busy fractions of the dominant kernel from the same committed PMC passes (what actually bounds it)."""
    path = os.path.join(ROOT, "profiles", "r1", f"pmc_flow_stack_kernel_{precision}_end.json")
    if n != 1 << 20 or not os.path.exists(path):
        return None
    with open(path) as fh:
        c = json.load(fh)["counters"]
    try:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        simds = 1024.0
        return {"valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cycles),     # quad-cycles -> cycles
                "matrix_pipe_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * cycles),
                "valu_instructions": c["SQ_INSTS_VALU"], "mfma_instructions": c["SQ_INSTS_MFMA"]}
    except KeyError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # RNF_BENCH_SHARED_GPU=1 (test rig only: a 1-GPU box): every rank uses cuda:0 and the collective runs over gloo, so that the
    # N > 1 control flow of this script can be exercised without N GPUs.  The driver never sets it.
    shared_gpu = os.environ.get("RNF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from rotationnormflow_amd import get_precision, synth
    from rotationnormflow_amd.dist import all_reduce_nll
    from rotationnormflow_amd.utils.fisher import MatrixFisherN

    cfg, weights, fl = build_flow(device)
    n = 1 << args.batch_log2
    A = synth.fisher_A("diag531")
    base = MatrixFisherN(torch.from_numpy(A))
    # rank r evaluates its own shard of the global batch (seeded per rank): no data-path collective
    R = torch.from_numpy(synth.uniform_rotations(n, seed=synth.RD_SEED + rank)).to(device)

    def step():
        res = fl.log_prob(R, base=base)
        return all_reduce_nll(res["sum"]) if distributed else res["sum"]

    with torch.no_grad():
        # the SMU needs ~10 launches (50 ms) of this kernel to settle on its clock (first launch 5.9 ms, steady state 4.8 ms,
        # profiles/r1/rocprofv3_kernel_stats_f16x2_final.csv); these launches are outside both the W warm-up steps and the K timed steps
        for _ in range(CLOCK_SETTLE_LAUNCHES):
            fl.log_prob(R, base=base)
        for _ in range(args.warmup):
            tot = step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()
            res = fl.log_prob(R, base=base)           # the fused stack kernel (+ a 1-block finalize)
            ev[i][1].record()
            tot = all_reduce_nll(res["sum"]) if distributed else res["sum"]
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    tot = tot.cpu().numpy()
    mean_nll = -float(tot[0] / tot[1])

    if rank == 0:
        total_rot = n * world * args.steps
        value = total_rot / elapsed
        achieved_tflops = FLOP_PER_ROTATION * n / (kernel_ms * 1e-3) / 1e12
        precision = get_precision()
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        nw = 8                                                   # waves per workgroup the library picks for this launch (rnf_api.hip run_flow)
        if precision == "f16x2" and os.environ.get("RNF_WIDE") != "0":
            nw = 16 if n > cus * 256 else (4 if n <= cus * 128 else 8)
        traffic, traffic_src = pmc_traffic(precision, n)
        if precision == "f16x2":
            # the conditioner GEMMs run on the fp16 matrix cores (3 fp16 MFMAs with fp32 accumulate per fp32 product-sum),
            # so the MFMA roofline of this kernel is the dense fp16 peak; `achieved` stays the ALGORITHMIC fp32 FLOP rate
            # (executed matrix FLOPs are 3x that).  The kernel is VALU-issue bound (segment math), see DESIGN.md section 3.
            roofline = {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved_tflops / PEAK_F16_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": f"rnf::flow_stack_kernel<0,0,{nw},true,1>", "kernel_ms": kernel_ms,
                        "algorithmic_flop_per_rotation": FLOP_PER_ROTATION, "executed_mfma_tflops": 3 * achieved_tflops,
                        "frac_of_fp32_mfma_peak": achieved_tflops / PEAK_FP32_MFMA_TFLOPS,
                        "note": "fp32 operands split into two fp16 terms (hi + unscaled lo, 2^-24 absolute floor), three fp16 MFMAs into "
                                "one fp32 accumulator; binding limit is VALU issue of the per-segment trig/softplus math, not MFMA "
                                "and not HBM"}
        else:
            roofline = {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": "rnf::flow_stack_kernel<0,0,8,true,0>", "kernel_ms": kernel_ms,
                        "algorithmic_flop_per_rotation": FLOP_PER_ROTATION,
                        "note": "exact fp32-input MFMA; shares the FMA datapath with the VALU segment math on gfx950"}
        busy = pmc_busy(precision, n)
        if busy:
            roofline["pmc"] = busy
        out = {
            "metric": "rotation log_prob evals/s (24-layer MobiusAffine + matrix-Fisher base), mean NLL alongside",
            "value": value, "unit": "rotations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if precision == "fp32" else "f32 (GEMM operands as fp16 hi+lo pairs, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "fisher24: 24-layer MobiusAffine (48 layers, K=64) + matrix-Fisher base A=diag(5,3,1), "
                                   "forward log_prob only, uniform-SO(3) inputs, trained-like random weights",
                       "rotations_per_gpu": n, "global_batch": n * world, "parallelism": f"batch-sharded x{world}, "
                       "one RCCL all-reduce of {sum log p, count} per step" if world > 1 else "single GPU"},
            "mean_nll": mean_nll,
            "clock_settle_launches": CLOCK_SETTLE_LAUNCHES,
            "roofline": roofline,
            "hbm": {"achieved": BYTES_PER_ROTATION * n / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": BYTES_PER_ROTATION * n / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                    "algorithmic_bytes_per_rotation": BYTES_PER_ROTATION},
        }
        if world == 1 and not args.no_cpu_baseline:
            # parity spot check of the benchmarked weights/inputs against the oracle (fp64) on the first 2048 rotations
            from oracle import flow_oracle as orc
            torch.set_num_threads(host_threads())
            sub = R[:2048]
            with torch.no_grad():
                got = fl.log_prob(sub, base=base)["logp"].cpu().double().numpy()
            want, _ = orc.log_prob(cfg, weights, sub.cpu().numpy(), None, A, torch.float64)
            out["parity"] = {"samples": 2048, "mean_nll_abs_err": abs(float(got.mean() - want.numpy().mean())),
                             "max_abs_err": float(np.abs(got - want.numpy()).max())}
            out["cpu_baseline"] = cpu_baseline(cfg, weights, A)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
