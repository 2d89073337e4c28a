#!/usr/bin/env python3
"""Benchmark of the hot path: rotation log-prob evaluations per second + mean NLL.

Default workload (BASELINE.json configs[1], "fisher24" = --config C2): 24-layer MobiusAffine flow (48 layers: [Moebius,
Uncondition16Trans] x 24, K = 64 segments) + matrix-Fisher base, forward log_prob only, 2^20 uniform-SO(3) rotations per GPU, fp32.
A "step" = one fused density evaluation of the whole per-GPU batch (inputs resident in HBM) + the mean-NLL reduction
(on N > 1 GPUs: one RCCL all-reduce of {sum log p, count}).  Weak scaling: every rank evaluates its own 2^20 shard.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C4|C5|C5u] [--batch-log2 B] [--no-cpu-baseline]

--config C3 (BASELINE.json configs[2]) is the STRONG-scaling workload: one global batch of 2^22 rotations split contiguously over the
N ranks (2^19 per GPU at N = 8, all 2^22 on one GPU at N = 1), uniform base, one all-reduce of {sum log p, count}; every other config
is weak scaling (2^20 rotations per GPU).

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process, before this process
touches the GPU) and relays the child's JSON line and exit code; under torchrun (WORLD_SIZE set) it is one rank.
Prints ONE JSON line on rank 0.  The line carries both arithmetics of the conditioner GEMMs: `value` is the default
split-precision fp16-MFMA path ("f16x2"), `secondary` the exact fp32-MFMA path.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md chip table: f32-input MFMA = f32 VALU peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md chip table: BF16/FP16 MFMA dense
PEAK_HBM_GBPS = 8000.0
CLOCK_SETTLE_LAUNCHES = 16                # untimed launches in front of the warm-up steps (GPU clock ramp), reported in the JSON line

# algorithmic GEMM FLOP per rotation (exact 2*MAC of the conditioner MLPs) and algorithmic HBM bytes per
# rotation (inputs read once + what the step writes); preset of rotationnormflow_amd.configs; direction; matrix-Fisher base.
WORKLOADS = {
    "C1": dict(preset="C1", direction="forward", fisher=False, flop=8 * 57_728, bytes=36 + 4,
               text="peak: 8-layer MobiusAffine (16 layers, K=64), forward log_prob, uniform-SO(3) inputs"),
    "C2": dict(preset="C2", direction="forward", fisher=True, flop=24 * 57_728, bytes=36 + 4,
               text="fisher24: 24-layer MobiusAffine (48 layers, K=64) + matrix-Fisher base A=diag(5,3,1), forward log_prob only, "
                    "uniform-SO(3) inputs, trained-like random weights"),
    "C3": dict(preset="C3", direction="forward", fisher=False, flop=24 * 57_728, bytes=36 + 4, strong=True, batch_log2=22,
               text="cone: 24-layer MobiusAffine (48 layers, K=64), forward log_prob, ONE global batch of 2^22 uniform-SO(3) rotations "
                    "split contiguously over the ranks (strong scaling), trained-like random weights"),
    "C4": dict(preset="C4", direction="forward", fisher=False, flop=24 * 90_496 + 59_392, bytes=36 + 4 * 256 + 4,
               text="SYMSOL-I structure: Condition16Trans + 24 Moebius (3+256 inputs) + 23 Uncondition16Trans, F=256 precomputed "
                    "features per rotation, forward log_prob"),
    "C5": dict(preset="C5", direction="inverse", fisher=True, flop=42 * 123_264, bytes=36 + 4 * 512 + 36 + 4,
               text="inverse sampling: draw base samples from MF(diag(5,3,1)) on the device, push them through the inverse of a "
                    "42-layer Moebius-only conditional flow (F=512 precomputed features), return rotation + log-det"),
    "C5u": dict(preset="C5u", direction="inverse", fisher=True, flop=42 * 57_728, bytes=36 + 36 + 4,
                text="inverse sampling, unconditional variant: MF(diag(5,3,1)) samples through the inverse of a 42-layer Moebius-only flow"),
}


def self_launch(args, argv):
    """--gpus N > 1 outside torchrun: run the N ranks as a child torchrun and relay its output.  Nothing here touches the GPU
    (torch.cuda.device_count() does not initialise it on this image)."""
    import torch
    shared = os.environ.get("RNF_BENCH_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not shared:
        print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    return child.returncode


def build_flow(device, preset):
    import torch
    from rotationnormflow_amd import make_config, synth
    from rotationnormflow_amd.flow.flow import Flow
    cfg = make_config(preset)
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


def cpu_baseline(cfg, weights, A, wl, feat_dim, budget_s=12.0):
    """The oracle (torch CPU restatement of the reference path, parity-pinned to it) timed on this box's host cores,
    on a bounded sample sized from a short probe so the default run stays within minutes."""
    import torch
    from oracle import flow_oracle as orc          # measured as the BASELINE only; never used by the product
    from rotationnormflow_amd import synth
    threads = host_threads()
    torch.set_num_threads(threads)

    def run(R, f):
        if wl["direction"] == "forward":
            orc.log_prob(cfg, weights, R, f, A if wl["fisher"] else None, torch.float32)
        else:
            orc.flow_inverse(cfg, weights, R, f, dtype=torch.float32)

    probe_n = 1024 if wl["direction"] == "forward" else 256
    R = synth.uniform_rotations(probe_n, seed=1)
    f = synth.features(probe_n, feat_dim, seed=5) if feat_dim else None
    run(R[:128], None if f is None else f[:128])                              # warm-up
    t0 = time.perf_counter()
    run(R, f)
    probe_rate = probe_n / (time.perf_counter() - t0)
    n = int(min(max(probe_rate * budget_s, 1024), 131072)) // 512 * 512
    R = synth.uniform_rotations(n, seed=2)
    f = synth.features(n, feat_dim, seed=6) if feat_dim else None
    chunk = 16384                                                            # bound the [N,K,3,3] temporaries
    t0 = time.perf_counter()
    for s in range(0, n, chunk):
        run(R[s:s + chunk], None if f is None else f[s:s + chunk])
    dt = time.perf_counter() - t0
    what = "forward log_prob" if wl["direction"] == "forward" else "inverse pass (base sampling not included)"
    return dict(value=n / dt, unit="rotations/s", cores=threads, kind="port",
                sample=f"{n} rotations of the same workload ({wl['preset']} {what}), fp32, torch-CPU oracle, {threads} threads, one pass of {dt:.1f} s")


def committed_pmc(workload, precision, n):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload (tools/profile_round.sh collects
    FETCH_SIZE and WRITE_SIZE each in its own run; tools/pmc_summary.py applies the gfx950 correction).  A summary is only replayed when
    it was taken from the kernel sources this build was made from (`csrc_sha` recorded by pmc_summary.py) and its launch size divides
    this run's batch (the conditional / inverse configs run in chunks of 2^18 rotations: the per-launch bytes of the projection pre-pass
    and of the stack kernel are added and scaled to the step); otherwise (None, reason)."""
    from rotationnormflow_amd.build import source_hash
    names = [f"pmc_{workload}_{precision}.json"] if precision == "f16x2" else []
    names += [f"pmc_{workload}_stack.json", f"pmc_{workload}_featproj.json"] if precision == "f16x2" else []
    found = []
    for rnd in ("r3", "r2", "r1"):
        found = [os.path.join(ROOT, "profiles", rnd, nm) for nm in names if os.path.exists(os.path.join(ROOT, "profiles", rnd, nm))]
        if found:
            break
    if not found:
        return None, "no committed PMC summary for this workload"
    total, primary = 0.0, None
    for path in found:
        with open(path) as fh:
            d = json.load(fh)
        rel = os.path.relpath(path, ROOT)
        if d.get("csrc_sha") != source_hash():
            return None, f"{rel} was collected from other kernel sources (csrc_sha {d.get('csrc_sha')} != {source_hash()}): not replayed"
        per = d.get("rotations_per_launch") or 0
        if per <= 0 or n % per:
            return None, f"{rel} was collected at another launch size: not replayed"
        if "hbm_bytes_per_launch" in d:
            total += d["hbm_bytes_per_launch"] * (n // per)
        if primary is None:
            primary = dict(d, source=rel)
    primary["hbm_bytes_per_launch"] = total
    primary["source"] = ", ".join(os.path.relpath(p, ROOT) for p in found)
    return primary, None


def pmc_fractions(d):
    c = d["counters"]
    try:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        simds = 1024.0
        return {"valu_issue_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cycles),     # quad-cycles -> cycles
                "matrix_pipe_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * cycles),
                "valu_instructions": c["SQ_INSTS_VALU"], "mfma_instructions": c["SQ_INSTS_MFMA"]}
    except KeyError:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-log2", type=int, default=None, help="log2 of the batch: per GPU (weak-scaling configs, default 20) or global (C3, default 22)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the exact-fp32 leg")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args, sys.argv[1:]))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
            sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a launcher started this rank (torchrun sets RANK): the collective leg runs even at world size 1, so that RCCL initialisation, the
    # barrier and the {sum, count} all-reduce execute on a 1-GPU box exactly as they do on N (a 1-rank communicator)
    distributed = world > 1 or "RANK" in os.environ
    if os.environ.get("RNF_BENCH_HANG_DUMP"):                 # diagnostics: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RNF_BENCH_HANG_DUMP"]), exit=True)

    import numpy as np
    import torch

    # RNF_BENCH_SHARED_GPU=1 (test rig only: a 1-GPU box): every rank uses cuda:0 and the collective runs over gloo, so that the
    # N > 1 control flow of this script can be exercised without N GPUs.  The driver never sets it.
    shared_gpu = os.environ.get("RNF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        sys.exit(2)
    backend = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = "gloo" if shared_gpu else "nccl"
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from rotationnormflow_amd import get_precision, set_precision, synth
    from rotationnormflow_amd.dist import all_reduce_nll
    from rotationnormflow_amd.utils.fisher import MatrixFisherN

    wl = WORKLOADS[args.config]
    cfg, weights, fl = build_flow(device, wl["preset"])
    strong = bool(wl.get("strong"))
    batch_log2 = args.batch_log2 if args.batch_log2 is not None else wl.get("batch_log2", 20)
    A = synth.fisher_A("diag531")
    base = MatrixFisherN(torch.from_numpy(A).to(device)) if wl["fisher"] else None
    feat_dim = fl.feature_dim if cfg.condition else 0
    if strong:
        # ONE global batch, the same for every world size; rank r evaluates rows [r N / G, (r + 1) N / G) (dist.shard_bounds, the partition
        # of torch's scatter on dim 0 = the reference's nn.DataParallel, agent.py:22): no data-path collective
        from rotationnormflow_amd.dist import shard_bounds
        n_global = 1 << batch_log2
        lo, hi = shard_bounds(n_global, rank, world)
        R = torch.from_numpy(synth.uniform_rotations(n_global, seed=synth.RD_SEED)[lo:hi].copy()).to(device)
        feat = torch.from_numpy(synth.features(n_global, feat_dim, seed=synth.RD_SEED + 1000)[lo:hi].copy()).to(device) if feat_dim else None
        n = hi - lo
    else:
        # weak scaling: rank r evaluates its own batch (seeded per rank): no data-path collective
        n = 1 << batch_log2
        n_global = n * world
        R = torch.from_numpy(synth.uniform_rotations(n, seed=synth.RD_SEED + rank)).to(device)
        feat = torch.from_numpy(synth.features(n, feat_dim, seed=synth.RD_SEED + 1000 + rank)).to(device) if feat_dim else None

    if wl["direction"] == "forward":
        def evaluate():
            return fl.log_prob(R, feat, base=base)["sum"]
    else:
        def evaluate():                                        # eval.py:327-347: base samples + their log-density, inverse pass, log p = base - ldj
            z = base._sample(n).reshape(-1, 3, 3)
            lp = base._log_prob(z)
            _, ldj = fl.inverse(z, feat)
            lp = (lp - ldj).double()
            return torch.stack((lp.sum(), torch.tensor(float(n), dtype=torch.float64, device=device)))

    def timed(steps, warmup, settle):
        """-> (elapsed s over `steps` (max over ranks), mean HIP-event ms of the library calls, last {sum, count})."""
        with torch.no_grad():
            # the SMU needs ~10 launches (50 ms) of this kernel to settle on its clock; these launches are outside both the W warm-up
            # steps and the K timed steps and are reported in the JSON line
            for _ in range(settle):
                evaluate()
            tot = None
            for _ in range(warmup):
                tot = all_reduce_nll(evaluate()) if distributed else evaluate()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            t0 = time.perf_counter()
            for i in range(steps):
                ev[i][0].record()                              # the library launches on torch's current stream: these events bracket its kernels
                part = evaluate()
                ev[i][1].record()
                tot = all_reduce_nll(part) if distributed else part
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev])), tot.cpu().numpy()

    def roofline_of(precision, kernel_ms):
        achieved = wl["flop"] * n / (kernel_ms * 1e-3) / 1e12
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        if precision == "f16x2":
            nw = 8
            if wl["direction"] == "forward" and os.environ.get("RNF_WIDE") != "0" and not cfg.condition:
                nw = 16 if n > cus * 256 else (4 if n <= cus * 128 else 8)
            elif wl["direction"] == "forward" and os.environ.get("RNF_WIDE") != "0":
                nw = 16 if min(n, 1 << 18) > cus * 256 else 8
            # The conditioner GEMMs run on the fp16 matrix cores (3 fp16 MFMAs with fp32 accumulate per fp32 product-sum), so the
            # matrix roofline of this kernel is the dense fp16 peak; `achieved` stays the ALGORITHMIC fp32 FLOP rate (executed matrix
            # FLOPs are 3x that).  What binds is VALU issue (per-segment softplus / arctangent / reciprocal math): bound = "valu".
            r = {"bound": "valu", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_MFMA_TFLOPS,
                 "kernel": f"rnf::flow_stack_kernel<{1 if wl['direction'] == 'inverse' else 0},...,{nw} waves,f16x2>", "kernel_ms": kernel_ms,
                 "algorithmic_flop_per_rotation": wl["flop"], "executed_mfma_tflops": 3 * achieved,
                 "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                 "note": "frac = algorithmic GEMM FLOP rate / dense fp16 MFMA peak; the kernel is VALU-issue bound (valu_issue_frac is the "
                         "binding fraction, from PMC), neither matrix- nor HBM-bound"}
        else:
            r = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                 "kernel": f"rnf::flow_stack_kernel<{1 if wl['direction'] == 'inverse' else 0},...,8 waves,fp32>", "kernel_ms": kernel_ms,
                 "algorithmic_flop_per_rotation": wl["flop"],
                 "note": "exact fp32-input MFMA; shares the FMA datapath with the VALU segment math on gfx950"}
        pmc, why = committed_pmc(args.config, precision, n)
        if pmc:
            r["traffic"] = pmc.get("hbm_bytes_per_launch")          # HBM bytes of one step (all launches of the step, all its kernels)
            r["traffic_source"] = f"committed profile {pmc['source']} (csrc_sha {pmc['csrc_sha']}, kernel {pmc.get('kernel')})"
            r.update(pmc_fractions(pmc))
        else:
            r["traffic"] = None
            r["traffic_source"] = why
        return r

    primary = get_precision()
    elapsed, kernel_ms, tot = timed(args.steps, args.warmup, CLOCK_SETTLE_LAUNCHES)
    packed = fl._packed(device)
    primary_used = packed.precision                       # "fp32" when a weight left the fp16 range and the packer fell back
    mean_nll = -float(tot[0] / tot[1])
    secondary = None
    if not args.no_secondary and primary_used == "f16x2":
        set_precision("fp32")
        try:
            s_steps = max(2, min(args.steps, 8))
            s_elapsed, s_kernel_ms, s_tot = timed(s_steps, 2, 4)
            secondary = {"dtype": "f32 (exact fp32-input MFMA)", "value": n_global * s_steps / s_elapsed, "unit": "rotations/s",
                         "steps": s_steps, "ms_per_step": s_elapsed / s_steps * 1e3, "mean_nll": -float(s_tot[0] / s_tot[1]),
                         "roofline": roofline_of("fp32", s_kernel_ms)}
        finally:
            set_precision(primary)

    if rank == 0:
        value = n_global * args.steps / elapsed
        out = {
            "metric": "rotation log_prob evals/s (24-layer MobiusAffine + matrix-Fisher base), mean NLL alongside" if args.config == "C2"
                      else f"rotation evals/s, workload {args.config}",
            "value": value, "unit": "rotations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,         # the reference publishes no number for this metric; see vs_cpu_baseline
            "dtype": "f32" if primary_used == "fp32" else "f32 (GEMM operands as fp16 hi+lo pairs, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{args.config} -- {wl['text']}", "rotations_per_gpu": n, "global_batch": n_global,
                       "parallelism": (f"batch-sharded x{world}, one all-reduce of {{sum log p, count}} per step" if world > 1 else "single GPU")},
            "rccl_ranks": world, "backend": backend,
            "mean_nll": mean_nll,
            "clock_settle_launches": CLOCK_SETTLE_LAUNCHES,
            "roofline": roofline_of(primary_used, kernel_ms),
            "hbm": {"achieved": wl["bytes"] * n / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": wl["bytes"] * n / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "algorithmic_bytes_per_rotation": wl["bytes"]},
        }
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            # parity spot check of the benchmarked weights/inputs against the oracle (fp64) on the first rotations, for BOTH arithmetics, with
            # the error of the oracle's own fp32 evaluation (= the reference's arithmetic) of the same rows beside it
            from oracle import flow_oracle as orc
            torch.set_num_threads(host_threads())
            m = 2048 if wl["direction"] == "forward" else 256
            sub = R[:m]
            fsub = None if feat is None else feat[:m]
            sub_np, fsub_np = sub.cpu().numpy(), None if fsub is None else fsub.cpu().numpy()

            def oracle(dtype):
                if wl["direction"] == "forward":
                    return orc.log_prob(cfg, weights, sub_np, fsub_np, A if wl["fisher"] else None, dtype)[0].double().numpy()
                return orc.flow_inverse(cfg, weights, sub_np, fsub_np, dtype=dtype)[1].double().numpy()

            def product():
                with torch.no_grad():
                    if wl["direction"] == "forward":
                        return fl.log_prob(sub, fsub, base=base)["logp"].cpu().double().numpy()
                    return fl.inverse(sub, fsub)[1].cpu().double().numpy()

            def stats(got, want):
                e = np.abs(got - want)
                return {"mean_abs_err_of_the_mean": abs(float(got.mean() - want.mean())), "mean_abs_err": float(e.mean()),
                        "p99_abs_err": float(np.quantile(e, 0.99)), "max_abs_err": float(e.max())}
            want = oracle(torch.float64)
            what = "per-rotation log p (mean = -mean NLL)" if wl["direction"] == "forward" else "per-rotation log-det of the inverse pass"
            out["parity"] = {"samples": m, "quantity": what, "against": "fp64 oracle (pinned to the reference's fp64 run to 1e-11)",
                             **stats(product(), want), "reference_fp32": stats(oracle(torch.float32), want)}
            if secondary:
                set_precision("fp32")
                try:
                    secondary["parity"] = {"samples": m, **stats(product(), want)}
                finally:
                    set_precision(primary)
            out["cpu_baseline"] = cpu_baseline(cfg, weights, A, wl, feat_dim)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
