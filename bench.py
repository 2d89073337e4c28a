#!/usr/bin/env python3
"""Benchmark of the hot path: rotation log-prob evaluations per second + mean NLL.

Headline workload (BASELINE.json configs[1], "fisher24" = C2): 24-layer MobiusAffine flow (48 layers: [Moebius, Uncondition16Trans] x 24,
K = 64 segments) + matrix-Fisher base, forward log_prob only, 2^20 uniform-SO(3) rotations per GPU, fp32.
A "step" = one fused density evaluation of the whole per-GPU batch (inputs resident in HBM) + the mean-NLL reduction (on N > 1 GPUs: one
RCCL all-reduce of {sum log p, count}).  Weak scaling: every rank evaluates its own 2^20 shard.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C4|C5|C5u] [--batch-log2 B]
                    [--no-cpu-baseline] [--no-secondary] [--no-configs] [--no-pmc]

ONE JSON line on rank 0.  At N = 1 with the default workload the line also carries (round 4):
  * `value_fp32_exact` / `ms_per_step_fp32_exact`: the same workload on the exact fp32-input MFMA kernels (the unconditional number; `value`
    is the default split-precision arithmetic -- fp32 operands carried as fp16 hi+lo pairs, fp32 accumulate);
  * `configs`: every other BASELINE.json config on the same clock -- C1, C3 (its 2^22 rotations on this one GPU), C4, C5, C5u -- each with
    value, ms_per_step, both arithmetics, roofline (frac, kernel list, HBM traffic, VALU-issue and matrix-pipe fractions), parity against the
    fp64 oracle with the reference-fp32 arithmetic's own error beside it, and a CPU baseline; plus `configs.train`: one training iteration
    of the reference's recipe, eager with torch's default Adam, as the unedited driver runs it;
  * `roofline.traffic` and the binding fractions MEASURED IN THIS RUN: before this process touches the GPU it runs three short
    `rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --pmc-child ...` children (FETCH_SIZE, WRITE_SIZE and the SQ/GRBM group each in
    its own pass, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024) over the same workloads of the same library.
    If rocprofv3 is unavailable the committed summaries under profiles/ are replayed when their source hash matches (else null + the reason).

--config C3 (BASELINE.json configs[2]) is the STRONG-scaling workload: one global batch of 2^22 rotations split contiguously over the N
ranks (2^19 per GPU at N = 8, all 2^22 on one GPU at N = 1), uniform base, one all-reduce of {sum log p, count}; every other config is weak
scaling (2^20 rotations per GPU).  An explicit --config runs that workload alone (no `configs` object).

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its N ranks itself (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process, before this process touches the GPU) and relays the child's JSON line
and exit code; under torchrun (WORLD_SIZE set) it is one rank.
"""
import argparse
import contextlib
import csv
import glob
import io
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md chip table: f32-input MFMA = f32 VALU peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md chip table: BF16/FP16 MFMA dense
PEAK_HBM_GBPS = 8000.0
CLOCK_SETTLE_LAUNCHES = 16                # untimed launches in front of the warm-up steps (GPU clock ramp), reported in the JSON line
SIMDS = 1024                              # 256 CUs x 4

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
    # The reference's REAL conditional evaluation pattern (agent.py:238-263, eval.py:322-347): one image feature against `number_queries`
    # rotations -- the reference materialises feature.repeat; here the feature rows are SHARED (feature_repeat = Q): the projection runs once
    # per image, the stack kernels read a 64-float record per (layer, image).  2048 images x 512 queries = 2^20 rotations.
    # flop: the per-rotation GEMM work (conditioner with its 3 rotation inputs) + the per-image feature part amortised over Q rotations.
    "C4q": dict(preset="C4", direction="forward", fisher=False, queries=512, flop=24 * 57_728 + (59_392 + 24 * 2 * 256 * 64) // 512,
                bytes=36 + 4 + 4 * 256 // 512,
                text="SYMSOL-I structure as the reference evaluates it: 2048 image features (F=256) x 512 query rotations each (shared feature "
                     "rows), forward log_prob"),
    "C5q": dict(preset="C5", direction="inverse", fisher=True, queries=512, flop=42 * 57_728 + (42 * 2 * 512 * 64) // 512,
                bytes=36 + 36 + 4 + 4 * 512 // 512 + 36 // 512 + 1,
                text="pose estimation as the reference runs it (agent.py:238-283): per image feature (F=512, 2048 images) 512 base samples from "
                     "MF(diag(5,3,1)) through the inverse of the 42-layer conditional flow, log p = base - ldj, arg-max rotation per image"),
    # Weights the REFERENCE's own training produced (tests/golden/make_trained.py: its Flow under torch.optim.Adam) at the headline structures:
    # the VALU-bound forward kernels are data independent, the inverse root finder's pass count is not -- both directions are timed.
    "C2t": dict(preset="C2", direction="forward", fisher=False, flop=24 * 57_728, bytes=36 + 4, weights="tests/golden/trained_c2.pth",
                also_inverse=True, text="the C2 structure with reference-trained weights (trained_c2.pth), forward log_prob and inverse pass"),
    "C4t": dict(preset="C4", direction="forward", fisher=False, flop=24 * 90_496 + 59_392, bytes=36 + 4 * 256 + 4,
                weights="tests/golden/trained_c4.pth", features="tests/golden/trained_c4.npz", also_inverse=True,
                text="the C4 structure with reference-trained weights (trained_c4.pth) and the features it was trained on (one class un-normalised)"),
}
ALL_CONFIGS = ["C2", "C1", "C3", "C4", "C5", "C5u", "C4q", "C5q", "C2t", "C4t"]
PMC_GROUPS = {
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "sq": "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE",
}
PMC_EVALS = 2                             # profiled evaluations per workload and arithmetic in a --pmc-child pass (behind one warm-up)
MARKER = "FillFunctor<short>"             # an int16 fill: a kernel nothing else launches, separates the sections of a --pmc-child pass


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


def build_flow(device, preset, weights_path=None):
    import torch
    from rotationnormflow_amd import make_config, synth
    from rotationnormflow_amd.flow.flow import Flow
    cfg = make_config(preset)
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    if weights_path:                                          # a checkpoint in Agent.save_ckpt's layout (agent.py:139-152)
        from rotationnormflow_amd.harness import load_reference_checkpoint
        weights_path = weights_path if os.path.isabs(weights_path) else os.path.join(ROOT, weights_path)
        sd = load_reference_checkpoint(weights_path)
        weights = {k: v.numpy() for k, v in sd.items()}
        assert {k: tuple(v.shape) for k, v in weights.items()} == shapes, "checkpoint does not fit the workload's flow"
    else:
        weights = synth.fill_state_dict(shapes, seed=2024, regime="trained")
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    fl = fl.to(device).eval()
    if weights_path:                                          # what the checkpoint's sidecar fixes (harness.read_sidecar), if there is one
        from rotationnormflow_amd.harness import read_sidecar
        side = read_sidecar(weights_path)
        if cfg.condition and side.get("feature_mean_square") is not None:
            fl.set_feature_scale(float(side["feature_mean_square"]))
        if side.get("rootfinder_first_order") is not None:
            fl.set_rootfinder_order(int(side["rootfinder_first_order"]))
    return cfg, weights, fl


def threads_available():
    """Hardware threads this process may run on (scheduler affinity: cgroup-limited boxes report far more in os.cpu_count())."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def host_threads():
    """Threads the baseline uses: the available ones, capped at 32 so the oracle's small ATen ops do not drown in oversubscription (the line
    says "32 of N": `cores` and `threads_available`)."""
    return max(1, min(threads_available(), 32))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, weights, A, wl, feat_dim, sizes, budget_s, queries=None):
    """the oracle (torch-CPU restatement of the reference path, parity-pinned to it) on this box's host cores, fp32,
    no_grad, the same seeded recipe as the GPU run, at each N of ``sizes``: one warm-up, then best of 3 -- fewer repetitions (never fewer
    than one) where three would overrun ``budget_s`` seconds, which `sample` then says.  `value` is the best rate over the sizes."""
    import numpy as np
    import torch
    from oracle import flow_oracle as orc          # measured as the BASELINE only; never used by the product
    from rotationnormflow_amd import synth
    threads = host_threads()
    torch.set_num_threads(threads)

    def run(R, f):
        chunk = 16384                                                        # bound the [N,K,3,3] temporaries
        t0 = time.perf_counter()
        for s in range(0, R.shape[0], chunk):
            if wl["direction"] == "forward":
                orc.log_prob(cfg, weights, R[s:s + chunk], None if f is None else f[s:s + chunk], A if wl["fisher"] else None, torch.float32)
            else:
                orc.flow_inverse(cfg, weights, R[s:s + chunk], None if f is None else f[s:s + chunk], dtype=torch.float32)
        return time.perf_counter() - t0

    runs, best_rate, spent = [], 0.0, 0.0
    for n in sizes:
        R = synth.uniform_rotations(n, seed=2)
        f = synth.features(n, feat_dim, seed=6) if feat_dim else None
        if f is not None and queries:                        # what the reference does: the image feature repeated over its queries (agent.py:240-244)
            f = np.repeat(f[: max(1, n // queries)], queries, axis=0)[:n]
        m = min(n, 512)
        t_warm = run(R[:m], None if f is None else f[:m])                    # warm-up (thread pool, allocator)
        est = n / best_rate if best_rate > 0 else t_warm * n / m            # (the rate of the previous, smaller size predicts this one)
        reps = 3 if spent + 3 * est <= budget_s else max(1, int((budget_s - spent) / max(est, 1e-9)))
        reps = min(reps, 3)
        times = [run(R, f) for _ in range(reps)]
        spent += sum(times) + t_warm
        runs.append({"n": n, "best_s": min(times), "repetitions": reps, "rotations_per_s": n / min(times)})
        best_rate = max(best_rate, n / min(times))
    what = "forward log_prob" if wl["direction"] == "forward" else "inverse pass (base sampling not included)"
    best = max(runs, key=lambda r: r["rotations_per_s"])
    return dict(value=best_rate, unit="rotations/s", cores=threads, threads_available=threads_available(), kind="port", cpu=cpu_model(), runs=runs,
                n=best["n"], best_s=best["best_s"], repetitions=best["repetitions"],
                sample=f"{wl['preset']} {what}, fp32, torch-CPU oracle, {threads} of {threads_available()} threads; " +
                       "; ".join(f"N={r['n']}: best of {r['repetitions']} = {r['best_s']:.2f} s" for r in runs))


# ---------------------------------------------------------------------------------------------------------------------------------------------
# one workload on the device
# ---------------------------------------------------------------------------------------------------------------------------------------------
class Workload:
    """Flow, inputs (resident in HBM) and the step of one BASELINE config on ``device``."""

    def __init__(self, name, device, batch_log2=None, rank=0, world=1, share=None, weights_path=None):
        import numpy as np
        import torch
        from rotationnormflow_amd import synth
        from rotationnormflow_amd.utils.fisher import MatrixFisherN
        self.name, self.device, self.wl = name, device, WORKLOADS[name]
        wl = self.wl
        self.weights_path = weights_path or wl.get("weights")
        if (share is not None and share.wl["preset"] in ("C2", "C3") and wl["preset"] in ("C2", "C3") and not self.weights_path
                and not share.weights_path):
            self.cfg, self.weights, self.fl = share.cfg, share.weights, share.fl      # C2 and C3: the same flow and weights
        else:
            self.cfg, self.weights, self.fl = build_flow(device, wl["preset"], self.weights_path)
        self.strong = bool(wl.get("strong"))
        log2n = batch_log2 if batch_log2 is not None else wl.get("batch_log2", 20)
        self.A = synth.fisher_A("diag531")
        self.base = MatrixFisherN(torch.from_numpy(self.A).to(device)) if wl["fisher"] else None
        self.feat_dim = self.fl.feature_dim if self.cfg.condition else 0
        gen = torch.Generator(device=device)
        if self.strong:
            # ONE global batch, the same for every world size; rank r evaluates rows [r N / G, (r + 1) N / G) (dist.shard_bounds, the partition
            # of torch's scatter on dim 0 = the reference's nn.DataParallel, agent.py:22): no data-path collective
            from rotationnormflow_amd.dist import shard_bounds
            self.n_global = 1 << log2n
            lo, hi = shard_bounds(self.n_global, rank, world)
            self.R = torch.from_numpy(synth.uniform_rotations(self.n_global, seed=synth.RD_SEED)[lo:hi].copy()).to(device)
            self.n = hi - lo
            fseed, skip = synth.RD_SEED + 1000, lo
        else:
            # weak scaling: rank r evaluates its own batch (seeded per rank): no data-path collective
            self.n = 1 << log2n
            self.n_global = self.n * world
            self.R = torch.from_numpy(synth.uniform_rotations(self.n, seed=synth.RD_SEED + rank)).to(device)
            fseed, skip = synth.RD_SEED + 1000 + rank, 0
        self.feat = None
        self.queries = wl.get("queries")                      # Q rotations share one feature row (feature_repeat)
        if self.queries and (self.n % self.queries or skip % self.queries):
            raise SystemExit(f"bench.py: workload {name} needs a batch that is a multiple of {self.queries} rotations per rank")
        if self.feat_dim:
            # precomputed features ~ N(0, 1) [N, F] drawn on the device (2 GB for C5: a host generator would take longer than the bench)
            gen.manual_seed(fseed)
            per = self.queries or 1
            if wl.get("features"):                             # the feature rows a trained checkpoint was fitted on, repeated over the batch
                rows = torch.from_numpy(np.load(os.path.join(ROOT, wl["features"]))["test_feat"]).to(device)
                self.feat = rows[(torch.arange(self.n, device=device) + skip) % rows.shape[0]].contiguous()
            else:
                full = torch.randn(((skip + self.n) // per, self.feat_dim), generator=gen, device=device, dtype=torch.float32)
                self.feat = full[skip // per:].contiguous()

    def evaluate(self):
        import torch
        if self.wl["direction"] == "forward":
            return self.fl.log_prob(self.R, self.feat, base=self.base, feature_repeat=self.queries)["sum"]
        # eval.py:327-347: base samples + their log-density, inverse pass, log p = base - ldj
        z = self.base._sample(self.n).reshape(-1, 3, 3)
        lp = self.base._log_prob(z)
        rot, ldj = self.fl.inverse(z, self.feat, feature_repeat=self.queries)
        lp = lp - ldj
        if self.queries:                                      # agent.py:264-283: the most likely of each image's samples is the pose estimate
            best = lp.view(-1, self.queries).argmax(dim=1)
            self.estimate = rot.view(-1, self.queries, 3, 3)[torch.arange(best.numel(), device=best.device), best]
        lp = lp.double()
        return torch.stack((lp.sum(), torch.tensor(float(self.n), dtype=torch.float64, device=self.device)))

    def timed(self, steps, warmup, settle, dist=None):
        """-> (elapsed s over `steps` (max over ranks), mean HIP-event ms of the library calls, last {sum, count})."""
        import numpy as np
        import torch
        from rotationnormflow_amd.dist import all_reduce_nll
        distributed = dist is not None
        with torch.no_grad():
            # the SMU needs ~10 launches (50 ms) of this kernel to settle on its clock; these launches are outside both the W warm-up
            # steps and the K timed steps and are reported in the JSON line
            for _ in range(settle):
                self.evaluate()
            tot = None
            for _ in range(warmup):
                tot = all_reduce_nll(self.evaluate()) if distributed else self.evaluate()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            t0 = time.perf_counter()
            for i in range(steps):
                ev[i][0].record()                              # the library launches on torch's current stream: these events bracket its kernels
                part = self.evaluate()
                ev[i][1].record()
                tot = all_reduce_nll(part) if distributed else part
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev])), tot.cpu().numpy()

    def parity(self, rows):
        """Error statistics of `rows` rotations of THIS run's inputs and weights against the fp64 oracle, for the arithmetic in force, with
        the error of the oracle's own fp32 evaluation (= the reference's arithmetic) of the same rows beside it."""
        import numpy as np
        import torch
        from oracle import flow_oracle as orc
        torch.set_num_threads(host_threads())
        wl = self.wl
        q = self.queries
        if q:
            rows = max(q, rows // q * q)
        sub = self.R[:rows]
        fsub = None if self.feat is None else self.feat[:rows // (q or 1)]
        sub_np = sub.cpu().numpy()
        fsub_np = None if fsub is None else (fsub.repeat_interleave(q, dim=0) if q else fsub).cpu().numpy()     # the reference's feature.repeat

        def oracle(dtype):
            if wl["direction"] == "forward":
                return orc.log_prob(self.cfg, self.weights, sub_np, fsub_np, self.A if wl["fisher"] else None, dtype)[0].double().numpy()
            return orc.flow_inverse(self.cfg, self.weights, sub_np, fsub_np, dtype=dtype)[1].double().numpy()

        def product():
            with torch.no_grad():
                if wl["direction"] == "forward":
                    return self.fl.log_prob(sub, fsub, base=self.base, feature_repeat=q)["logp"].cpu().double().numpy()
                return self.fl.inverse(sub, fsub, feature_repeat=q)[1].cpu().double().numpy()

        def stats(got, want):
            e = np.abs(got - want)
            return {"mean_abs_err_of_the_mean": abs(float(got.mean() - want.mean())), "mean_abs_err": float(e.mean()),
                    "p99_abs_err": float(np.quantile(e, 0.99)), "max_abs_err": float(e.max())}
        want = oracle(torch.float64)
        what = "per-rotation log p (mean = -mean NLL)" if wl["direction"] == "forward" else "per-rotation log-det of the inverse pass"
        return {"samples": rows, "quantity": what, "against": "fp64 oracle (pinned to the reference's fp64 run to 1e-11)"}, stats, product, want, oracle


def roofline_of(w, precision, kernel_ms, pmc):
    """`achieved` = ALGORITHMIC GEMM FLOP of one step / the step's HIP-event time.  f16x2: the conditioner GEMMs run on the fp16 matrix cores
    (3 fp16 MFMAs with fp32 accumulate per fp32 product-sum), so the matrix roofline is the dense fp16 peak while `achieved` stays the
    algorithmic fp32 FLOP rate (executed matrix FLOPs are 3x that); what binds is VALU issue (per-segment softplus / arctangent /
    reciprocal math): bound = "valu", with `valu_issue_frac` (PMC, this run) as the binding fraction.  fp32: exact fp32-input MFMA."""
    wl = w.wl
    achieved = wl["flop"] * w.n / (kernel_ms * 1e-3) / 1e12
    if precision == "f16x2":
        r = {"bound": "valu", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_MFMA_TFLOPS,
             "kernel_ms": kernel_ms, "algorithmic_flop_per_rotation": wl["flop"], "executed_mfma_tflops": 3 * achieved,
             "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
             "note": "frac = algorithmic GEMM FLOP rate / dense fp16 MFMA peak; the stack kernel is VALU-issue bound (valu_issue_frac), neither "
                     "matrix- nor HBM-bound"}
    elif precision == "bf16x3":
        r = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_MFMA_TFLOPS,
             "kernel_ms": kernel_ms, "algorithmic_flop_per_rotation": wl["flop"], "executed_mfma_tflops": 6 * achieved,
             "frac_executed": 6 * achieved / PEAK_F16_MFMA_TFLOPS,
             "note": "frac = algorithmic GEMM FLOP rate / dense bf16 MFMA peak (= the fp16 peak); six bf16 MFMAs per fp32 product-sum are "
                     "executed (frac_executed); matrix time and VALU time add on a SIMD (matrix_pipe_frac + valu_issue_frac ~ 1)"}
    else:
        r = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
             "kernel_ms": kernel_ms, "algorithmic_flop_per_rotation": wl["flop"],
             "note": "exact fp32-input MFMA; shares the FMA datapath with the VALU segment math on gfx950"}
    r["traffic_algorithmic"] = wl["bytes"] * w.n
    sec = (pmc or {}).get("sections", {}).get(f"{w.name}:{precision}")
    if sec and sec.get("rotations") == w.n:
        r.update(sec["roofline"])
        r["traffic_source"] = pmc["source"]
    else:
        committed, why = committed_pmc(w.name, precision, w.n)
        if committed:
            r.update(committed)
        else:
            r["traffic"] = None
            r["traffic_source"] = (pmc or {}).get("error") or why
    if "kernel" not in r:
        r["kernel"] = f"rnf::flow_stack_kernel<{1 if wl['direction'] == 'inverse' else 0},...>"
    return r


def committed_pmc(workload, precision, n):
    """Fallback when the live PMC passes could not run: counters of the committed rocprofv3 passes of THIS workload (profiles/r*/pmc_live.json,
    written by `bench.py --save-pmc`), replayed only when they were taken from the kernel sources this build was made from."""
    from rotationnormflow_amd.build import source_hash
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_live.json")), reverse=True):
        with open(path) as fh:
            d = json.load(fh)
        rel = os.path.relpath(path, ROOT)
        if d.get("csrc_sha") != source_hash():
            return None, f"{rel} was collected from other kernel sources (csrc_sha {d.get('csrc_sha')} != {source_hash()}): not replayed"
        sec = d.get("sections", {}).get(f"{workload}:{precision}")
        if not sec or sec.get("rotations") != n:
            return None, f"{rel} holds no pass of this workload at this batch size"
        return dict(sec["roofline"], traffic_source=f"committed {rel} (csrc_sha {d['csrc_sha']})"), None
    return None, "rocprofv3 passes unavailable and no committed PMC summary"


# ---------------------------------------------------------------------------------------------------------------------------------------------
# live PMC passes: children of this process under rocprofv3, run BEFORE this process touches the GPU
# ---------------------------------------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    """`rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --pmc-child --pmc-configs C2,C1,...`: every workload once per arithmetic,
    sections separated by a marker kernel; prints the section table (JSON) that pmc_parse joins with the profiler's CSV."""
    import torch
    from rotationnormflow_amd import set_precision
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    marker = torch.empty(64, dtype=torch.int16, device=device)       # (empty, not zeros: a zeros() would itself launch the marker kernel)
    sections = []
    prev = None
    evals, settle = (args.pmc_evals or PMC_EVALS), (args.pmc_settle or 0)
    for name in args.pmc_configs.split(","):
        w = Workload(name, device, args.batch_log2, share=prev)
        prev = w
        for precision in ("f16x2", "fp32", "bf16x3"):
            set_precision(precision)
            with torch.no_grad():
                w.evaluate()                                    # warm-up of this arithmetic: packs, allocates; profiled too, then dropped by position
                for _ in range(settle):                         # (kernel-trace pass: let the clock settle as the timed region of the bench does)
                    w.evaluate()
                torch.cuda.synchronize()
                marker.fill_(1)
                for _ in range(evals):
                    w.evaluate()
                torch.cuda.synchronize()
                marker.fill_(2)
            sections.append({"key": f"{name}:{precision}", "rotations": w.n, "evals": evals,
                             "packed_precision": w.fl._packed(device, w.feat).precision})
        set_precision("f16x2")
        w.R = w.feat = None
        torch.cuda.empty_cache()
    print("PMC_SECTIONS " + json.dumps(sections), flush=True)


def pmc_parse(csv_paths, sections):
    """Join one pass's counter_collection.csv with the child's section table: dispatches in order, an int16 fill opens and closes every
    section.  -> {section key: {kernel name: {calls, ns, counters{name: sum}}}}"""
    disp = {}
    for path in csv_paths:
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                           "vgpr": int(r["VGPR_Count"]), "wg": int(r["Workgroup_Size"]), "c": defaultdict(float)})
                d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    out, idx, inside = {}, 0, False
    for k in sorted(disp):
        d = disp[k]
        if MARKER in d["kernel"]:
            if inside:
                idx += 1
            inside = not inside
            continue
        if not inside or idx >= len(sections) or "rnf::" not in d["kernel"]:
            continue
        sec = out.setdefault(sections[idx]["key"], {})
        e = sec.setdefault(d["kernel"], {"calls": 0, "ns": 0, "vgpr": d["vgpr"], "wg": d["wg"], "counters": defaultdict(float)})
        e["calls"] += 1
        e["ns"] += d["ns"]
        for cn, cv in d["c"].items():
            e["counters"][cn] += cv
    return out


def short_kernel(name):
    name = name.replace("void ", "").replace("rnf::", "")
    return name.split("(")[0]


def pmc_sections_summary(passes, sections, trace=None):
    """-> {section key: {rotations, roofline{traffic, traffic_fetch, traffic_write, kernel, kernels[...], valu_issue_frac, ...}}} per STEP."""
    out = {}
    for s in sections:
        key, evals = s["key"], s["evals"]
        kernels = {}
        for kname, e in (trace or {}).get(key, {}).items():      # kernel-trace pass without counters (same box, same process shape)
            kernels.setdefault(kname, {"name": short_kernel(kname), "calls_per_step": e["calls"] / evals})["ms_per_call_trace"] = e["ns"] / e["calls"] * 1e-6
        for grp, per in passes.items():
            for kname, e in per.get(key, {}).items():
                k = kernels.setdefault(kname, {"name": short_kernel(kname), "calls_per_step": e["calls"] / evals})
                k.update(vgpr=e["vgpr"], workgroup=e["wg"])
                k.setdefault("ms_per_call_under_pmc", {})[grp] = e["ns"] / e["calls"] * 1e-6
                c = e["counters"]
                if "FETCH_SIZE" in c:
                    k["fetch_bytes_per_step"] = c["FETCH_SIZE"] * 1024 * 2 / evals        # gfx950: 64 B tallied per 128-B request (MI355X_MICROARCH.md)
                if "WRITE_SIZE" in c:
                    k["write_bytes_per_step"] = c["WRITE_SIZE"] * 1024 / evals
                if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
                    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                                  # the counter sums the 8 XCDs
                    k["valu_issue_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cycles)      # quad-cycles -> cycles
                    k["matrix_pipe_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * cycles)
                    k["valu_instructions_per_step"] = c.get("SQ_INSTS_VALU", 0.0) / evals
                    k["mfma_instructions_per_step"] = c.get("SQ_INSTS_MFMA", 0.0) / evals
                    wc = c.get("SQ_WAVE_CYCLES", 0.0)
                    if wc > 0:
                        k["wave_wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / wc
                        k["wave_wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
                        k["wave_active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
                    k["gpu_cycles_per_step"] = cycles / evals
        if not kernels:
            continue
        kernels = {n: k for n, k in kernels.items() if "ms_per_call_under_pmc" in k}
        if not kernels:
            continue
        klist = sorted(kernels.values(), key=lambda k: -k["calls_per_step"] * max(k["ms_per_call_under_pmc"].values()))
        for k in klist:
            k["ms_per_call_under_pmc"] = min(k["ms_per_call_under_pmc"].values())
        klist = [k for k in klist if k["ms_per_call_under_pmc"] * k["calls_per_step"] > 0.002]          # drop the early-exit guard re-runs and finalizers
        dom = klist[0]
        fetch = sum(k.get("fetch_bytes_per_step", 0.0) for k in kernels.values())
        write = sum(k.get("write_bytes_per_step", 0.0) for k in kernels.values())
        roof = {"traffic": fetch + write if ("fetch" in passes and "write" in passes) else None, "traffic_fetch": fetch, "traffic_write": write,
                "kernel": dom["name"], "kernels": klist}
        if "ms_per_call_trace" in dom:                           # rocprofv3 --kernel-trace average of the dominant kernel on THIS box, and of the step
            roof["kernel_ms_trace"] = dom["ms_per_call_trace"]
            roof["step_ms_trace"] = sum(k.get("ms_per_call_trace", 0.0) * k["calls_per_step"] for k in klist)
        tot_cyc = sum(k.get("gpu_cycles_per_step", 0.0) for k in klist)
        if tot_cyc > 0:                                          # step-level fractions: cycle-weighted over the step's kernels
            roof["valu_issue_frac"] = sum(k.get("valu_issue_frac", 0.0) * k.get("gpu_cycles_per_step", 0.0) for k in klist) / tot_cyc
            roof["matrix_pipe_frac"] = sum(k.get("matrix_pipe_frac", 0.0) * k.get("gpu_cycles_per_step", 0.0) for k in klist) / tot_cyc
        out[key] = {"rotations": s["rotations"], "packed_precision": s.get("packed_precision"), "roofline": roof}
    return out


def kernel_trace_pass(exe, work, configs, batch_log2):
    """`rocprofv3 --kernel-trace` (no counters) around the same child: {section key: {kernel name: {calls, ns}}} or {} on failure.  Counter
    collection stretches the kernels (C5u: 12.9 ms under --pmc against 12.0), and boxes differ by 4 - 6 %: this is the SAME-BOX kernel-trace
    average the bench line's HIP-event `kernel_ms` can be reconciled with (`kernel_ms_trace`)."""
    d = os.path.join(work, "trace")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
           "--pmc-configs", ",".join(configs), "--pmc-evals", "6", "--pmc-settle", str(CLOCK_SETTLE_LAUNCHES)]
    if batch_log2 is not None:
        cmd += ["--batch-log2", str(batch_log2)]
    try:
        res = subprocess.run(cmd, cwd=work, env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")), capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return {}
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("PMC_SECTIONS ")]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if res.returncode != 0 or not line or not files:
        return {}
    sections = json.loads(line[0][len("PMC_SECTIONS "):])
    disp = []
    for path in files:
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                disp.append((int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out, idx, inside = {}, 0, False
    for _, kname, ns in sorted(disp):
        if MARKER in kname:
            if inside:
                idx += 1
            inside = not inside
            continue
        if not inside or idx >= len(sections) or "rnf::" not in kname:
            continue
        e = out.setdefault(sections[idx]["key"], {}).setdefault(kname, {"calls": 0, "ns": 0})
        e["calls"] += 1
        e["ns"] += ns
    return out


def run_pmc_passes(configs, batch_log2, keep_dir=None):
    """Three rocprofv3 children (one per counter group).  Must be called before this process initialises the GPU."""
    from rotationnormflow_amd.build import source_hash
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found on this box: no live PMC passes"}
    passes, sections, t0 = {}, None, time.time()
    work = tempfile.mkdtemp(prefix="rnf_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for grp, counters in PMC_GROUPS.items():
            d = os.path.join(work, grp)
            cmd = [exe, "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
                                                       sys.executable, os.path.abspath(__file__), "--pmc-child", "--pmc-configs", ",".join(configs)]
            if batch_log2 is not None:
                cmd += ["--batch-log2", str(batch_log2)]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            try:
                res = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                return {"error": f"rocprofv3 pass '{grp}' timed out"}
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("PMC_SECTIONS ")]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not line or not files:
                return {"error": f"rocprofv3 pass '{grp}' failed (rc {res.returncode}): {(res.stderr or res.stdout)[-300:]}"}
            sections = json.loads(line[0][len("PMC_SECTIONS "):])
            passes[grp] = pmc_parse(files, sections)
        trace = kernel_trace_pass(exe, work, configs, batch_log2)      # the same child WITHOUT counters: this box's kernel-trace averages
        out = {"sections": pmc_sections_summary(passes, sections, trace), "csrc_sha": source_hash(), "seconds": time.time() - t0,
               "source": f"this run: rocprofv3 --pmc passes ({', '.join(PMC_GROUPS)}; each group its own process) over {PMC_EVALS} evaluations per "
                         "workload and arithmetic; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (gfx950 correction of MI355X_MICROARCH.md), summed over "
                         "every kernel of a step"}
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            with open(os.path.join(keep_dir, "pmc_live.json"), "w") as fh:
                json.dump(out, fh, indent=1)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------------------------------
def profiler_attached():
    """True when this process runs under rocprofv3 / rocprof (tool library preloaded)."""
    env = os.environ
    return bool(env.get("ROCP_TOOL_LIBRARIES") or env.get("ROCPROFILER_LIBRARY_PATH") or env.get("HSA_TOOLS_LIB")
                or "rocprof" in env.get("LD_PRELOAD", "") or "rocprofiler" in env.get("LD_PRELOAD", ""))


def device_state(w, seconds=1.2):
    """Socket power, shader clock and junction temperature (rocm-smi) under this workload: it is launched back to back for `seconds`, OUTSIDE
    every timed region, and rocm-smi is read once near the end.  The split-precision kernels run at the board's power limit, so the clock
    the chip grants differs between boxes and with it every time in the line (profiles/README.md "Power and clock"); None when rocm-smi
    is not there."""
    import re
    import subprocess
    import torch
    if profiler_attached():
        # rocm-smi is a `#!/usr/bin/env python3` script: started from a process a profiler preloaded into, the child would carry the tool
        # library through an exec hop (forbidden on this pool once the GPU is initialised) and its 1.2 s of launches would sit in the trace
        return None
    exe = os.path.realpath(shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi")
    if not os.path.exists(exe):
        return None
    # a fresh interpreter on the script itself (no /usr/bin/env hop), with nothing of a profiler / preload in its environment
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX"))}
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            proc = None
            while time.perf_counter() - t0 < seconds:
                for _ in range(10):
                    w.evaluate()
                torch.cuda.synchronize()
                if proc is None and time.perf_counter() - t0 > 0.6 * seconds:      # launches keep going while rocm-smi reads the sensors
                    proc = subprocess.Popen([sys.executable, exe, "--showpower", "--showclocks", "--showtemp"], stdout=subprocess.PIPE,
                                            stderr=subprocess.DEVNULL, text=True, env=env)
            text = proc.communicate(timeout=20)[0] if proc else ""
        f = lambda pat: (lambda m: float(m.group(1)) if m else None)(re.search(pat, text))
        return {"workload": w.name, "socket_power_w": f(r"Power \(W\): ([0-9.]+)"), "sclk_mhz": f(r"sclk clock level: \d+: \((\d+)Mhz\)"),
                "junction_c": f(r"Sensor junction\) \(C\): ([0-9.]+)"), "how": f"rocm-smi read {0.6 * seconds:.1f} s into {seconds:.1f} s of back-to-back launches, outside the timed region"}
    except Exception as exc:                                     # a diagnostic must never cost the bench line
        return {"workload": w.name, "error": repr(exc)}


def measure(w, args, dist, pmc, steps, warmup, want_secondary, want_parity, cpu_sizes, cpu_budget, blocks=1):
    """One workload -> its record (rank 0 fills the host-side legs).  blocks: the K timed steps are measured this many times and the
    fastest block is kept.  The headline leg (`value`) is ALWAYS one block, as the bench contract says; the `configs` entries and the
    exact-fp32 legs use two (one sweep of round 4 caught a leg whose eight launches ran at half clock: 8.7 ms per step for a kernel that
    takes 4.7 before and after)."""
    import torch
    from rotationnormflow_amd import get_precision, set_precision
    rank0 = dist is None or dist.get_rank() == 0
    world = 1 if dist is None else dist.get_world_size()
    primary = get_precision()

    def timed(n_steps, n_warmup, settle, n_blocks):
        best = w.timed(n_steps, n_warmup, settle, dist)
        for _ in range(n_blocks - 1):
            again = w.timed(n_steps, 0, 0, dist)
            if again[0] < best[0]:
                best = again
        return best
    elapsed, kernel_ms, tot = timed(steps, warmup, CLOCK_SETTLE_LAUNCHES, blocks)
    used = w.fl._packed(w.device, w.feat).precision      # "fp32" when a weight left the fp16 range and the packer fell back
    rec = {"workload": f"{w.name} -- {w.wl['text']}", "value": w.n_global * steps / elapsed, "unit": "rotations/s", "steps": steps, "warmup": warmup,
           "timed_blocks": blocks, "ms_per_step": elapsed / steps * 1e3,
           "dtype": "f32" if used == "fp32" else "f32 (GEMM operands as fp16 hi+lo pairs, fp32 accumulate)",
           "rotations_per_gpu": w.n, "global_batch": w.n_global, "mean_nll": -float(tot[0] / tot[1]),
           "roofline": roofline_of(w, used, kernel_ms, pmc),
           "hbm": {"achieved": w.wl["bytes"] * w.n / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                   "frac": w.wl["bytes"] * w.n / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "algorithmic_bytes_per_rotation": w.wl["bytes"]}}
    if w.weights_path:
        rec["weights"] = os.path.basename(w.weights_path)
    if w.wl.get("also_inverse"):                             # the inverse pass of the same flow on the same rows (root-finder passes depend on the weights)
        with torch.no_grad():
            for _ in range(3):
                w.fl.inverse(w.R, w.feat)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                w.fl.inverse(w.R, w.feat)
            torch.cuda.synchronize()
        rec["ms_per_step_inverse"] = (time.perf_counter() - t0) / steps * 1e3
        from rotationnormflow_amd import runtime as _rt
        rec["fallback_fired"] = bool(_rt.fallback_fired(w.device))
    secondary = None
    if want_secondary and used == "f16x2":
        set_precision("fp32")
        try:
            s_steps = max(2, min(steps, 8))
            s_elapsed, s_kernel_ms, s_tot = timed(s_steps, 2, 4, 2)
            secondary = {"dtype": "f32 (exact fp32-input MFMA)", "value": w.n_global * s_steps / s_elapsed, "unit": "rotations/s",
                         "steps": s_steps, "timed_blocks": 2, "ms_per_step": s_elapsed / s_steps * 1e3, "mean_nll": -float(s_tot[0] / s_tot[1]),
                         "roofline": roofline_of(w, "fp32", s_kernel_ms, pmc)}
        finally:
            set_precision(primary)
        rec["value_fp32_exact"] = secondary["value"]
        rec["ms_per_step_fp32_exact"] = secondary["ms_per_step"]
        rec["secondary"] = secondary
        # the STRICT arithmetic (round 6): bf16x3 -- 24-bit operands as three bf16 terms on the bf16 MFMA, nothing data- or scale-dependent
        set_precision("bf16x3")
        try:
            s_steps = max(2, min(steps, 8))
            s_elapsed, s_kernel_ms, s_tot = timed(s_steps, 2, 4, 2)
            strict = {"dtype": "f32 (GEMM operands as bf16 hi+mid+lo triples, fp32 accumulate)", "value": w.n_global * s_steps / s_elapsed,
                      "unit": "rotations/s", "steps": s_steps, "timed_blocks": 2, "ms_per_step": s_elapsed / s_steps * 1e3,
                      "mean_nll": -float(s_tot[0] / s_tot[1]), "roofline": roofline_of(w, "bf16x3", s_kernel_ms, pmc)}
        finally:
            set_precision(primary)
        rec["value_strict"] = strict["value"]
        rec["ms_per_step_strict"] = strict["ms_per_step"]
        rec["strict"] = strict
    elif used == "fp32":
        rec["value_fp32_exact"], rec["ms_per_step_fp32_exact"] = rec["value"], rec["ms_per_step"]
    if rank0 and world == 1:
        host_legs(w, rec, want_parity, cpu_sizes, cpu_budget)
    return rec


def host_legs(w, rec, want_parity, cpu_sizes, cpu_budget):
    """The host-side legs of a record, outside every timed region: parity of this run's rows against the fp64 oracle, and the CPU baseline
    (the oracle timed on this box's host cores).  At N = 1 they run right after the workload; at N > 1 rank 0 runs them after the process
    group is gone (nobody waits in a collective while one rank computes on the host)."""
    import torch
    from rotationnormflow_amd import get_precision, set_precision
    primary = get_precision()
    secondary = rec.get("secondary")
    if want_parity:
        head, stats, product, want, oracle = w.parity(2048 if w.wl["direction"] == "forward" else 256)
        rec["parity"] = {**head, **stats(product(), want), "reference_fp32": stats(oracle(torch.float32), want)}
        if secondary:
            set_precision("fp32")
            try:
                secondary["parity"] = {"samples": head["samples"], **stats(product(), want)}
            finally:
                set_precision(primary)
        if rec.get("strict"):
            set_precision("bf16x3")
            try:
                rec["strict"]["parity"] = {"samples": head["samples"], **stats(product(), want)}
            finally:
                set_precision(primary)
    if cpu_sizes:
        rec["cpu_baseline"] = cpu_baseline(w.cfg, w.weights, w.A, w.wl, w.feat_dim, cpu_sizes, cpu_budget, w.queries)
        rec["vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]


def measure_training(device, batch=1024, steps=40, warmup=10):
    """One training iteration of the reference's unconditional recipe (settings/raw.yml: 24 layer pairs, K = 64, batch 1024; agent.py:75-92:
    forward, loss = mean(-ldjs), zero_grad, backward, optimizer.step()) exactly as the reference's unedited driver runs it: the flow from
    ``get_flow`` (flattened parameters), ``torch.optim.Adam(flow.parameters(), lr)`` with torch's defaults, eager.  Beside it the same loop
    on per-tensor parameters (``Flow(config)``) and the HIP-graph replay of the iteration."""
    import torch
    from rotationnormflow_amd import make_config, synth
    from rotationnormflow_amd.flow.flow import Flow, get_flow
    from rotationnormflow_amd.harness import GraphedTrainStep
    cfg = make_config("C2")
    R = torch.from_numpy(synth.uniform_rotations(batch, seed=1)).to(device)
    out = {"workload": "training iteration (forward + backward + Adam), settings/raw.yml recipe: 24-layer MobiusAffine, K = 64, batch 1024, eager, "
                       "torch.optim.Adam defaults (agent.py:23,75-92)", "batch": batch, "unit": "ms per iteration", "timed_blocks": 3}

    def run(flow, opt):
        def step():
            opt.zero_grad()
            _, ldj = flow(R)
            loss = (-ldj).mean()
            loss.backward()
            opt.step()
        for _ in range(warmup):
            step()
        best = float("inf")
        for _ in range(3):                                   # an eager loop is exposed to the host: the fastest of three timed blocks
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best

    with contextlib.redirect_stdout(io.StringIO()):
        flat, classic, graphed = get_flow(cfg), Flow(cfg), get_flow(cfg)
    flat, classic, graphed = flat.to(device).train(), classic.to(device).train(), graphed.to(device).train()
    out["ms_per_iteration"] = run(flat, torch.optim.Adam(flat.parameters(), 1e-4))
    out["value"] = batch / (out["ms_per_iteration"] * 1e-3)
    out["parameters"] = f"flattened: {len(list(flat.parameters()))} tensor" if flat.is_flat else "per tensor"
    out["ms_per_iteration_per_tensor_parameters"] = run(classic, torch.optim.Adam(classic.parameters(), 1e-4))
    opt = torch.optim.Adam(graphed.parameters(), 1e-4, fused=True, capturable=True)
    gstep = GraphedTrainStep(graphed, opt, tuple(R.shape))
    for _ in range(warmup):
        gstep(R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gstep(R)
    torch.cuda.synchronize()
    out["ms_per_iteration_hip_graph"] = (time.perf_counter() - t0) / steps * 1e3
    return out


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _sig(x, digits=6):
    """Round floats to `digits` significant digits (the compact line is read by people and by a parser with a size limit)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


THIN_CONFIGS = ("C3", "C2t", "C4t")
COMPACT_LIMIT = 3000                      # bytes; the driver keeps ~8 KB of stdout tail and parses the LAST line (BENCH_r04: a 40 KB line was lost)
ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "kernel_ms_trace", "traffic", "traffic_algorithmic", "valu_issue_frac",
             "matrix_pipe_frac")


def compact_record(out, full_path=None):
    """The driver-facing line: every contract key, the roofline / cpu_baseline objects in short form, one small entry per other config."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data"))
    cfg = dict(out["config"])
    cfg["workload"] = cfg["workload"][:100]
    c["config"] = cfg
    c.update(_pick(out, ("rccl_ranks", "backend", "value_strict", "value_fp32_exact", "mean_nll")))      # (their ms_per_step: full record)
    if "strict" in out:                                        # the strict leg's own roofline and parity, short
        st = out["strict"]
        c["strict"] = {"dtype": "bf16x3", **_pick(st.get("roofline", {}), ("frac", "frac_executed", "kernel_ms", "valu_issue_frac", "matrix_pipe_frac"))}
        if "parity" in st:
            c["strict"]["parity_max"] = st["parity"].get("max_abs_err")
    if "roofline" in out:
        c["roofline"] = _pick(out["roofline"], ROOF_KEYS)
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]                             # the sample as NUMBERS (round 5 cut a sentence off mid-number) + a short label
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "threads_available", "kind", "cpu", "n", "best_s", "repetitions"))
        c["cpu_baseline"]["sample"] = f"oracle fp32, best of {cb.get('repetitions', '?')} at N={cb.get('n', '?')}"
        c["vs_cpu_baseline"] = out.get("vs_cpu_baseline")
    if "parity" in out:
        c["parity"] = _pick(out["parity"], ("mean_abs_err_of_the_mean", "max_abs_err"))
        c["parity"]["reference_fp32_max"] = out["parity"].get("reference_fp32", {}).get("max_abs_err")
    if "device_state" in out:
        c["device_state"] = _pick(out["device_state"], ("socket_power_w", "sclk_mhz"))
    small = {}
    for name, r in out.get("configs", {}).items():
        if name == "train":
            small[name] = _pick(r, ("ms_per_iteration", "ms_per_iteration_hip_graph"))
            continue
        thin = name in THIN_CONFIGS
        e = _pick(r, ("value", "ms_per_step", "ms_per_step_inverse") if thin else ("value", "ms_per_step", "value_strict"))     # (value_fp32_exact: full record)
        roof = r.get("roofline", {})
        if not thin:                       # (C3 runs C2's kernel; the trained-weights entries are about time and parity)
            e.update(_pick(roof, ("frac", "valu_issue_frac", "matrix_pipe_frac")))
            if roof.get("traffic") and roof.get("traffic_algorithmic"):
                e["traffic_x"] = roof["traffic"] / roof["traffic_algorithmic"]
        if "parity" in r:
            e["parity_max"] = r["parity"].get("max_abs_err")
        for k in ("value", "value_strict", "value_fp32_exact"):            # whole rotations per second: shorter than a float and exact enough at 3 digits
            if k in e:
                e[k] = int(float(f"{e[k]:.3g}"))
        small[name] = e
    if small:
        c["configs"] = _sig(small, 3)
    if full_path:
        c["full_record"] = full_path
    c = _sig(c, 5)
    if "mean_nll" in out:
        c["mean_nll"] = out["mean_nll"]                    # the statistic itself: every digit
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT and "configs" in c:      # never lose the headline to an over-long line: first thin the per-config entries
        c["configs"] = {k: _pick(v, ("value", "ms_per_step", "frac", "ms_per_iteration")) for k, v in c["configs"].items()}
        line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:                         # ... then drop the optional parts, largest first
        for k in ("configs", "parity", "device_state", "cpu_baseline"):
            if k == "cpu_baseline":
                c.get(k, {}).pop("sample", None)
            else:
                c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
    return line


def emit(out, full_out):
    """stdout: `BENCH_FULL {...}` (everything: per-kernel counter lists, both arithmetics, parity blocks, CPU runs) on an earlier line and in
    `full_out` (default bench_full.json beside this script); then, as the LAST line, the compact record (< COMPACT_LIMIT bytes)."""
    path = full_out or os.path.join(ROOT, "bench_full.json")
    rel = None
    try:
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError as exc:
        print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
    print("BENCH_FULL " + json.dumps(out), flush=True)
    print(compact_record(out, rel), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS), help="run this workload alone (default: C2 headline + every other config)")
    ap.add_argument("--batch-log2", type=int, default=None, help="log2 of the batch: per GPU (weak-scaling configs, default 20) or global (C3, default 22)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs (CPU baseline and the oracle parity block)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the exact-fp32 leg")
    ap.add_argument("--no-configs", action="store_true", help="headline workload only")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes")
    ap.add_argument("--save-pmc", default=None, metavar="DIR", help="also write the live PMC summary to DIR/pmc_live.json (profiles/<round>/)")
    ap.add_argument("--weights", default=None, metavar="CKPT", help="load the headline workload's flow from this checkpoint (Agent.save_ckpt "
                    "layout, e.g. tests/golden/trained_c2.pth) instead of the recipe weights")
    ap.add_argument("--full-out", default=None, metavar="PATH", help="where the full record goes (default: bench_full.json beside bench.py); "
                    "stdout carries it on a `BENCH_FULL ` line and ends with the compact record")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-evals", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-settle", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-configs", default=",".join(ALL_CONFIGS), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.pmc_child:
        return pmc_child(args)

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
    if os.environ.get("RNF_BENCH_DIE_RANK") == str(rank):     # test rig: this rank dies before the rendezvous; the launcher must end the others
        print(f"bench.py: rank {rank} exits on request (RNF_BENCH_DIE_RANK)", file=sys.stderr)
        sys.exit(3)
    if os.environ.get("RNF_BENCH_HANG_DUMP"):                 # diagnostics: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RNF_BENCH_HANG_DUMP"]), exit=True)

    headline = args.config or "C2"
    full = args.config is None and not args.no_configs and not distributed
    others = [c for c in ALL_CONFIGS if c != headline] if full else []

    import torch                                              # (importing torch and counting devices does not initialise the GPU)

    # live counter passes: children under rocprofv3, BEFORE this process touches the GPU (only in the single-process run)
    pmc = None
    if not distributed and not args.no_pmc:
        if torch.cuda.device_count() < 1:
            print("bench.py: no GPU visible", file=sys.stderr)
            sys.exit(2)
        pmc = run_pmc_passes([headline] + others, args.batch_log2, args.save_pmc)

    # RNF_BENCH_SHARED_GPU=1 (test rig only: a 1-GPU box): every rank uses cuda:0 and the collective runs over gloo, so that the
    # N > 1 control flow of this script can be exercised without N GPUs.  The driver never sets it.
    shared_gpu = os.environ.get("RNF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        sys.exit(2)
    backend, dist = None, None
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

    want_host = world == 1 and not args.no_cpu_baseline
    w = Workload(headline, device, args.batch_log2, rank, world, weights_path=args.weights)
    if w.feat is not None and distributed:
        from rotationnormflow_amd.dist import calibrate_feature_scale
        calibrate_feature_scale(w.fl, w.feat)                 # one calibration for all ranks: identical packed images (dist.py)
    head = measure(w, args, dist, pmc, args.steps, args.warmup, not args.no_secondary, want_host,
                   (4096, 65536) if want_host else None, 36.0)
    state = device_state(w) if (rank == 0 and world == 1) else None
    configs = {}
    prev = w
    for name in others:
        w.R = w.feat = None
        torch.cuda.empty_cache()
        w = Workload(name, device, args.batch_log2, 0, 1, share=prev)
        prev = w
        c_steps = 6 if name == "C3" else 10
        # bounded host legs for the secondary configs: N = 4096 (1024 for the inverse passes, which the oracle runs at ~2e3 rotations/s)
        sizes = ((4096,) if w.wl["direction"] == "forward" else (1024,)) if want_host else None
        second = not args.no_secondary and not w.weights_path      # (the trained-weights entries: default arithmetic, both directions)
        configs[name] = measure(w, args, None, pmc, c_steps, 3, second, want_host, sizes, 10.0, blocks=2)
    if full and rank == 0:
        w.R = w.feat = None
        torch.cuda.empty_cache()
        configs["train"] = measure_training(device)           # the reference's training step, on the driver's clock too

    ranks_seen = world
    if distributed:
        # the rank count READ BACK from the communicator: one more real all-reduce (a 1 from every rank), then the group is torn down;
        # rank 0's host legs run after that, so no rank sits in a collective while another computes on the host
        one = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(one)
        ranks_seen = int(round(float(one.item())))
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        # N > 1: a short CPU baseline and the parity block of rank 0's shard, after the timed region and after the collectives
        host_legs(w, head, True, (4096,), 8.0)
    if rank == 0:
        strong = bool(WORKLOADS[headline].get("strong"))
        out = {
            "metric": "rotation log_prob evals/s (24-layer MobiusAffine + matrix-Fisher base), mean NLL alongside" if headline == "C2"
                      else f"rotation evals/s, workload {headline}",
            "value": head["value"], "unit": "rotations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,         # the reference publishes no number for this metric; see vs_cpu_baseline
            "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": head["workload"], "rotations_per_gpu": head["rotations_per_gpu"], "global_batch": head["global_batch"],
                       "parallelism": (f"batch-sharded x{world}, one all-reduce of {{sum log p, count}} per step" if world > 1 else "single GPU")},
            "rccl_ranks": ranks_seen, "backend": backend, "mean_nll": head["mean_nll"], "clock_settle_launches": CLOCK_SETTLE_LAUNCHES,
        }
        for k in ("value_fp32_exact", "ms_per_step_fp32_exact", "value_strict", "ms_per_step_strict", "strict", "roofline", "hbm", "secondary", "parity", "cpu_baseline", "vs_cpu_baseline"):
            if k in head:
                out[k] = head[k]
        if state is not None:
            out["device_state"] = state
        if configs:
            out["configs"] = configs
        if pmc is not None:
            out["pmc"] = {k: pmc[k] for k in ("source", "seconds", "csrc_sha", "error") if k in pmc}
        emit(out, args.full_out)


if __name__ == "__main__":
    main()
