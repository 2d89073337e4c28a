#!/usr/bin/env python3
"""Diagnostic: where does a wave spend its cycles?  Builds a SEPARATE library with -DRNF_STAMPS (s_memtime stamps at the
phase boundaries of the layer loop; the shipped librnf_hip.so contains none), runs a workload and prints the share of
wave-cycles per phase.  Read the SHARES, not the run time (stamps fence the scheduler).

    python tools/phase_stamps.py [--preset C2] [--direction forward] [--batch-log2 20] [--build-only]
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ["0 G-load + sync staging", "1 frame + hidden layers (H)", "2 barrier B1", "3 fc_last tiles + segments (L)",
          "4 barrier B2", "5 layer finish", "6 affine16 layer", "7 tile prologue/epilogue", "8 fused projection tile 0",
          "9 fused projection tile 1", "10 fused tile start (features + first projection)", "11 fused: the two extra barriers"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="C2")
    ap.add_argument("--direction", default="forward")
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()

    out = os.path.join(ROOT, "tools", "_build", "librnf_hip_stamps.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "rotationnormflow_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DRNF_STAMPS"] + os.environ.get("RNF_STAMPS_FLAGS", "").split() + [
                        "-shared", "-fPIC", "-o", out, os.path.join(csrc, "rnf_api.hip")], check=True)
    if args.build_only:
        return

    import torch
    from rotationnormflow_amd import _lib, make_config, synth
    _lib.LIB_PATH = out
    from rotationnormflow_amd.flow.flow import Flow
    from rotationnormflow_amd.utils.fisher import MatrixFisherN

    dev = torch.device("cuda", 0)
    cfg = make_config(args.preset)
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
    w = synth.fill_state_dict(shapes, seed=2024, regime="trained")
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    fl = fl.to(dev).eval()
    n = 1 << args.batch_log2
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).to(dev)
    feat = torch.from_numpy(synth.features(n, fl.feature_dim, seed=43)).to(dev) if cfg.condition else None
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")))
    stamps = torch.zeros(12, dtype=torch.int64, device=dev)
    os.environ["RNF_STAMPS_PTR"] = str(stamps.data_ptr())

    def step():
        if args.direction == "forward":
            fl.log_prob(R, feat, base=base)
        else:
            fl.inverse(R, feat)

    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        stamps.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(float)
    waves = (n + 31) // 32
    print(f"staging={os.environ.get('RNF_STAGING', 'dma')} preset={args.preset} {args.direction} n={n}: "
          f"{a.elapsed_time(b):.2f} ms (instrumented)")
    print(f"{'phase':40s} {'share':>7s} {'cycles/wave-tile':>18s}")
    phases = list(PHASES)
    if args.direction != "forward":                 # the inverse pass places the stamps differently (flow_kernels.h: b2_sync / mobius_inv_finish / b2_issue)
        phases[3] = "3 fc_last tiles + segment parameters (L)"
        phases[4] = "4 barrier B2 + ROOT FINDER + DMA issue"
        phases[5] = "5 behind the layer"
    for name, v in zip(phases, s):
        print(f"{name:40s} {v / s.sum() * 100:6.1f}% {v / waves:18.0f}")
    print(f"{'total':40s} {100:6.1f}% {s.sum() / waves:18.0f}")


if __name__ == "__main__":
    main()
