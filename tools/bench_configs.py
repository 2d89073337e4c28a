#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one MI355X (C1..C5, forward / inverse), as a small table + JSON lines.
Not the driver's benchmark (that is bench.py, config C2); this is the per-config evidence quoted in DESIGN.md / profiles/.

    python tools/bench_configs.py [--batch-log2 20] [--steps 5] [--only C2,C4]
"""
import argparse
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from rotationnormflow_amd import make_config, synth  # noqa: E402
from rotationnormflow_amd.flow.flow import Flow  # noqa: E402
from rotationnormflow_amd.utils.fisher import MatrixFisherN  # noqa: E402

# GEMM FLOP per rotation, name, preset, direction, fisher
RUNS = [
    ("C1 8-layer uncond fwd", "C1", "forward", False, 461_824),
    ("C2 24-layer uncond + Fisher fwd", "C2", "forward", True, 1_385_472),
    ("C4 24-layer cond F=256 fwd", "C4", "forward", False, 2_231_296),
    ("C5 42-layer Mobius-only cond F=512 inverse", "C5", "inverse", True, 5_177_088),
    ("C5u 42-layer Mobius-only uncond inverse", "C5u", "inverse", False, 2_424_576),
    ("C2 inverse (24-layer uncond)", "C2", "inverse", False, 1_385_472),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = 1 << args.batch_log2
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).to(dev)
    only = set(filter(None, args.only.split(",")))
    for name, preset, direction, fisher, flop in RUNS:
        if only and preset not in only:
            continue
        cfg = make_config(preset)
        with contextlib.redirect_stdout(io.StringIO()):
            fl = Flow(cfg)
        shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
        w = synth.fill_state_dict(shapes, seed=7, regime="trained")
        fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        fl = fl.to(dev).eval()
        feat = None
        if cfg.condition:
            feat = torch.from_numpy(synth.features(n, fl.feature_dim, seed=43)).to(dev)
        base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531"))) if fisher else None

        def step():
            if direction == "forward":
                return fl.log_prob(R, feat, base=base)["sum"]
            out, ldj = fl.inverse(R, feat)
            return ldj

        with torch.no_grad():
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                step()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        rec = dict(config=name, rotations=n, ms=ms, rot_per_s=n / ms * 1e3, gemm_tflops=flop * n / ms / 1e9,
                   frac_fp32_mfma_peak=flop * n / ms / 1e9 / 157.3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
