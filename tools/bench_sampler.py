#!/usr/bin/env python3
"""Throughput of the on-device matrix-Fisher sampler (MatrixFisherN._sample, utils/fisher.py:117-207,234-243) and of the C5 pose pipeline
it feeds: draw base samples -> Flow.inverse -> score (agent.py:238-263).   python tools/bench_sampler.py [--log2 20]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rotationnormflow_amd import synth  # noqa: E402
from rotationnormflow_amd.utils.fisher import MatrixFisherN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    n = 1 << a.log2
    for kind, rows in (("diag531", 1), ("tilted", 1), ("diag531", 1024)):
        A = torch.from_numpy(synth.fisher_A(kind)).cuda().repeat(rows, 1, 1)
        base = MatrixFisherN(A)
        per = n // rows
        base._sample(per)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = base._sample(per)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps(dict(metric="matrix-Fisher samples/s", A=kind, rows=rows, samples=rows * per, ms=dt * 1e3,
                              samples_per_s=rows * per / dt, out_shape=list(out.shape))))


if __name__ == "__main__":
    main()
