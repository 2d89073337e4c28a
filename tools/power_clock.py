#!/usr/bin/env python3
"""Diagnostic: socket power, shader clock and junction temperature (rocm-smi) WHILE a BASELINE config runs back to back for a few seconds.

    python tools/power_clock.py [C2 C4 C5 C5u ...] [--seconds 6] [--precision fp32]

One JSON line per config: ms per step over the sustained run, and the samples taken during it (the sampler thread polls rocm-smi every
~0.3 s; the first second is dropped: the SMU needs that long to settle)."""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
    f = lambda pat: (lambda m: float(m.group(1)) if m else None)(re.search(pat, out))
    return {"power_w": f(r"Power \(W\): ([0-9.]+)"), "sclk_mhz": f(r"sclk clock level: \d+: \((\d+)Mhz\)"),
            "junction_c": f(r"Sensor junction\) \(C\): ([0-9.]+)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C2", "C4", "C5", "C5u"])
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--lib", default=None, help="path of a variant librnf_hip.so (tools/ab_variants.py --build), e.g. a knock-out build")
    args = ap.parse_args()
    import torch
    if args.lib:
        from rotationnormflow_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    import bench
    from rotationnormflow_amd import set_precision
    if args.precision:
        set_precision(args.precision)
    dev = torch.device("cuda", 0)
    for name in args.configs:
        w = bench.Workload(name, dev)
        with torch.no_grad():
            for _ in range(5):
                w.evaluate()
            torch.cuda.synchronize()
            samples, stop = [], threading.Event()

            def poll():
                t0 = time.perf_counter()
                while not stop.is_set():
                    s = sample()
                    s["t"] = round(time.perf_counter() - t0, 2)
                    samples.append(s)
                    time.sleep(0.25)
            th = threading.Thread(target=poll)
            th.start()
            t0, steps = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(20):
                    w.evaluate()
                torch.cuda.synchronize()
                steps += 20
            elapsed = time.perf_counter() - t0
            stop.set()
            th.join()
        kept = [s for s in samples if s["t"] >= 1.0 and s["power_w"] is not None]
        mean = lambda k: round(sum(s[k] for s in kept) / max(len(kept), 1), 1)
        print(json.dumps({"config": name, "lib": os.path.basename(args.lib) if args.lib else "shipped", "precision": args.precision or "f16x2", "ms_per_step": round(elapsed / steps * 1e3, 3), "samples": len(kept),
                          "power_w_mean": mean("power_w"), "power_w_max": max((s["power_w"] for s in kept), default=None),
                          "sclk_mhz_mean": mean("sclk_mhz"), "sclk_mhz_min": min((s["sclk_mhz"] for s in kept), default=None),
                          "junction_c_max": max((s["junction_c"] for s in kept), default=None)}), flush=True)
        del w
        torch.cuda.empty_cache()
        time.sleep(2.0)


if __name__ == "__main__":
    main()
