#!/usr/bin/env python3
"""A/B of several builds of librnf_hip.so in ONE process, interleaved rounds (cdna_hip_programming.md section 5.4 rule 24).

    python tools/ab_variants.py --build base= lean="-DRNF_X=1" ...     # builds tools/_build/librnf_<name>.so (hipcc flags after '=')
    python tools/ab_variants.py --run base lean [--preset C2] [--direction forward] [--rounds 7] [--steps 6]

--run prints one JSON line per variant: median / min ms per launch over the rounds, and the mean NLL (must agree between variants).
A variant named `cur` is the shipped rotationnormflow_amd/librnf_hip.so.
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
SRC = os.path.join(ROOT, "rotationnormflow_amd", "csrc", "rnf_api.hip")


def lib_path(name):
    return os.path.join(ROOT, "rotationnormflow_amd", "librnf_hip.so") if name == "cur" else os.path.join(BUILD, f"librnf_{name}.so")


def build(specs):
    os.makedirs(BUILD, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Xarch_host", "-mf16c"] + flags.split() + [
            "-shared", "-fPIC", "-o", lib_path(name), SRC]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait():
            raise SystemExit(f"build of variant {name} failed")
        print("built", lib_path(name))


def run(args):
    import numpy as np
    import torch
    from rotationnormflow_amd import _lib, make_config, runtime, synth
    from rotationnormflow_amd.flow.flow import Flow
    from rotationnormflow_amd.utils.fisher import MatrixFisherN

    dev = torch.device("cuda", 0)
    cfg = make_config(args.preset)
    n = 1 << args.batch_log2
    R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).to(dev)
    base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")))
    handles, flows = {}, {}
    for name in args.run:
        handles[name] = _lib.load(lib_path(name))
        with contextlib.redirect_stdout(io.StringIO()):
            fl = Flow(cfg)
        shapes = {k: tuple(v.shape) for k, v in fl.state_dict().items()}
        w = synth.fill_state_dict(shapes, seed=2024, regime="trained")
        fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        flows[name] = fl.to(dev).eval()          # one module (and one pack cache) per variant: image layouts may differ between builds
    feat = None
    if cfg.condition:
        feat = torch.from_numpy(synth.features(n, flows[args.run[0]].feature_dim, seed=43)).to(dev)
    def step(name):
        _lib._lib = handles[name]
        fl = flows[name]
        if args.direction == "inverse":
            return fl.inverse(R, feat)[1]
        return fl.log_prob(R, feat, base=base)["sum"]

    times = {name: [] for name in args.run}
    out = {}
    with torch.no_grad():
        for name in args.run:                       # clock settle + packing
            for _ in range(12):
                out[name] = step(name)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for name in args.run:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                step(name)
                a.record()
                for _ in range(args.steps):
                    out[name] = step(name)
                b.record()
                torch.cuda.synchronize()
                times[name].append(a.elapsed_time(b) / args.steps)
    for name in args.run:
        t = np.array(times[name])
        o = out[name].cpu().double().numpy()
        stat = -float(o[0] / o[1]) if args.direction != "inverse" else float(o.mean())
        print(json.dumps({"variant": name, "preset": args.preset, "direction": args.direction, "n": n, "ms_median": float(np.median(t)),
                          "ms_min": float(t.min()), "ms_all": [round(float(x), 4) for x in t],
                          "mean_nll" if args.direction != "inverse" else "mean_ldj": stat}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="*", default=[])
    ap.add_argument("--run", nargs="*", default=[])
    ap.add_argument("--preset", default="C2")
    ap.add_argument("--direction", default="forward")
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    if args.build:
        build(args.build)
    if args.run:
        run(args)


if __name__ == "__main__":
    main()
