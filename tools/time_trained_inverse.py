#!/usr/bin/env python3
"""Inverse-pass time of the reference-trained checkpoints (tests/golden/trained_c2.pth, trained_c4.pth) for one or more builds of the library
(tools/ab_variants.py --build): root-finder pass counts depend on the weights, the synthetic presets are mild.
    python tools/time_trained_inverse.py <variant|cur> [...]"""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rotationnormflow_amd import _lib, harness, synth  # noqa: E402
from tests.trained_helpers import load_trained  # noqa: E402

n = 1 << 20
dev = torch.device("cuda", 0)
R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).to(dev)
for name in sys.argv[1:]:
    path = os.path.join(ROOT, "rotationnormflow_amd", "librnf_hip.so") if name == "cur" else os.path.join(ROOT, "tools", "_build", f"librnf_{name}.so")
    _lib._lib = _lib.load(path)
    for ck in ("trained_c2", "trained_c4", "trained_c1"):
        cfg, ckpt, w, fx, spec = load_trained(ck)
        fl = harness.build_flow_from_checkpoint(cfg, ckpt)
        feat = None
        if cfg.condition:
            base = torch.from_numpy(fx["test_feat"]).to(dev)
            feat = base[torch.arange(n, device=dev) % base.shape[0]].contiguous()
        with torch.no_grad():
            for _ in range(6):
                out = fl.inverse(R, feat)[1]
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8):
                out = fl.inverse(R, feat)[1]
            b.record()
            torch.cuda.synchronize()
        print(json.dumps({"variant": name, "checkpoint": ck, "ms_inverse": a.elapsed_time(b) / 8, "mean_ldj": float(out.double().mean())}))
