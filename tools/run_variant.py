#!/usr/bin/env python3
"""Run one build of the library (tools/ab_variants.py --build) on one BASELINE preset for a few steps: the process rocprofv3 wraps in
tools/kernel_times.sh (per-kernel durations of a variant).   python3 tools/run_variant.py <variant|cur> <preset> [steps] [direction]"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rotationnormflow_amd import _lib, make_config, synth  # noqa: E402
from rotationnormflow_amd.flow.flow import Flow  # noqa: E402

name, preset = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
direction = sys.argv[4] if len(sys.argv) > 4 else ("inverse" if preset.startswith("C5") else "forward")
path = os.path.join(ROOT, "rotationnormflow_amd", "librnf_hip.so") if name == "cur" else os.path.join(ROOT, "tools", "_build", f"librnf_{name}.so")
_lib._lib = _lib.load(path)
dev = torch.device("cuda", 0)
cfg = make_config(preset)
with contextlib.redirect_stdout(io.StringIO()):
    fl = Flow(cfg)
w = synth.fill_state_dict({k: tuple(v.shape) for k, v in fl.state_dict().items()}, seed=2024, regime="trained")
fl.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
fl = fl.to(dev).eval()
n = 1 << int(os.environ.get("RNF_LOG2N", "20"))
R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).to(dev)
feat = torch.randn((n, fl.feature_dim), device=dev) if cfg.condition else None
with torch.no_grad():
    for _ in range(steps + 10):
        out = fl.inverse(R, feat)[1] if direction == "inverse" else fl.log_prob(R, feat)["sum"]
torch.cuda.synchronize()
print(name, preset, float(out.double().mean() if direction == "inverse" else -out[0] / out[1]))
