"""Diagnostic (GPU): LEAN 16-wave launch vs 4-wave launch of the same rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rotationnormflow_amd import runtime, synth
from rotationnormflow_amd.utils.fisher import MatrixFisherN
from tests.test_gpu_scale_properties import _c2
cfg, w, fl = _c2()
n = 1 << 20
R = torch.from_numpy(synth.uniform_rotations(n, seed=42)).cuda()
base = MatrixFisherN(torch.from_numpy(synth.fisher_A("diag531")))
with torch.no_grad():
    full = fl.log_prob(R, base=base)["logp"]
    print("fallback fired on full:", runtime.fallback_fired(R.device))
    small = fl.log_prob(R[:4096].contiguous(), base=base)["logp"]
    print("fallback fired on small:", runtime.fallback_fired(R.device))
d = (small - full[:4096]).abs()
print("max", d.max().item(), "n>2e-6:", int((d > 2e-6).sum()), "idx", torch.nonzero(d > 2e-6).flatten()[:10].tolist())
from oracle import flow_oracle as orc
idx = torch.nonzero(d > 2e-6).flatten()[:8].cpu().numpy()
if len(idx):
    want, _ = orc.log_prob(cfg, w, R[idx].cpu().numpy(), None, synth.fisher_A("diag531"), torch.float64)
    print("oracle", want.numpy()); print("full  ", full[idx].cpu().numpy()); print("small ", small[idx].cpu().numpy())
