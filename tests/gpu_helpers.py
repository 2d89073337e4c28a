"""Helpers for the -m gpu parity tests: build the product Flow for a golden case and run it through the C ABI."""
import numpy as np
import torch

from rotationnormflow_amd.flow.flow import Flow
from tests.helpers import load_case


def product_flow(cfg, weights, device="cuda"):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        fl = Flow(cfg)
    fl.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return fl.to(device).eval()


def run_case(name, device="cuda"):
    cfg, w, R, feat, fx, spec = load_case(name)
    fl = product_flow(cfg, w, device)
    Rd = torch.from_numpy(R).to(device)
    fd = None if feat is None else torch.from_numpy(feat).to(device)
    if fd is not None and spec.get("regime") == "imbalanced":
        # un-normalised features (x30, synth.FEATURE_SCALE): what Flow.calibrate_feature_scale is for.  (Round 6: nothing is measured by
        # default -- the packed images are a function of the weights and of this one explicit number)
        fl.calibrate_feature_scale(fd)
    with torch.no_grad():
        if spec["direction"] == "forward":
            Rt, ldj = fl(Rd, fd)
        else:
            Rt, ldj = fl.inverse(Rd, fd)
    torch.cuda.synchronize()
    return fl, Rt.cpu().numpy().astype(np.float64), ldj.cpu().numpy().astype(np.float64), fx, spec, (Rd, fd)
