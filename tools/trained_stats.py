#!/usr/bin/env python3
"""How "trained" the reference-trained checkpoints are (tests/golden/make_trained.py): over the fixture's test rows, the largest raw
segment-weight pre-activation s, the largest raw centre norm |w| (before the 0.7 / (1 + |w|) squash) and the largest conditioner output of
any Moebius layer, evaluated with the CPU oracle in fp64 (checker only; no GPU).  `python tools/trained_stats.py [name ...]`"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as orc  # noqa: E402
from tests.golden.trained_cases import TRAINED  # noqa: E402
from tests.trained_helpers import load_trained  # noqa: E402


def main():
    seen = {}
    real = orc.conditioner

    def spy(x, p, prefix):
        out = real(x, p, prefix)
        seen.setdefault(prefix, []).append(out.detach())
        return out
    orc.conditioner = spy
    for name in (sys.argv[1:] or list(TRAINED)):
        cfg, ckpt, w, fx, spec = load_trained(name)
        seen.clear()
        n = min(512, fx["test_rot"].shape[0])
        feat = fx["test_feat"][:n] if "test_feat" in fx else None
        orc.flow_forward(cfg, w, fx["test_rot"][:n], feat, dtype=torch.float64)
        K = cfg.segments
        s_max = w_max = c_max = 0.0
        for prefix, outs in seen.items():
            c = torch.cat(outs)
            if c.shape[1] != 4 * K:
                continue                                      # a Condition16Trans network
            s_max = max(s_max, float(c[:, :K].abs().max()))
            w_max = max(w_max, float(c[:, K:].reshape(-1, K, 3).norm(dim=-1).max()))
            c_max = max(c_max, float(c.abs().max()))
        print(json.dumps({"checkpoint": name, "rows": n, "layers": len(seen), "max_abs_s": s_max, "max_raw_centre_norm": w_max,
                          "max_abs_conditioner_output": c_max, "ldj_range": [float(fx["ldj64"].min()), float(fx["ldj64"].max())],
                          "mean_ll64": float(fx["mean_ll64"])}))


if __name__ == "__main__":
    main()
