"""Inverse-pass errors of the HIP path against the reference's golden vectors, for every inverse case: log-det and rotation error beside the
reference's own fp32-vs-fp64 spread (calibrates the tolerances of tests/test_gpu_parity.py).  `python tools/inverse_stats.py` (GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rotationnormflow_amd import runtime  # noqa: E402
from tests.golden.cases import CASES  # noqa: E402
from tests.gpu_helpers import run_case  # noqa: E402

CELL = np.pi / 2 ** 14


def main():
    for name, spec in CASES.items():
        if spec["direction"] != "inverse":
            continue
        row = {"case": name}
        for prec in ("f16x2", "bf16x3", "fp32"):
            runtime.set_precision(prec)
            _, Rt, ldj, fx, _, _ = run_case(name)
            err = np.abs(ldj - fx["ldj64"])
            rerr = np.abs(Rt - fx["rot64"]).reshape(len(err), -1).max(1)
            row[prec] = {"ldj_mean": float(err.mean()), "ldj_p99": float(np.quantile(err, 0.99)), "ldj_max": float(err.max()),
                         "rot_mean": float(rerr.mean()), "rot_p99": float(np.quantile(rerr, 0.99)), "rot_max": float(rerr.max()),
                         "rot_max_cells": float(rerr.max() / CELL), "frac_rot_gt_half_cell": float(np.mean(rerr > 0.5 * CELL))}
        noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
        rnoise = np.abs(fx["rot32"].astype(np.float64) - fx["rot64"]).reshape(len(noise), -1).max(1)
        row["reference_fp32"] = {"ldj_mean": float(noise.mean()), "ldj_p99": float(np.quantile(noise, 0.99)), "ldj_max": float(noise.max()),
                                 "rot_mean": float(rnoise.mean()), "rot_max": float(rnoise.max()), "rot_max_cells": float(rnoise.max() / CELL),
                                 "frac_rot_gt_half_cell": float(np.mean(rnoise > 0.5 * CELL))}
        print(json.dumps(row))


if __name__ == "__main__":
    main()
