"""Per-sample log-det error of the HIP path against the reference's fp64 golden vectors, for both GEMM arithmetics, beside the
reference's own fp32-vs-fp64 noise.  `python tools/parity_stats.py [case ...]` (GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rotationnormflow_amd import runtime  # noqa: E402
from tests.gpu_helpers import run_case  # noqa: E402


def main():
    names = sys.argv[1:] or ["c2_trained", "c4_trained", "c1_default"]
    if names == ["--all-forward"]:
        from tests.golden.cases import CASES
        names = [k for k, v in CASES.items() if v["direction"] == "forward"]
    for name in names:
        row = {"case": name}
        for prec in ("f16x2", "bf16x3", "fp32"):
            runtime.set_precision(prec)
            _, _, ldj, fx, _, _ = run_case(name)
            err = np.abs(ldj - fx["ldj64"])
            noise_ = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
            e32 = np.abs(ldj - fx["ldj32"].astype(np.float64))
            row[prec] = {"mean": float(err.mean()), "p99": float(np.quantile(err, 0.99)), "max": float(err.max()),
                         "mean_ldj_err": float(abs(ldj.mean() - fx["ldj64"].mean())),
                         # SURVEY 8(c) per-sample form: |hip - ref64| <= max(1e-5, 2 |ref32 - ref64|), and p99 against the fp32 reference
                         "per_sample_pass": float(np.mean(err <= np.maximum(1e-5, 2 * noise_))),
                         "per_sample_worst_excess": float(np.max(err - np.maximum(1e-5, 2 * noise_))),
                         "p99_vs_ref32": float(np.quantile(e32, 0.99))}
        noise = np.abs(fx["ldj32"].astype(np.float64) - fx["ldj64"])
        row["reference_fp32"] = {"mean": float(noise.mean()), "p99": float(np.quantile(noise, 0.99)), "max": float(noise.max())}
        print(json.dumps(row))


if __name__ == "__main__":
    main()
