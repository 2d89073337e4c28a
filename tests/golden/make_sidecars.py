"""Writes the feature-scale sidecars (``<ckpt>.rnf.json``, rotationnormflow_amd/harness.py) of the conditional reference-trained checkpoints:
the mean square of the feature rows the checkpoint was trained / is evaluated on (the fixture's own ``test_feat``), quantised as the packers see
it.  Pure numpy -- no GPU, no reference import.   python tests/golden/make_sidecars.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from rotationnormflow_amd import runtime  # noqa: E402
from tests.golden.trained_cases import TRAINED  # noqa: E402

for name, spec in TRAINED.items():
    if not spec["cfg"].get("condition"):
        continue
    fx = np.load(os.path.join(HERE, name + ".npz"))
    f = fx["test_feat"].astype(np.float64)
    ms = runtime.quantise_feature_ms(float((f * f).mean()))
    with open(os.path.join(HERE, name + ".pth.rnf.json"), "w") as fh:
        json.dump({"feature_mean_square": ms}, fh)
    print(name, ms)
