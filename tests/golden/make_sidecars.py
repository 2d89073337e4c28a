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

# first-pass order of the inverse root finder (Flow.set_rootfinder_order), where a same-box A/B on the GPU decided it
# (profiles/r6/ab_centre.jsonl: trained_c4 inverts in 10.02 ms with the third-order first pass, 9.66 ms with the fourth-order one)
ROOTFINDER_FIRST_ORDER = {"trained_c4": 4}

for name, spec in TRAINED.items():
    if not spec["cfg"].get("condition"):
        continue
    fx = np.load(os.path.join(HERE, name + ".npz"))
    f = fx["test_feat"].astype(np.float64)
    ms = runtime.quantise_feature_ms(float((f * f).mean()))
    side = {"feature_mean_square": ms}
    if name in ROOTFINDER_FIRST_ORDER:
        side["rootfinder_first_order"] = ROOTFINDER_FIRST_ORDER[name]
    with open(os.path.join(HERE, name + ".pth.rnf.json"), "w") as fh:
        json.dump(side, fh)
    print(name, side)
