#!/usr/bin/env python3
"""tests/golden/param_order.json: for every distinct flow structure of the fixture tables, the order in which the REFERENCE's
``Flow`` yields ``named_parameters()`` (= the numbering of ``optim.Adam(flow.parameters())``'s state, agent.py:23,143,193-196) and
``state_dict()`` keys, with the shapes.  Names and integers only; run in the build container:

    python tests/golden/make_param_order.py
"""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference tree not present; this fixture can only be regenerated in the build container")
sys.path[:0] = [os.path.join(REPO, "oracle", "stubs"), REF, REPO]

import flow.flow as ref_flow_mod  # noqa: E402

assert ref_flow_mod.__file__.startswith(REF), ref_flow_mod.__file__

from rotationnormflow_amd.configs import make_config  # noqa: E402
from tests.golden.cases import CASES  # noqa: E402
from tests.golden.trained_cases import TRAINED, TRAJ  # noqa: E402


def main():
    out, seen = {}, {}
    for name, spec in list(CASES.items()) + list(TRAINED.items()) + list(TRAJ.items()):
        sig = json.dumps(spec["cfg"], sort_keys=True)
        if sig in seen:
            continue
        seen[sig] = name
        with contextlib.redirect_stdout(io.StringIO()):
            fl = ref_flow_mod.Flow(make_config(**spec["cfg"]))
        out[name] = {"cfg": spec["cfg"],
                     "parameters": [[k, list(p.shape)] for k, p in fl.named_parameters()],
                     "state_dict": [[k, list(v.shape)] for k, v in fl.state_dict().items()]}
    with open(os.path.join(HERE, "param_order.json"), "w") as fh:         # one line per structure
        fh.write("{\n")
        keys = sorted(out)
        for i, k in enumerate(keys):
            fh.write(json.dumps(k) + ": " + json.dumps(out[k], separators=(",", ":"), sort_keys=True) + (",\n" if i + 1 < len(keys) else "\n"))
        fh.write("}\n")
    print(f"{len(out)} structures -> param_order.json")


if __name__ == "__main__":
    main()
