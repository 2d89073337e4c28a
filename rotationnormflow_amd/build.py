"""Build librnf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m rotationnormflow_amd.build`"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rnf_api.hip")
OUT = os.path.join(HERE, "librnf_hip.so")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".hip"))) + [
    os.path.join(os.path.dirname(HERE), "include", "rnf_hip.h")]


def source_hash() -> str:
    """Hash of the kernel sources (csrc/*.h, *.hip, include/rnf_hip.h): recorded beside committed profiler summaries so that bench.py
    only replays counters that were collected from the sources the running library was built from."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: packed f32 VALU (v_pk_*) beside MFMAs is slower than scalar (MI355X_MICROARCH.md, "price of one
    # filler beside MFMAs"), and SLP packing would fuse the hand-placed per-MFMA slices of the segment math back together
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Xarch_host", "-mf16c", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
