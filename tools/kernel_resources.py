#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table of the shipped library: rebuilds with -Rpass-analysis=kernel-resource-usage and prints
one line per kernel instantiation.  `python tools/kernel_resources.py [log]` (no GPU needed)."""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        txt = open(sys.argv[1]).read()
    else:
        txt = subprocess.run([sys.executable, "-m", "rotationnormflow_amd.build", "--force", "-v"], cwd=ROOT, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split("\n")[0].split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"VGPRs Spill"), g(r"SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    names = [r[0] for r in rows]
    if filt:
        names = subprocess.run([filt] + names, capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':100s} VGPR AGPR vspill sspill scratch occ")
    for r, d in zip(rows, names):
        d = d.replace("rnf::", "").replace("void ", "")
        d = re.sub(r"\(.*\)$", "", d)
        print(f"{d[:100]:100s} {r[1]:4d} {r[2]:4d} {r[3]:6d} {r[4]:6d} {r[5]:7d} {r[6]:3d}")


if __name__ == "__main__":
    main()
