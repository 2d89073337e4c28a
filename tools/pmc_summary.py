#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes for one kernel: mean counter value per dispatch (over the dispatches of the LARGEST grid, i.e.
the benchmark launches, not the parity spot check), plus the HBM traffic per launch with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request: x2; both counters are in KiB).

    python tools/pmc_summary.py <out.csv> <out.json> <kernel substring> <rocprofv3 output dir> [more dirs ...]

The JSON records the hash of the kernel sources (rotationnormflow_amd.build.source_hash) and the rotations per launch
(RNF_PMC_ROTATIONS, default 2^20): bench.py replays a committed summary only when both match the running build / batch.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_csv, out_json, needle = sys.argv[1:4]
    rows = defaultdict(list)          # counter -> [(grid, kernel name, value)]
    meta = {}
    for d in sys.argv[4:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            info = {}
            with open(path, newline="") as fh:
                for r in csv.DictReader(fh):
                    if needle not in r["Kernel_Name"]:
                        continue
                    key = (r["Dispatch_Id"], r["Counter_Name"])
                    per_dispatch[key] += float(r["Counter_Value"])
                    info[r["Dispatch_Id"]] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]),
                                              r["Kernel_Name"])
            for (disp, name), v in per_dispatch.items():
                rows[name].append((info[disp][0], info[disp][4], v))
                meta[info[disp][4]] = info[disp]
    # the instantiation that did the work: largest grid, and among equal grids the largest counter total (the range guard enqueues an
    # early-exit launch of the exact-fp32 instantiation behind every split-precision launch; it must not dilute the means)
    with open(out_csv, "w") as fh:
        fh.write("counter,mean_per_dispatch,dispatches,grid_size,kernel\n")
        summary = {}
        chosen = None
        for name in sorted(rows):
            totals = defaultdict(float)
            for g, k, v in rows[name]:
                totals[(g, k)] += v
            g, k = max(totals, key=lambda t: (t[0], totals[t]))
            chosen = chosen or k
            vals = [v for gg, kk, v in rows[name] if (gg, kk) == (g, k)]
            summary[name] = sum(vals) / len(vals)
            fh.write(f"{name},{summary[name]:.1f},{len(vals)},{g},\"{k}\"\n")
    meta = {chosen: meta[chosen]} if chosen else {}
    from rotationnormflow_amd.build import source_hash
    out = {"kernel": next(iter(meta.values()))[4] if meta else None, "counters": summary, "csrc_sha": source_hash(),
           "rotations_per_launch": int(os.environ.get("RNF_PMC_ROTATIONS", 1 << 20))}
    if meta:
        g, wg, vgpr, lds, _ = next(iter(meta.values()))
        out.update(workgroup_size=wg, vgpr_count=vgpr, lds_block_size=lds)
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        out["hbm_fetch_bytes_per_launch"] = summary["FETCH_SIZE"] * 1024 * 2
        out["hbm_write_bytes_per_launch"] = summary["WRITE_SIZE"] * 1024
        out["hbm_bytes_per_launch"] = out["hbm_fetch_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    with open(out_json, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
