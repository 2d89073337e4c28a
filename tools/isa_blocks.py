#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -save-temps .s file: per basic block the number of VALU, transcendental,
matrix, LDS, vector-memory and scalar instructions (diagnostic for the VALU-issue-bound stack kernel).

    python tools/isa_blocks.py file.s <mangled-kernel-name-substring> [--min N]
"""
import re
import sys

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64")


def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E"))
    blocks, cur = [], dict(name="entry", line=start, valu=0, trans=0, mfma=0, ds=0, vmem=0, salu=0, mov=0, cnd=0, bar=0, perm=0)
    for i in range(start + 1, len(lines)):
        l = lines[i]
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section"):      # end of the function (early s_endpgm exits are not)
            break
        if s.startswith(".LBB") or re.match(r"^; %bb\.\d+", s):
            blocks.append(cur)
            cur = dict(name=s.split()[0] if s.startswith(".LBB") else s.split()[1], line=i, valu=0, trans=0, mfma=0, ds=0, vmem=0, salu=0, mov=0, cnd=0, bar=0, perm=0)
            continue
        if not l.startswith("\t") or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith(TRANS):
                cur["trans"] += 1
            if op.startswith("v_mov") or op.startswith("v_pk_mov") or op.startswith("v_accvgpr"):
                cur["mov"] += 1
            if op.startswith("v_cndmask"):
                cur["cnd"] += 1
            if op.startswith("v_permlane") or "dpp" in s:
                cur["perm"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op == "s_barrier":
            cur["bar"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
    blocks.append(cur)
    tot = {k: sum(b[k] for b in blocks) for k in ("valu", "trans", "mfma", "ds", "vmem", "salu", "mov", "cnd")}
    print("total", tot)
    for b in blocks:
        if b["valu"] + b["mfma"] + b["ds"] >= minn:
            print(f"{b['name']:<14} @{b['line'] - start:<6} valu {b['valu']:<5} (trans {b['trans']:<3} mov {b['mov']:<3} cnd {b['cnd']:<3} perm {b['perm']:<2}) "
                  f"mfma {b['mfma']:<3} ds {b['ds']:<3} vmem {b['vmem']:<3} salu {b['salu']:<3} bar {b['bar']}")


if __name__ == "__main__":
    main()
