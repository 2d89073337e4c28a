"""Summarise a rocprofv3 --kernel-trace --output-format csv trace: per kernel calls / average / min / max (ns)."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print("Name,Calls,AverageNs,MinNs,MaxNs,TotalNs,Percentage")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'"{k[:100]}",{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)},{sum(v)},{100 * sum(v) / tot:.2f}')
