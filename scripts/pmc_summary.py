"""Summarises rocprofv3 --pmc CSVs (one directory per pass) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernel_filter = sys.argv[2] if len(sys.argv) > 2 else "k_"
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for path in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if kernel_filter not in k:
            continue
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in tot:
    print("==", k)
    for c, v in sorted(tot[k].items()):
        n = len(disp[(k, c)])
        print(f"  {c:32s} {v / n:20.1f}  (per dispatch, {n} dispatches)")
for log in sorted(glob.glob(os.path.join(root, "*.log"))):
    lines = [l for l in open(log).read().splitlines() if "rays=" in l]
    if lines:
        print(os.path.basename(log), lines[-1])
