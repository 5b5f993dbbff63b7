"""Per-launch averages of rocprofv3 --pmc counters for one kernel (default: ms_search_kernel).
usage: summarize_pmc.py <counter_collection.csv>... ; prints one JSON object."""
import csv
import json
import sys
from collections import defaultdict

kernel = "ms_search_kernel"
tot, disp = defaultdict(float), defaultdict(set)
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                disp[row["Counter_Name"]].add((path, row["Dispatch_Id"]))
print(json.dumps({c: {"per_launch": tot[c] / max(1, len(disp[c])), "launches": len(disp[c])} for c in sorted(tot)}, indent=1))
