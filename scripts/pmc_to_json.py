#!/usr/bin/env python3
"""gpurun_out/<tag>/{fetch,write,sq,tcc}_summary.csv (scripts/gpu_pmc.sh) -> profiles/pmc_latest.json"""
import csv
import json
import os
import sys

tag_dir = sys.argv[1]
out = {}
for name in ("fetch", "write", "sq", "tcc"):
    path = os.path.join(tag_dir, name + "_summary.csv")
    if not os.path.exists(path):
        continue
    for row in csv.DictReader(open(path)):
        k = row["kernel"].replace("void ", "").replace("wm::", "").split("<")[0]
        d = out.setdefault(k, {"dispatches": int(row["dispatches"])})
        unit = "_kb" if row["counter"] in ("FETCH_SIZE", "WRITE_SIZE") else ""
        d[row["counter"] + unit + "_per_dispatch"] = float(row["per_dispatch"])
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(out.get("k_nn_grid", {}), indent=1))
