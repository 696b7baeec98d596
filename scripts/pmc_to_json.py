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
    lines = open(path).read().splitlines()[1:]
    for line in lines:
        # kernel names (rocPRIM templates) contain commas: the four numeric/text fields are last
        kernel, dispatches, counter, _total, per = line.rsplit(",", 4)
        k = kernel.replace("void ", "").replace("wm::", "").split("<")[0]
        d = out.setdefault(k, {"dispatches": int(dispatches)})
        unit = "_kb" if counter in ("FETCH_SIZE", "WRITE_SIZE") else ""
        d[counter + unit + "_per_dispatch"] = float(per)
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(out.get("k_nn_grid", {}), indent=1))
