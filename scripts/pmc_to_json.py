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
out["tag"] = os.path.basename(os.path.normpath(tag_dir))
# the kernel sources this capture belongs to: bench.py drops the traffic figures when the tree differs
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    import bench
    out["csrc_sha256"] = bench.csrc_sha256()
except Exception as e:  # (never lose a capture over the stamp)
    print("no source stamp:", e)
# optional third argument: the NDT counter directory (scripts/gpu_pmc_ndt.sh): f64 flops per source point
# of the derivative kernel = (ADD + MUL + 2 FMA) wave-instructions x 64 lanes x the active-lane fraction
if len(sys.argv) > 3:
    nd = sys.argv[3]
    tot = {}
    disp = 0
    for name in ("f64", "sq2"):
        path = os.path.join(nd, name + "_summary.csv")
        if not os.path.exists(path):
            continue
        for line in open(path).read().splitlines()[1:]:
            kernel, dispatches, counter, total, _per = line.rsplit(",", 4)
            if "k_ndt_derivs" in kernel:
                tot[counter] = tot.get(counter, 0.0) + float(total)
                if counter == "SQ_INSTS_VALU_FMA_F64":
                    disp += int(dispatches)
    if "SQ_INSTS_VALU_FMA_F64" in tot and disp:
        # SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 16) is 4.0 with all 64 lanes active
        lanes = tot.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(tot.get("SQ_ACTIVE_INST_VALU", 0.0) * 64.0, 1.0) \
            if "SQ_THREAD_CYCLES_VALU" in tot else 1.0
        winst = tot.get("SQ_INSTS_VALU_ADD_F64", 0) + tot.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * tot["SQ_INSTS_VALU_FMA_F64"]
        n_points = float(os.environ.get("NDT_POINTS", "2000000"))
        out["k_ndt_derivs"] = {"dispatches": disp, "f64_wave_flop_instructions_per_dispatch": winst / disp,
                               "active_lane_fraction": lanes,
                               "f64_flops_per_source_point": winst / disp * 64.0 * lanes / n_points,
                               "source": os.path.basename(os.path.normpath(nd))}
# optional fourth argument: the GICP / NDT traffic directory (scripts/gpu_pmc_other.sh): FETCH / WRITE per
# dispatch of k_gicp_fdf (launched evaluations) and of the k_ndt_derivs variants (dispatch-weighted)
if len(sys.argv) > 4:
    od = sys.argv[4]
    acc = {}
    for name in ("fetch", "write"):
        path = os.path.join(od, name + "_summary.csv")
        if not os.path.exists(path):
            continue
        for line in open(path).read().splitlines()[1:]:
            kernel, dispatches, counter, total, _per = line.rsplit(",", 4)
            k = kernel.replace("void ", "").replace("wm::", "").split("<")[0]
            a = acc.setdefault(k, {}).setdefault(counter, [0.0, 0])
            a[0] += float(total)
            a[1] += int(dispatches)
    for k, cs in acc.items():
        d = out.setdefault(k, {})
        for counter, (total, disp) in cs.items():
            d[counter + "_kb_per_dispatch"] = total / max(disp, 1)
            d.setdefault("dispatches_traffic", disp)
        d["traffic_source"] = os.path.basename(os.path.normpath(od))
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps({k: out.get(k, {}) for k in ("k_nn_grid", "k_nn_cert")}, indent=1))
