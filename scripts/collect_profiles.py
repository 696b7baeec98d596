#!/usr/bin/env python3
"""gpurun_out/<tag>_{stats,pmc,detail,ndt}/ (scripts/gpu_capture_all.sh) -> profiles/<tag>_*  and
profiles/pmc_latest.json.      usage: scripts/collect_profiles.py <tag>"""
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")


def cp(a, b):
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, b))
        print("profiles/" + b)


cp(os.path.join(src, tag + "_stats", "kernel_stats.csv"), tag + "_bench_kernel_stats.csv")
cp(os.path.join(src, tag + "_stats", "kernel_stats_solo.csv"), tag + "_bench_kernel_stats_one_at_a_time.csv")
cp(os.path.join(src, tag + "_stats", "bench_line.json"), tag + "_bench_line.json")
cp(os.path.join(src, tag + "_bench_line_noprof.json"), tag + "_bench_line_noprof.json")
cp(os.path.join(src, tag + "_multimatcher_cpp.jsonl"), tag + "_multimatcher_cpp.jsonl")
cp(os.path.join(src, tag + "_batch_scaled_testscan.txt"), tag + "_batch_scaled_testscan.txt")
for name in ("fetch", "write", "sq", "tcc"):
    cp(os.path.join(src, tag + "_pmc", name + "_summary.csv"), "%s_pmc_%s_summary.csv" % (tag, name))
for name in ("sq1", "sq2", "sq3", "f64"):
    cp(os.path.join(src, tag + "_ndt", name + "_summary.csv"), "%s_pmc_ndt_%s_summary.csv" % (tag, name))
for name in ("fetch", "write"):
    cp(os.path.join(src, tag + "_other", name + "_summary.csv"), "%s_pmc_gicp_ndt_%s_summary.csv" % (tag, name))
cp(os.path.join(src, tag + "_pmc", "pmc_latest.json"), "pmc_latest.json")
tabs = {}
for f in sorted(glob.glob(os.path.join(src, tag + "_detail", "*_per_dispatch.csv"))):
    for r in csv.DictReader(open(f)):
        tabs.setdefault(int(r["iteration"]), {}).update({k: float(v) for k, v in r.items() if k != "iteration"})
if tabs:
    cols = sorted(tabs[0].keys())
    path = os.path.join(dst, tag + "_pmc_search_per_iteration.csv")
    with open(path, "w") as f:
        f.write("iteration," + ",".join(cols) + "\n")
        for i in sorted(tabs):
            f.write("%d," % i + ",".join("%.6g" % tabs[i].get(c, float("nan")) for c in cols) + "\n")
    print("profiles/" + os.path.basename(path))
