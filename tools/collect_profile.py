#!/usr/bin/env python3
"""Copy the parts of a tools/prof_*.sh output directory (gpurun_out/prof_<tag>) that are worth
committing into profiles/<tag>: kernel stats, the lasso kernels' PMC rows (condensed columns),
summary.txt and hbm_traffic.json.   usage: tools/collect_profile.py <tag>"""
import csv, glob, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
for name in ("summary.txt", "hbm_traffic.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(os.path.join(dst, os.path.basename(d) + ".csv"), "w", newline="") as out:
            w = csv.writer(out)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for r in csv.DictReader(open(f)):
                if "lasso" in r["Kernel_Name"]:
                    w.writerow([r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]])
print("collected into", dst, sorted(os.listdir(dst)))
