#!/bin/bash
# usage: tools_pmc.sh <tag> "<counters>" <bench --pmc-child args>   (one rocprofv3 --pmc pass; per-kernel means)
tag=$1; ctr=$2; shift 2
out=/root/repo/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 /root/repo/bench.py --pmc-child "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bulk" in r["Kernel_Name"] or "k_pass" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    vv = v[len(v)//2:]
    print(f"{k:60s} {c:26s} n={len(v):3d} mean={sum(vv)/len(vv):16.1f}")
PY
