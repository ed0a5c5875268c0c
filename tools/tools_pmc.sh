#!/bin/bash
# usage: tools_pmc.sh <tag> "<counters>" <bench args>
tag=$1; ctr=$2; shift 2
out=/root/repo/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_bulk" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    vv = v[len(v)//2:]
    print(f"{k:28s} n={len(v):3d} mean={sum(vv)/len(vv):16.1f}")
PY
