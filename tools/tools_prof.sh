#!/bin/bash
# Profiling recipe behind profiles/ (run on the GPU box through gpurun):
#   tools_prof.sh <tag> <bench args...>
# 1) rocprofv3 --kernel-trace --stats  2) --pmc FETCH_SIZE  3) --pmc WRITE_SIZE  (separate
#    passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass, MI355X_MICROARCH.md)
# Summaries are written to gpurun_out/prof_<tag>/summary.txt; copy them into profiles/.
set -o pipefail
tag=$1; shift
out=/root/repo/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# fixed launch-shape rules: the tuner's trial launches (different band heights) would otherwise be
# averaged into the per-kernel statistics
export FDTD2D_AUTOTUNE=${FDTD2D_AUTOTUNE:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/stats.log 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $out/pmc_write.log; exit 1; }
python3 /root/repo/tools/tools_prof_summary.py $out "$@" > $out/summary.txt
cat $out/summary.txt
