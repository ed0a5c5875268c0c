#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box via gpurun):
#   tools_prof.sh <tag> <bench args...>
# 1) rocprofv3 --kernel-trace --stats   2) --pmc FETCH_SIZE   3) --pmc WRITE_SIZE
set -o pipefail
tag=$1; shift
out=/root/repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/stats.log 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $out/pmc_write.log; exit 1; }
find $out -name "*.csv" | head -20
