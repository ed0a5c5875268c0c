#!/bin/bash
# Profiling recipe behind profiles/r02_*_stats.txt (run on the GPU box through gpurun):
#   tools_prof.sh <tag> <bench args...>
# rocprofv3 --kernel-trace --stats over `python bench.py --pmc off --no-cpu-baseline --no-secondary
# --no-autotune <args>` (fixed launch shape: give the shape of the bench line being documented with
# --band-rows/--waves/--edge-rows; the tuner's trial launches would otherwise be averaged into the
# per-kernel statistics).  HBM traffic is NOT collected here: bench.py measures it itself (--pmc live,
# separate rocprofv3 --pmc passes in child processes).  Summary -> gpurun_out/prof_<tag>/summary.txt.
set -o pipefail
tag=$1; shift
out=/root/repo/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 /root/repo/bench.py --pmc off --no-cpu-baseline --no-secondary --no-autotune "$@" > $out/stats.log 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
python3 /root/repo/tools/tools_prof_summary.py $out "$@" > $out/summary.txt
cat $out/summary.txt
