#!/bin/bash
# The round's measurement set on one GPU box:  tools/prof_round.sh <outdir under gpurun_out>
#   1. the driver's command (python bench.py --steps 20 --warmup 5)          -> bench_default.json
#   2. the same workloads at 400 steps                                        -> bench_400.json
#   3. rocprofv3 --kernel-trace --stats of bench.py at the launch shapes (1) used (tools_prof.sh)
set -o pipefail
out=/root/repo/gpurun_out/$1
mkdir -p $out
cd /root/repo
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { echo bench failed; tail -5 $out/bench_default.err; exit 1; }
timeout -k 10 400 python bench.py --steps 400 --no-cpu-baseline > $out/bench_400.json 2> $out/bench_400.err || { echo bench 400 failed; tail -5 $out/bench_400.err; exit 1; }
python tools/show_bench.py $out/bench_default.json $out/bench_400.json > $out/bench_summary.txt 2>&1
cat $out/bench_summary.txt
shape() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d if sys.argv[2] == "0" else d["secondary"][int(sys.argv[2]) - 1]
s = r["roofline"]["launch_shape"]
print(f"--band-rows {s['band_rows']} --waves {s['waves_per_strip']} --edge-rows {s['edge_strip_band_rows']}")
PY
}
bash tools/tools_prof.sh r02b_16384 $(shape $out/bench_400.json 0) > /dev/null && cp gpurun_out/prof_r02b_16384/summary.txt $out/stats_16384.txt
bash tools/tools_prof.sh r02b_cfg2 --grid 4096 $(shape $out/bench_400.json 1) > /dev/null && cp gpurun_out/prof_r02b_cfg2/summary.txt $out/stats_cfg2_4096.txt
bash tools/tools_prof.sh r02b_cfg3 --grid 8192 --materials ring --steps 160 --warmup 32 $(shape $out/bench_400.json 2) > /dev/null && cp gpurun_out/prof_r02b_cfg3/summary.txt $out/stats_cfg3_8192_ring.txt
tail -n 12 $out/stats_16384.txt
