#!/bin/bash
# The round's measurement set on one GPU box:  tools/prof_round.sh <tag>   (files land in gpurun_out/<tag>/)
#   1. the driver's command (python bench.py --steps 20 --warmup 5)            -> bench_default.json
#   2. the same workloads at 400 steps                                         -> bench_400.json
#   3. rocprofv3 --kernel-trace --stats of the driver's command with the launch shapes of (1) and the tuner off (its
#      trial launches would be averaged into the per-kernel statistics)        -> stats_default.txt
set -o pipefail
out=/root/repo/gpurun_out/$1
mkdir -p $out
cd /root/repo
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { echo bench failed; tail -5 $out/bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --steps 400 --no-cpu-baseline > $out/bench_400.json 2> $out/bench_400.err || { echo bench 400 failed; tail -5 $out/bench_400.err; exit 1; }
python tools/show_bench.py $out/bench_default.json $out/bench_400.json > $out/bench_summary.txt 2>&1
cat $out/bench_summary.txt
shapes=$(python3 - $out/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["roofline"]
keys = ("band_rows", "waves_per_level_group", "edge_strip_band_rows", "waves_side_by_side", "xcd_map", "filler_band_rows", "filler_bands_per_strip", "zone_tiles_fused")
for s in (d["launch_shape"], dict(d["steady_state"]["launch_shape"], pass_steps=d["steady_state"]["steps_per_launch"])):
    print("--shape", ":".join(str(int(s[k])) for k in ("pass_steps",) + keys), end=" ")
PY
)
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 /root/repo/bench.py --steps 20 --warmup 5 --pmc off \
    --no-secondary --no-cpu-baseline --no-autotune $shapes > $out/prof_bench.json 2> $out/prof.err ) || { echo profile failed; tail -5 $out/prof.err; exit 1; }
python3 - $out <<'PY' > $out/stats_default.txt
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --pmc off --no-secondary --no-cpu-baseline --no-autotune <shapes of bench_default.json>")
for r in csv.DictReader(open(f[0])):
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.2f} min_us {float(r['MinNs'])/1e3:10.2f} max_us {float(r['MaxNs'])/1e3:10.2f} pct {r['Percentage']}")
PY
head -12 $out/stats_default.txt
