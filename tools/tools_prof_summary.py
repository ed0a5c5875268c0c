#!/usr/bin/env python3
"""Condense one tools_prof.sh output directory into a text summary for profiles/."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
print("command: rocprofv3 --kernel-trace --stats -- python bench.py --pmc off --no-cpu-baseline --no-secondary --no-autotune " + " ".join(sys.argv[2:]))
for leg in ("stats", "pmc_fetch", "pmc_write"):
    log = os.path.join(out, leg + ".log")
    if os.path.exists(log):
        lines = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            rl = d["roofline"]
            print(f"bench line under {leg}: value={d['value']} Mcell-steps/s, avg_launch_ms={rl.get('avg_launch_ms')}, "
                  f"steady_state_value={rl.get('steady_state_value')}, launch_shape={rl.get('launch_shape')}")
print()
print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
for f in glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        print(f"{row['Name'][:90]:90s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):12.0f} "
              f"min_ns={row['MinNs']:>10s} max_ns={row['MaxNs']:>10s} pct={row['Percentage']}")
print()
print("== steady state from the kernel trace: last 20 dispatches of each pass kernel ==")
for f in glob.glob(os.path.join(out, "stats", "*", "*kernel_trace.csv")):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_bulk" in row["Kernel_Name"] or "k_pass" in row["Kernel_Name"]:
            per[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in per.items():
        w = v[-20:]
        print(f"{k[:90]:90s} n={len(v):4d} last{len(w)}_avg_ns={sum(w)/len(w):12.0f}")
print()
print("(HBM traffic of the same kernel and launch shape: collected by bench.py itself, --pmc live)")
