#!/usr/bin/env python3
"""Condense one tools_prof.sh output directory into a text summary for profiles/."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
print("command: python bench.py --no-cpu-baseline " + " ".join(sys.argv[2:]))
for leg in ("stats", "pmc_fetch", "pmc_write"):
    log = os.path.join(out, leg + ".log")
    if os.path.exists(log):
        lines = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            print(f"bench line under {leg}: value={d['value']} Mcell-steps/s, roofline={json.dumps(d['roofline'])}")
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
print("== PMC (separate passes; KiB per dispatch as reported) ==")
acc = collections.defaultdict(list)
for leg in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, leg, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
tot = collections.defaultdict(dict)
for (k, c), v in sorted(acc.items()):
    # skip warm-up dispatches of the hot kernel: use the last half
    vv = v[len(v) // 2:] if len(v) > 4 else v
    mean = sum(vv) / len(vv)
    tot[k][c] = mean
    print(f"{k:70s} {c:11s} n={len(v):4d} mean={mean:14.1f} KiB")
print()
print("== HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes ==")
print("   (FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming reads: x2; WRITE_SIZE exact)")
for k, d in tot.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and d["WRITE_SIZE"] > 1000:
        rd, wr = 2 * d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
        print(f"{k:70s} read={rd/1e6:10.1f} MB write={wr/1e6:10.1f} MB total={(rd+wr)/1e6:10.1f} MB")
