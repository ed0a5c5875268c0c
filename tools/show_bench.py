#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: python tools/show_bench.py file.json"""
import json, sys
for path in sys.argv[1:]:
  d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
  print("==", path)
  for r in [d] + d.get("secondary", []):
      rl = r["roofline"]
      print(r["config"]["grid"], r["config"]["materials"], r["config"]["boundary"], "steps", r["steps"], "value", r["value"],
            "| steady", rl.get("steady_state_value"), "launch_ms", rl.get("avg_launch_ms"), "incl_gaps", rl.get("avg_launch_ms_incl_gaps"),
            "| frac", rl.get("frac"), "overfetch", rl.get("overfetch"), "valu", rl.get("valu_frac"), "alg x_peak", rl["algorithmic"]["x_peak"],
            "| shape", rl.get("launch_shape"), rl.get("traffic_note", ""))
  if "cpu_baseline" in d:
      print("cpu:", d["cpu_baseline"])
