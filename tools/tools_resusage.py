#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as a table."""
import re, sys, subprocess
txt = sys.stdin.read()
rows = []
cur = None
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':70s} VGPR AGPR SGPR  spillV spillS occ  LDS")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void fdtd::", "")
    if pat and not re.search(pat, d):
        continue
    print(f"{d[:70]:70s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('TotalSGPRs','?'):>4}  "
          f"{r.get('VGPR Spill', r.get('VGPRs Spill','?')):>6} {r.get('SGPR Spill', r.get('SGPRs Spill','?')):>6} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>3}  {r.get('LDS Size [bytes/block]','?')}")
