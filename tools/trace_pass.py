#!/usr/bin/env python3
"""Schedule of one level-split pass launch from per-workgroup time stamps (profiling build):

    make -f tools/Makefile.trace
    FDTD2D_LIB=build/trace/libfdtd2d.so python tools/trace_pass.py [grid] [materials] [band_rows] [waves]

Prints, for the last 16-step launch: its span in shader cycles, how long zone tiles / edge strips /
plain strips live, how many workgroups are resident over time and when the first-round slots free up.
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdtd2d_amd as fd  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
br = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nw = int(sys.argv[4]) if len(sys.argv) > 4 else 0
path = os.path.join(tempfile.gettempdir(), f"fdtd2d_trace_{os.getpid()}.txt")
os.environ["FDTD2D_TRACE_FILE"] = path
with fd.Engine(g, g, dtype=np.float32) as e:
    e.set_materials()
    if br:
        e.set_shape((br, nw))
    N = int(os.environ.get("TRACE_N", "16"))      # pass length to look at (16, 20 or 24)
    e.prepare(N)
    e.run(N * 4)
    ms = np.sort(e.time_launches(16, N))
    e.sync()
    shape = (e.info(19), e.info(20))
t = np.loadtxt(path, dtype=np.uint64).reshape(-1, 9)
os.remove(path)
t0, t1, kind, hw = t[:, 1].astype(np.int64), t[:, 2].astype(np.int64), t[:, 3].astype(int), t[:, 4].astype(np.int64)
xcc = hw >> 32
hw = hw & 0xffffffff
for x in np.unique(xcc):        # s_memtime is a per-XCD counter: align every XCD at its first workgroup
    m = xcc == x
    base = t0[m].min()
    t0[m] -= base
    t1[m] -= base
span = t1.max()
print(f"{g}x{g} run({os.environ.get('TRACE_N', '16')}): shape (band rows, waves) = {shape}, launch {np.median(ms) * 1e3:.1f} us by HIP events, "
      f"{len(t)} workgroups, span {span} cycles ({span / (np.median(ms) * 1e3):.0f} cycles/us)")
for k, name in ((0, "zone tiles"), (1, "edge strips"), (2, "plain strips")):
    m = kind == k
    if m.any():
        d = (t1 - t0)[m]
        print(f"  {name:13s} n={m.sum():5d} lifetime mean {d.mean():9.0f} min {d.min():9d} max {d.max():9d} cycles; "
              f"starts: {np.percentile(t0[m], [0, 50, 100]).astype(int)}  ends: {np.percentile(t1[m], [0, 50, 100]).astype(int)}")
# barrier waits per wave role (plain strips): which wave do the others wait for?
m = kind == 2
if m.any():
    life = (t1 - t0)[m].astype(np.float64)
    for w in range(4):
        bw = t[:, 5 + w].astype(np.float64)[m]
        print(f"  plain strips, wave {w}: waits at the tick barrier {100 * (bw / life).mean():5.1f} % of the workgroup's lifetime")
# where the hardware puts wave 0 of a workgroup: SIMD, wave slot, workgroup slot (HW_ID bits 5:4, 3:0, 19:16)
for name, sh, msk in (("SIMD of wave 0", 4, 3), ("wave slot of wave 0", 0, 15), ("workgroup slot (TG_ID)", 16, 15)):
    vals, cnt = np.unique((hw[kind == 2] >> sh) & msk, return_counts=True)
    print(f"  plain strips, {name}: " + ", ".join(f"{int(v)}: {int(c)}" for v, c in zip(vals, cnt)))
# resident workgroups over time
ts = np.linspace(0, span, 21)
res = [int(((t0 <= x) & (t1 > x)).sum()) for x in ts]
print("  resident workgroups at 0,5,..100 % of the span:", res)
busy = (t1 - t0).sum()
print(f"  sum of lifetimes / (span x 1024 slots) = {busy / (span * 1024.0):.3f}")
cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1)
per = {}
for x, c, a, b in zip(xcc, cu, t0, t1):
    per.setdefault((int(x), int(c)), []).append((a, b))
life = np.array([sum(b - a for a, b in v) for v in per.values()])
print(f"  {len(per)} (XCC, CU) pairs seen; workgroup-cycles per CU: min {life.min()} median {int(np.median(life))} max {life.max()}")
