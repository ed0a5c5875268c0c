#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools_isa.py file.s <mangled-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = None
for n, l in enumerate(lines):
    if re.match(r"^[A-Za-z_][\w$.]*:", l) and pat in l and not l.startswith(".L"):
        start = n
        break
if start is None:
    sys.exit("kernel not found")
ins = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
print(lines[start], "total", len(ins))
grp = collections.Counter()
for k, v in c.items():
    if k.startswith(("v_mov", "v_accvgpr", "v_pk_mov")): grp["moves"] += v
    elif k.startswith(("global_load", "buffer_load")): grp["vmem_load"] += v
    elif k.startswith(("global_store", "buffer_store")): grp["vmem_store"] += v
    elif k.startswith("scratch"): grp["scratch"] += v
    elif k.startswith("ds_"): grp["lds"] += v
    elif k.startswith("s_waitcnt"): grp["waitcnt"] += v
    elif k.startswith("s_nop"): grp["s_nop"] += v
    elif k.startswith("s_"): grp["salu"] += v
    elif k.startswith("v_"): grp["valu_other"] += v
    else: grp["other"] += v
print(dict(grp))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 22):
    print(f"   {k:28s} {v}")
