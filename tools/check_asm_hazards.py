#!/usr/bin/env python3
"""Scan a hipcc -S listing for the hazard the compiler cannot see: an SGPR written by a VALU instruction (v_readlane_b32,
v_readfirstlane_b32, v_cmp_* to an SGPR pair ...) and read by a vector-memory instruction inside an inline-asm block fewer
than 5 instructions later.  usage: check_asm_hazards.py file.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
bad = 0
recent = []          # (sgpr set, age)
in_asm = False
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    m = re.match(r"([a-z_0-9]+)\s+(.*)", t)
    if not m or t.startswith(".") or t.startswith(";"):
        continue
    op, args = m.group(1), m.group(2)
    # age the window by one instruction (s_nop N counts N+1)
    step = 1
    if op == "s_nop":
        step = int(args.split()[0], 0) + 1
    recent = [(s, a + step) for (s, a) in recent if a + step < 6]
    if in_asm and (op.startswith("buffer_") or op.startswith("global_")):
        used = set()
        for a, b in re.findall(r"s\[(\d+):(\d+)\]", args):
            used.update(range(int(a), int(b) + 1))
        used.update(int(x) for x in re.findall(r"\bs(\d+)\b", args))
        for s, a in recent:
            if s & used and a <= 5:
                bad += 1
                print(f"line {i + 1}: {t}   <- SGPR {sorted(s & used)} written by VALU {a} instruction(s) earlier")
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        d = args.split(",")[0].strip()
        mm = re.match(r"s(\d+)$", d)
        if mm:
            recent.append(({int(mm.group(1))}, 0))
print("hazards:", bad)
sys.exit(1 if bad else 0)
