#!/usr/bin/env python3
"""Scan a hipcc -S listing for the hazard the compiler cannot see: an SGPR written by a VALU instruction (v_readlane_b32,
v_readfirstlane_b32, v_cmp_* to an SGPR pair ...) and read by a vector-memory instruction inside an inline-asm block fewer
than 5 instructions later; and (2) a VGPR that is the destination of a vector-memory LOAD issued from an inline-asm block and is
read or overwritten by any other instruction before an s_waitcnt vmcnt(N) that covers the load (loads, stores and LDS-DMA complete
in issue order; the register allocator believes an asm output is there at once and may copy it: rcx_cplbwd.hip met exactly that).
usage: check_asm_hazards.py file.s"""
import re
import sys

raw_lines = open(sys.argv[1]).read().split("\n")
lines = []                      # .rept N ... .endr blocks (inline asm) expanded
i = 0
while i < len(raw_lines):
    m = re.match(r"\s*\.rept\s+(\d+)", raw_lines[i])
    if m:
        j = i + 1
        while not raw_lines[j].strip().startswith(".endr"):
            j += 1
        lines.extend(raw_lines[i + 1:j] * int(m.group(1)))
        i = j + 1
    else:
        lines.append(raw_lines[i])
        i += 1
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
    if op.startswith("s_") and not op.startswith(("s_cmp", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_setprio", "s_sleep", "s_endpgm")):
        # a scalar-ALU write replaces the VALU-written value: what a later vector-memory instruction reads is the SALU result (no wait states needed)
        d = args.split(",")[0].strip()
        wr = set()
        mm = re.match(r"s(\d+)$", d)
        if mm:
            wr.add(int(mm.group(1)))
        mm = re.match(r"s\[(\d+):(\d+)\]$", d)
        if mm:
            wr.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        if wr:
            recent = [(sg - wr, a) for (sg, a) in recent if sg - wr]
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        d = args.split(",")[0].strip()
        mm = re.match(r"s(\d+)$", d)
        if mm:
            recent.append(({int(mm.group(1))}, 0))


def vregs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store", "scratch_load", "scratch_store")
bad2 = 0
issued = 0                     # vector-memory operations issued so far in this function
flight = {}                    # vgpr -> (issue index, line) of the asm load that will write it
in_asm = False
func = "?"
for i, l in enumerate(lines):
    t = l.strip()
    if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", t) and not t.startswith(".L") and not t.startswith("BB"):
        func, issued, flight = t.rstrip(":"), 0, {}
        continue
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    m = re.match(r"([a-z_0-9]+)\s*(.*)", t)
    if not m or t.startswith(".") or t.startswith(";"):
        continue
    op, args = m.group(1), m.group(2).split(";")[0]
    if op == "s_endpgm":
        issued, flight = 0, {}
        continue
    if op == "s_waitcnt":
        mm = re.search(r"vmcnt\((\d+)\)", args)
        if mm:
            done = issued - int(mm.group(1))
            flight = {r: (k, ln) for r, (k, ln) in flight.items() if k > done}
        continue
    touched = vregs(args)
    if op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32"):
        # a packed source with op_sel == op_sel_hi reads ONE half of its register pair (the splat / pick forms)
        body = re.split(r"\s+op_sel", args)[0]
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", body)]
        def sel(name, default):
            mm = re.search(name + r":\[([01,]+)\]", args)
            v = [int(c) for c in mm.group(1).split(",")] if mm else []
            return v + [default] * (3 - len(v))
        lo, hi = sel("op_sel", 0), sel("op_sel_hi", 1)
        touched = vregs(ops[0])
        for k, o in enumerate(ops[1:4]):
            mm = re.match(r"v\[(\d+):(\d+)\]$", o)
            if mm:
                base = int(mm.group(1))
                touched.update({base + lo[k], base + hi[k]})
            else:
                touched.update(vregs(o))
    hit = touched & set(flight)
    if hit:
        bad2 += 1
        r0 = sorted(hit)[0]
        print(f"{func}: line {i + 1}: {t}   <- v{r0} is the destination of the asm load at line {flight[r0][1]}, still in flight")
        for r in hit:
            del flight[r]
    if op.startswith(VMEM):
        issued += 1
        if in_asm and "load" in op and " lds" not in args:
            dst = args.split(",")[0]
            for r in vregs(dst):
                flight[r] = (issued, i + 1)
print("hazards:", bad, "in-flight register uses:", bad2)
sys.exit(1 if bad or bad2 else 0)
