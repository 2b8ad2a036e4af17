#!/usr/bin/env python3
"""Build gate (recnext_amd/csrc/Makefile): list every kernel of the given hipcc -S listings with its register and scratch figures and fail
when a kernel the DEFAULT dispatch reaches has a private segment (registers spilled to memory).  Default dispatch = the inference
instantiations for bf16 / float32 activations of the fused kernels (k_recconv_cpt without TRAIN, k_recconv_cpl14, k_recconv_cpl7b) and the
tiled step kernels (k_upadd_cpt, k_down5_cpt, k_down7m2_cpt); the training-forward and float16 instantiations are reported only.
usage: check_scratch.py file.s [file.s ...]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
            try:
                out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout
                return out.strip().split("\n")
            except FileNotFoundError:
                continue
    except Exception:
        pass
    sys.exit("check_scratch.py: no demangler (c++filt / llvm-cxxfilt): cannot tell the gated kernels apart")


bad = 0
rows = []
for path in sys.argv[1:]:
    cur = {}
    for line in open(path):
        m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name" and "name" in cur and "private_segment_fixed_size" in cur:
            rows.append((path, cur))
            cur = {}
        if k == "name" and re.match(r"^_Z", v) is None:
            continue                                       # argument names of the metadata, not the kernel symbol
        cur[k] = v
        if all(x in cur for x in ("name", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count")):
            rows.append((path, cur))
            cur = {}
names = demangle([r[1]["name"] for r in rows])
for (path, r), name in zip(rows, names):
    priv = int(r["private_segment_fixed_size"])
    short = re.sub(r"^void rcx::", "", name)
    short = re.sub(r"\(.*$", "", short)
    fp16 = "_Float16" in short or "DF16_" in short
    # template arguments of k_recconv_cpt: <T, HALVES, MODE, PIXB, TIO, TRAIN, LV, STG>
    m = re.match(r"cpt::k_recconv_cpt<(\d+), (\d+), (\d+), (\d+), ([^,]+), (true|false), ", short)
    gated = False
    if m:
        gated = m.group(6) == "false" and not fp16
    elif re.match(r"(cpl14::k_recconv_cpl14|cpl14::k_recconv_cpl7b|cpl14::k_upadd_cpl14|cpl14::k_upadd_cpl7|cpl14::k_down5_cpl7|upcpt::k_upadd_cpt|upcpt::k_down5_cpt|upcpt::k_down7m2_cpt)<", short):
        gated = not fp16
    flag = ""
    if priv:
        flag = "  <-- SCRATCH" + (" (gated)" if gated else " (reported only)")
        bad += 1 if gated else 0
    print(f"{priv:5d} B scratch  {r['vgpr_count']:>3} VGPRs  {r['vgpr_spill_count']:>3} VGPR / {r['sgpr_spill_count']:>3} SGPR spills  {short}{flag}")
print(f"kernels: {len(rows)}  gated kernels with scratch: {bad}")
sys.exit(1 if bad else 0)
