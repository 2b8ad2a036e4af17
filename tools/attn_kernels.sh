#!/bin/bash
# Runs on the GPU box: RecAttn2d parity tests, a kernel trace of the RecNeXt-A3 forward (per-kernel medians) and the A3 bench line.
# usage: tools/attn_kernels.sh <tag> [kernel-name filter, default qkcore]
tag=${1:-attn}; filt=${2:-qkcore}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_recconv_gpu.py -q -m gpu -k "recattn or linear_attention" > $out/attn.txt 2>&1 || { tail -30 $out/attn.txt; exit 1; }
tail -2 $out/attn.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --model recnext_a3 --steps 20 --warmup 10 --no-cpu-baseline > $out/kt.log 2>&1 || { tail -20 $out/kt.log; exit 1; }
python3 - "$out" "$filt" <<EOF
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/kt/*/*_kernel_trace.csv")[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if filt == "all" or filt in n:
        g[(n[:90], r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:24]:
    v.sort()
    print(f"{sum(v)/31:8.1f} us/fwd  n/fwd {len(v)/31:5.1f}  median {v[len(v)//2]:7.1f} us  {k}")
EOF
timeout -k 10 200 python bench.py --model recnext_a3 --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('A3', r['value'], r['ms_per_step'])"
