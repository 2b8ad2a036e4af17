#!/bin/bash
# Run ON THE GPU BOX: round-6 A/B session -- config 5 with the XCD-aware tile order, k_recconv_cpl14 reload form at 1024 waves (RCX_CPL14_RL=1), M3 bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06a
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline > "$OUT/m3_bench.json" 2> "$OUT/m3_bench.err"
RCX_CPL14_RL=1 python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline > "$OUT/m3_bench_rl1.json" 2>> "$OUT/m3_bench.err"
python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --resolution 512 --batch 32 > "$OUT/m3_512_bench.json" 2>> "$OUT/m3_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt512" -- python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --resolution 512 --batch 32 > "$OUT/kt512.log" 2>&1
f=$(ls $OUT/kt512/*/*kernel_stats.csv | head -1); grep "rcx::" "$f" | cut -c1-220 > "$OUT/m3_512_kernel_stats.csv"
rm -rf "$OUT/kt512"
python3 - <<PY
import json
for f in ("m3_bench.json","m3_bench_rl1.json","m3_512_bench.json"):
    d=json.loads(open("$OUT/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    for k in d["token_mixers"]["per_kernel"]: print("    ", str(k)[:200])
PY
head -12 "$OUT/m3_512_kernel_stats.csv"
