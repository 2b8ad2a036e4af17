#!/bin/bash
# usage: tools/ab_blocks.sh <set> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]   -- per-block times of tools/bench_blocks.py (fresh inputs, bf16) under each environment
set=$1; shift
for e in "" "$@"; do
  echo "== ${e:-default}"
  env $e timeout -k 10 150 python tools/bench_blocks.py --sets $set --dtypes bf16 --fresh 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        r = json.loads(l)
        t = [f"{k}={v}" for k, v in r.items() if k not in ("set", "shape", "level", "dtype", "timing", "taps", "plan", "mode")]
        print(r["shape"], "L%d" % r["level"], " ".join(t), "|", r["plan"][:70])
'
done
