#!/bin/bash
# Copy the records tools/collect_all_r06.sh left under gpurun_out/ into profiles/ (tracked).  Run in the build container after the gpurun call.
set -e
cd "$(dirname "$0")/.."
for t in r06 r06_512 r06_a3 r06_m1 r06_m5; do cp gpurun_out/profiles_$t/* profiles/; done
cp gpurun_out/r06_gpu_tests.txt profiles/
cp gpurun_out/train_r06/m3_train_step_kernels.csv profiles/r06_train_m3_step_kernels.csv
cp gpurun_out/train_r06/m3_train_step.jsonl profiles/r06_train_m3_step.jsonl
cp gpurun_out/train_r06/bwd56_kernels.csv profiles/r06_train_bwd56_kernels.csv
cp gpurun_out/train_r06/bwd28_kernels.csv profiles/r06_train_bwd28_kernels.csv
cp gpurun_out/train_r06/bwd56_traffic.json profiles/r06_train_bwd56_traffic.json 2>/dev/null || true
cp gpurun_out/train_r06/bwd28_traffic.json profiles/r06_train_bwd28_traffic.json 2>/dev/null || true
cp gpurun_out/train_r06/blocks_fwd_bwd.jsonl profiles/r06_train_blocks_fwd_bwd.jsonl
cp gpurun_out/train_r06/blocks_fwd_bwd_batch256.jsonl profiles/r06_train_blocks_fwd_bwd_batch256.jsonl
python3 - <<'PY'
import json
from recnext_amd import build
fp = build.source_fingerprint()
for tag in ("r06", "r06_512", "r06_a3", "r06_m1", "r06_m5"):
    t = json.load(open(f"profiles/{tag}_traffic.json"))
    print(tag, "traffic fingerprint", "OK" if t["library_sources_sha256"] == fp else "STALE")
PY
