#!/bin/bash
# Run ON THE GPU BOX (via gpurun): evidence for the training path on the sources in the tree.
#   tools/collect_train.sh r06   ->  gpurun_out/train_<tag>/{blocks_fwd_bwd.jsonl, bwd56_kernels.csv, bwd28_kernels.csv, m3_train_step_kernels.csv, m3_train_step.jsonl}
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python3 tools/bench_backward.py 128 --bf16-only > "$OUT/blocks_fwd_bwd.jsonl" 2> "$OUT/blocks_fwd_bwd.err" || exit 1
cat > "$OUT/bwd_one.py" <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch, recnext_amd
dev = torch.device("cuda:0")
n, c, h, level = 128, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).to(torch.bfloat16).train()
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for i in range(12):
    y = mod(x); y.backward(g); x.grad = None
    for p in mod.parameters(): p.grad = None
torch.cuda.synchronize()
PY
for s in "64 56 4" "128 28 3"; do
  set -- $s
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt$2" -- python3 "$OUT/bwd_one.py" $1 $2 $3 > "$OUT/kt$2.log" 2>&1 || exit 1
  python3 tools/step_kernels.py "$OUT/kt$2" 12 "$OUT/bwd$2_kernels.csv" k_recconv_cpt 1 > "$OUT/bwd$2_kernels.txt"
  rm -rf "$OUT/kt$2"
done
# HBM traffic of the block kernels: PMC passes of the same block script (counters in their own runs)
for s in "64 56 4" "128 28 3"; do
  set -- $s
  rm -rf "$OUT/pmc$2"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc$2/pmc_fetch" -- python3 "$OUT/bwd_one.py" $1 $2 $3 > "$OUT/pmc$2.log" 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc$2/pmc_write" -- python3 "$OUT/bwd_one.py" $1 $2 $3 >> "$OUT/pmc$2.log" 2>&1 || exit 1
  python3 tools/pmc_kernels.py "$OUT/pmc$2" "$OUT/bwd$2_traffic.json" "forward + backward of recnext_amd.RecConv2d($1, level=$3) on 128 x $1 x $2 x $2 bf16, 12 iterations" > "$OUT/bwd$2_traffic.txt"
  rm -rf "$OUT/pmc$2"
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/ktm3" -- python3 tools/bench_train.py --which hip --batch 128 --steps 5 > "$OUT/m3_train_step.jsonl" 2> "$OUT/ktm3.log" || exit 1
python3 tools/step_kernels.py "$OUT/ktm3" 8 "$OUT/m3_train_step_kernels.csv" "k_recconv_cpt<4" 3 > "$OUT/m3_train_step_kernels.txt"
rm -rf "$OUT/ktm3"
python3 tools/bench_train.py --which hip --batch 128 --steps 8 >> "$OUT/m3_train_step.jsonl" 2>> "$OUT/ktm3.log"
tail -n 5 "$OUT/blocks_fwd_bwd.jsonl" "$OUT/bwd56_kernels.txt" "$OUT/m3_train_step.jsonl"
