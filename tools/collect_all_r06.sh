#!/bin/bash
# Run ON THE GPU BOX (gpurun): every round-6 record on the sources in the tree -- kernel tables + PMC traffic + bench lines for the five BASELINE
# configurations, the training-path records, the bench line with its cpu_baseline.  Raw output under gpurun_out/; tools/profile_summary.py (called by
# collect_profiles.sh) writes gpurun_out/profiles_<tag>/, which is copied into profiles/ afterwards.
set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
bash tools/collect_profiles.sh r06_512 --resolution 512 --batch 32 > gpurun_out/collect_r06_512.log 2>&1
bash tools/collect_profiles.sh r06_a3 --model recnext_a3 > gpurun_out/collect_r06_a3.log 2>&1
bash tools/collect_profiles.sh r06_m1 --model recnext_m1 > gpurun_out/collect_r06_m1.log 2>&1
bash tools/collect_profiles.sh r06_m5 --model recnext_m5 > gpurun_out/collect_r06_m5.log 2>&1
bash tools/collect_train.sh r06 > gpurun_out/collect_train_r06.log 2>&1
python3 tools/bench_backward.py 256 --bf16-only --hip-only > gpurun_out/train_r06/blocks_fwd_bwd_batch256.jsonl 2>/dev/null
python3 bench.py --steps 30 --warmup 10 > gpurun_out/r06_bench_with_cpu_baseline.json 2> gpurun_out/r06_bench.err
tail -c 600 gpurun_out/r06_bench_with_cpu_baseline.json
ls gpurun_out/profiles_r06*
