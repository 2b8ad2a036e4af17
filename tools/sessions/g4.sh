set -u
mkdir -p gpurun_out/g4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 -x -q 2>&1 | tail -30 > gpurun_out/g4/tests.log
tail -6 gpurun_out/g4/tests.log
timeout 300 python tools/bench_blocks.py --sets m3,m1,m5 --dtypes bf16 --iters 50 --json gpurun_out/g4/blocks.json > gpurun_out/g4/blocks.log 2>&1; grep -o '"shape[^}]*' gpurun_out/g4/blocks.log | cut -c1-260
timeout 600 python bench.py --steps 30 --warmup 10 > gpurun_out/g4/bench.json 2> gpurun_out/g4/bench.err; cut -c1-1200 gpurun_out/g4/bench.json
