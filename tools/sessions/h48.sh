set -u
mkdir -p gpurun_out/h48
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout -k 10 400 python bench.py --steps 30 --warmup 10 > gpurun_out/h48/bench.json 2> gpurun_out/h48/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/h48/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['token_mixers']['ms_per_step'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke:"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h48/kt -- python3 tools/bench_train.py --model recnext_m3 --batch 128 --steps 6 --which hip > gpurun_out/h48/kt.log 2>&1
f=$(find gpurun_out/h48/kt -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/h48/train_kernel_stats.csv; rm -rf gpurun_out/h48/kt
