set -u
mkdir -p gpurun_out/h27
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout -k 10 400 python bench.py --steps 30 --warmup 10 > gpurun_out/h27/bench.json 2> gpurun_out/h27/bench.err; cut -c1-400 gpurun_out/h27/bench.json
python -c "
import json; d=json.loads(open('gpurun_out/h27/bench.json').read()); print(d['roofline']); print(d['token_mixers']['ms_per_step'])"
timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | grep cfg4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | cut -c1-200
