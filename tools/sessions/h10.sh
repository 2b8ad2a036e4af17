set -u
mkdir -p gpurun_out/h10
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > gpurun_out/h10/tests.log; tail -4 gpurun_out/h10/tests.log
timeout -k 10 400 python bench.py --steps 30 --warmup 10 > gpurun_out/h10/bench.json 2> gpurun_out/h10/bench.err; cut -c1-900 gpurun_out/h10/bench.json
timeout -k 10 900 bash tools/collect_profiles.sh r02b > gpurun_out/h10/collect.log 2>&1; tail -5 gpurun_out/h10/collect.log
