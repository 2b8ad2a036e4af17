set -u
mkdir -p gpurun_out/h20
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > gpurun_out/h20/tests.log; tail -3 gpurun_out/h20/tests.log
timeout -k 10 400 python bench.py --steps 30 --warmup 10 > gpurun_out/h20/bench.json 2> gpurun_out/h20/bench.err; cut -c1-600 gpurun_out/h20/bench.json
timeout -k 10 900 bash tools/collect_profiles.sh r02d > gpurun_out/h20/collect.log 2>&1; tail -4 gpurun_out/h20/collect.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
