set -u
mkdir -p gpurun_out/h55
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_recconv_gpu.py tests/test_models.py -q -x -k "upadd or a3 or recattn or A3" 2>&1 | tail -3
timeout -k 10 200 python tools/bench_upadd.py 256 > gpurun_out/h55/upadd.jsonl 2>&1
RCX_UPADD_CPL=0 timeout -k 10 200 python tools/bench_upadd.py 256 >> gpurun_out/h55/upadd.jsonl 2>&1
grep '"shape": \[256, 256, 14' gpurun_out/h55/upadd.jsonl
timeout -k 10 300 python tools/bench_configs.py cfg4 2>&1 | tail -1
