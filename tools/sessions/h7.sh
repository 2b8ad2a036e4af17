set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_recconv_gpu.py -q -x -k "tiled_channel or golden" 2>&1 | tail -3
for a in "56 64 256 1" "28 128 256 1" "56 64 256 0" "28 128 256 0" "56 48 256 1" "28 96 256 1"; do timeout -k 5 60 ./tools/cpt_bench $a; done
timeout -k 5 60 ./tools/cpt_bench_stamps 56 64 256 1 | cut -c1-100
timeout -k 5 60 ./tools/cpt_bench_stamps 28 128 256 1 | cut -c1-100
