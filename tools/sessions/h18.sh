set -u
cd $GRAFT_REPO_ROOT
for n in 16 64 128 256; do echo "== N=$n"; timeout -k 5 60 ./tools/cpt_bench_stamps 56 64 $n 1 | cut -c1-96; done
