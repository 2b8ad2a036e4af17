set -u
cd $GRAFT_REPO_ROOT
for a in "56 64 256 1" "28 128 256 1" "56 64 256 0"; do echo "PF: "; timeout -k 5 60 ./tools/cpt_bench $a; echo "no PF: "; timeout -k 5 60 ./tools/cpt_bench_nopf $a; done
timeout -k 5 60 ./tools/cpt_bench_stamps 56 64 256 1 | cut -c1-96
timeout -k 5 60 ./tools/cpt_bench_stamps 28 128 256 1 | cut -c1-96
