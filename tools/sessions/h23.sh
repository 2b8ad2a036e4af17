set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for a in "56 64 256 1" "28 128 256 1"; do echo -n "old: "; timeout -k 5 60 ./tools/cpt_bench_old $a; echo -n "new: "; timeout -k 5 60 ./tools/cpt_bench_new $a; done; done
