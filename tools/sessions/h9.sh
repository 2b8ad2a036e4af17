set -u
cd $GRAFT_REPO_ROOT
for a in "56 64 256 1" "28 128 256 1" "56 64 256 0" "28 128 512 1" "56 80 256 1"; do timeout -k 5 60 ./tools/cpt_bench $a; RCX_CPT_GRID=100000 timeout -k 5 60 ./tools/cpt_bench $a;  done
