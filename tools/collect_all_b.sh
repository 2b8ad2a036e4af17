set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05_a3 --model recnext_a3 > gpurun_out/collect_r05_a3.log 2>&1
bash tools/collect_profiles.sh r05_512 --resolution 512 --batch 32 > gpurun_out/collect_r05_512.log 2>&1
echo done2
