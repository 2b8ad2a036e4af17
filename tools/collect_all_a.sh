set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
bash tools/collect_profiles.sh r05_m1 --model recnext_m1 > gpurun_out/collect_r05_m1.log 2>&1
bash tools/collect_profiles.sh r05_m5 --model recnext_m5 > gpurun_out/collect_r05_m5.log 2>&1
echo done3
