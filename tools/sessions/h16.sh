set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h16
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|SQC_INST\|INST_LEVEL\|SQ_WAIT_INST\|SQ_IFETCH" | cut -c1-160 | head -30 > gpurun_out/h16/counters.txt
cat gpurun_out/h16/counters.txt
OUT=gpurun_out/h16
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- ./tools/cpt_bench 56 64 256 1 3 > $OUT/g$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT k_recconv | tail -12
