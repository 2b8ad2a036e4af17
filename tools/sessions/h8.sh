set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h8
timeout -k 5 60 ./tools/cpt_bench_stamps 56 64 256 1 | cut -c1-220 | tail -2
timeout -k 5 60 ./tools/cpt_bench_stamps 28 128 256 1 | cut -c1-220 | tail -2
timeout -k 5 60 ./tools/cpt_bench_stamps 28 128 128 1 | cut -c1-220 | tail -2
OUT=gpurun_out/h8
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TA_BUSY_sum TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- ./tools/cpt_bench 56 64 256 1 3 > $OUT/g$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT k_recconv | tail -30
