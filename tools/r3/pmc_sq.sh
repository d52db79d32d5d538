#!/bin/bash
# round-3 diagnosis: SQ counters of the MFMA spreader and the two-copy gather at N = 1e7 (separate --pmc passes, no trace domains)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  [ -z "$SKIP_GATHER" ] && rocprofv3 --pmc $set -d $O/g_$tag -o run -- python3 $R/tools/interp_only.py 10000000 3 > $O/g_$tag.log 2>&1 || echo "gather pass $tag failed"
  rocprofv3 --pmc $set -d $O/s_$tag -o run -- python3 $R/tools/r3/spread_mfma_only.py 10000000 3 > $O/s_$tag.log 2>&1 || echo "spread pass $tag failed"
done
for d in $O/g_*/; do python3 $R/tools/pmc_summary.py $d interp_real2_pair; done > $O/gather_summary.txt 2>&1
for d in $O/s_*/; do python3 $R/tools/pmc_summary.py $d spread_mfma; done > $O/spread_summary.txt 2>&1
echo GATHER; cat $O/gather_summary.txt; echo SPREAD; cat $O/spread_summary.txt
# keep only the summaries and CSVs small
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
