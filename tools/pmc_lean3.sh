#!/bin/bash
# PMC passes over the split-bf16 lean kernel alone (tools/gemm_one.py under F2G_GEMM=bf16x3); run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lean3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_GEMM=bf16x3
SHAPE="${SHAPE:-38016 2560 1024}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/gemm_one.py $SHAPE > $O/log$i.txt 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f gemm_lean > $O/sum$i.txt 2>&1
done
cat $O/sum*.txt
