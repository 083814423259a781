#!/bin/bash
# PMC passes over the laboratory three-piece GEMM (tools/micro/x6_lab, LAB_ONLY = configuration, first shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_x6
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LAB_SHAPES=1
for cfg in ${CFGS:-9 8}; do
export LAB_ONLY=$cfg
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/c${cfg}_p$i -o p -- $R/tools/micro/x6_lab > $O/c${cfg}_log$i.txt 2>&1
  f=$(find $O/c${cfg}_p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f x6_kernel > $O/c${cfg}_sum$i.txt 2>&1
done
done
cat $O/*_sum*.txt
rm -rf $O/c*_p1 $O/c*_p2 $O/c*_p3
