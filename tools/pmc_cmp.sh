#!/bin/bash
# PMC comparison: split-bf16 lean forward kernel vs the K-major weight-gradient kernel (tools/gemm_one.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_cmp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_GEMM=bf16x3
for form in 0 2; do
  if [ $form = 0 ]; then SH="38016 2560 1024 0"; else SH="38016 1024 2560 2"; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM TA_BUSY_avr" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/f${form}_$i -o p -- python3 $R/tools/gemm_one.py $SH > $O/log_${form}_$i.txt 2>&1
  done
done
ls $O
