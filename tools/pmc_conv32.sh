#!/bin/bash
# PMC passes over the direct MRD conv kernels (tools/conv32_b3_bench.py runs fp32 and split-bf16 instances)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_conv32
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/conv32_b3_bench.py > $O/log$i.txt 2>&1
done
ls $O
