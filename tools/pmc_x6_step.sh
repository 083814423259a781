#!/bin/bash
# PMC passes over the stage-2 step in the bf16x6 mode (launch lanes off): MFMA / LDS / L2 counters of the
# library's six-product kernels (gemm_x6_kernel: forward / data gradient over three-piece images;
# gemm_leanw6_kernel: weight gradients with the split inside the kernel); per-kernel means
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_x6s
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_STREAMS=0
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --gemm bf16x6 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode > $O/log$i.txt 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 $R/tools/pmc_summary.py $f gemm_x6_kernel > $O/sum_x6_$i.txt 2>&1
    python3 $R/tools/pmc_summary.py $f gemm_leanw6_kernel > $O/sum_w6_$i.txt 2>&1
  fi
done
cat $O/sum_*.txt
rm -rf $O/p1 $O/p2
