#!/bin/bash
# PMC passes over the fused block kernels (tools/fused_multi_bench.py); run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_multi
rm -rf $O; mkdir -p $O
python3 $R/tools/fused_multi_bench.py > $O/times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for MODE in multi serial; do
export MODE
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${MODE}_p$i -o p -- python3 $R/tools/fused_multi_bench.py > $O/${MODE}_log$i.txt 2>&1
  f=$(find $O/${MODE}_p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f fused_ > $O/${MODE}_sum$i.txt 2>&1
done
done
cat $O/times.txt $O/*_sum*.txt
rm -rf $O/*_p1 $O/*_p2
