cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "conv32" > gpurun_out/r4_conv32_tests.txt 2>&1; tail -3 gpurun_out/r4_conv32_tests.txt
{
for v in 0 1 2 4 8 16 7 15; do
  if [ $v = 0 ]; then L=""; else L=$GRAFT_REPO_ROOT/tools/micro/libc6v$v.so; fi
  echo "## F2G_LABVAR=$v"
  F2G_LIB_PATH=$L MODE=bf16x6 ONLY=fwd python tools/conv32_probe.py 2>&1 | grep "all 45\|H= 47 Win=256\|H= 94 Win= 39\|H=188 Win= 20"
done
} > gpurun_out/r4_conv32x6_ablate.txt 2>&1
cat gpurun_out/r4_conv32x6_ablate.txt
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_c6
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  MODE=bf16x6 ONLY=fwd timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/conv32_probe.py > $O/log$i.txt 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f fwd6 > $O/sum_$i.txt 2>&1
done
cat $O/sum_*.txt > $GRAFT_REPO_ROOT/gpurun_out/r4_conv32x6_pmc.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r4_conv32x6_pmc.txt
rm -rf $O
