cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LAB_SHAPES=1 LAB_LEGS=0x41
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/labpmc_$tag -o p --output-format csv -- $R/tools/micro/gemm_lab > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/labpmc_$tag/p_counter_collection.csv | grep -v "n=   1 "
done
