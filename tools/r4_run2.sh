cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "conv32" 2>&1 | tail -5
for m in fp32 bf16x6; do echo "## MODE=$m"; MODE=$m python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids | grep "all 45\|Win=256\|Win= 20\|Win= 39 \|Win=128 "; done > gpurun_out/r4_conv32_probe_a.txt 2>&1
cat gpurun_out/r4_conv32_probe_a.txt
bash tools/pmc_lean_fp32.sh > gpurun_out/r4_pmc_lean_fp32.txt 2>&1
head -30 gpurun_out/r4_pmc_lean_fp32.txt
