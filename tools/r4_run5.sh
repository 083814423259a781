cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -k "conv32" > gpurun_out/r4_conv32_tests.txt 2>&1; tail -2 gpurun_out/r4_conv32_tests.txt
MODE=bf16x6 python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_conv32_probe_x6.txt
grep "all 45\|Win=256\|H= 94 Win= 39\|H=188 Win= 20" gpurun_out/r4_conv32_probe_x6.txt
ONLY=wgrad python tools/conv32_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_conv32_probe_wgrad.txt
grep "all 45\|Win=256\|H= 94 Win= 39\|H=188 Win= 20" gpurun_out/r4_conv32_probe_wgrad.txt
F2G_CONV32_WGRAD_V2=0 ONLY=wgrad python tools/conv32_probe.py 2>&1 | grep "all 45"
