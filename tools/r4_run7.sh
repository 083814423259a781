cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py tests/test_hip_gan.py -x -q -k "conv32 or discriminator_scores or reference_vectors" > gpurun_out/r4_t7.txt 2>&1; tail -2 gpurun_out/r4_t7.txt
F2G_BAND_LANES=1 python -m pytest tests/test_hip_gan.py -x -q -k "reference_vectors or concurrent" > gpurun_out/r4_t7b.txt 2>&1; tail -2 gpurun_out/r4_t7b.txt
ONLY=wgrad python tools/conv32_probe.py 2>&1 | grep "all 45\|Win=256\|H= 94 Win= 39\|H=188 Win= 20"
MODE=bf16x6 ONLY=wgrad python tools/conv32_probe.py 2>&1 | grep "all 45\|Win=256\|H= 94 Win= 39\|H=188 Win= 20"
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline"
for e in "F2G_BAND_LANES=0" "F2G_BAND_LANES=1"; do echo "# $e fp32"; env $e $B 2>/dev/null | tail -1 | cut -c1-160; echo "# $e bf16x6"; env $e $B --gemm bf16x6 2>/dev/null | tail -1 | cut -c1-160; done
