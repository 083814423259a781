cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( python -m pytest tests -x -q -m gpu ) > gpurun_out/r4_gputest_b.txt 2>&1; tail -3 gpurun_out/r4_gputest_b.txt
MODE=bf16x6 python tools/conv32_probe.py 2>&1 | grep "all 45"
python bench.py --steps 10 --warmup 3 > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_bench1.json').read().strip().split('\n')[-1])
print('fp32 step', d['ms_per_step'], d['value'], 'x3', d['fast_mode']['ms_per_step'], 'x6', d['fast_mode']['fp32_class']['ms_per_step'])
print(json.dumps(d['roofline']['mfma_class']['by_family']))
P
