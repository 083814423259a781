cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu --durations=25 ) > gpurun_out/r4_gputest_3modes.txt 2>&1
tail -5 gpurun_out/r4_gputest_3modes.txt
python tools/dbg/g_grad_split.py bf16x3 > gpurun_out/r4_g_grad_split.txt 2>&1
python bench.py > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err
tail -c 600 gpurun_out/r4_bench0.json
