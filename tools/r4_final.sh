cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests/test_hip_gan.py -x -q -k "full_width" --durations=10 ) 2>&1 | grep -E "s call|passed|failed|real" | head -14
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.json
