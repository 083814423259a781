cd $GRAFT_REPO_ROOT
B="python bench.py --gemm bf16x6 --steps 8 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline"
for m in 0 0x4 0xA 0x15 0x1F; do echo "# F2G_X6_HYBRID=$m"; F2G_X6_HYBRID=$m $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
