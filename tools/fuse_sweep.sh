#!/bin/bash
# fast-mode step time with the leaky-ReLU backward fused into the data-gradient epilogues (bit 1: MPD, bit 2: MRD)
for f in 0 1 2 3; do
  F2G_FUSE_LRELU=$f python bench.py --no-cpu-baseline --no-roofline --gemm ${1:-bf16x3} --no-fast-mode 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F2G_FUSE_LRELU=$f', d['ms_per_step'])"
done
