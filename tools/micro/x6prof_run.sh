#!/bin/bash
# in-kernel interval timing of gemm_x6f_kernel with every profiling lab build present (bit 256 set)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in $(ls tools/micro/libx6lab*.so | sort -V); do
  echo "== $f"; F2G_GEMM=bf16x6 F2G_LIB_PATH=$R/$f python tools/micro/x6prof.py 2>&1 | grep "^R="
done
