#!/bin/bash
# time the generator's GEMM shapes with the product library, with tools/micro/libold.so (the build before the
# change under test) and with every lab build present (tools/micro/libx6lab*.so); then the GEMM unit tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export F2G_GEMM=bf16x6
echo "== product"; python tools/gemm_bench.py 2>&1 | grep -E "^R=|precision" | head -7
for f in tools/micro/libold.so $(ls tools/micro/libx6lab*.so 2>/dev/null | sort -V); do
  echo "== $f"; F2G_LIB_PATH=$R/$f python tools/gemm_bench.py 2>&1 | grep "^R=" | head -6
done
