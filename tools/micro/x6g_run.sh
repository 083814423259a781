#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export F2G_GEMM=bf16x6
for rep in 1 2; do
echo "product        "; python tools/micro/x6g_bench.py 2>&1 | tail -1
echo "x6g=0 (x6f new)"; F2G_OPTS=x6g=0 python tools/micro/x6g_bench.py 2>&1 | tail -1
for f in $(ls tools/micro/libx6lab*.so 2>/dev/null | sort -V); do echo "$f"; F2G_LIB_PATH=$R/$f python tools/micro/x6g_bench.py 2>&1 | tail -1; done
echo "libold (x6f phased)"; F2G_OPTS=x6g=0 F2G_LIB_PATH=$R/tools/micro/libold.so python tools/micro/x6g_bench.py 2>&1 | tail -1
done
