#!/bin/bash
# lab builds of the GEMM translation unit with F2G_LABVAR switches -> tools/micro/libvarN.so
set -e
cd "$(dirname "$0")/../../flow2gan_amd/csrc"
OUT=../../tools/micro
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DF2G_LABVAR=$v -shared \
     gemm.hip narrow.hip capi.hip -o $OUT/libvar$v.so &
done
wait
ls -la $OUT/*.so
