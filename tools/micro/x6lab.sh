#!/bin/bash
# lab builds of the library with F2G_X6LAB ablations of gemm.hip only (the other objects are the product
# build's) -> tools/micro/libx6lab<N>.so, loaded through F2G_LIB_PATH.  Run `make` in csrc first.
set -e
cd "$(dirname "$0")/../../flow2gan_amd/csrc"
OUT=../../tools/micro
OBJS=$(ls *.o | grep -v '^gemm.o$')
for v in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DF2G_X6LAB=$v -c gemm.hip -o $OUT/x6lab$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libx6lab$v.so $OBJS $OUT/x6lab$v.o && rm -f $OUT/x6lab$v.o ) &
done
wait
ls -la $OUT/libx6lab*.so
