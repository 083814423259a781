#!/bin/bash
# lab builds of the whole library with F2G_LABVAR switches -> tools/micro/libvarN.so (loaded through
# F2G_LIB_PATH).  Bits (bf16 lean K loop): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no
# fragment reads -- results are garbage by construction, only the timing means something.
set -e
cd "$(dirname "$0")/../../flow2gan_amd/csrc"
OUT=../../tools/micro
# the same source list as the product library
SRCS=$(sed -n 's/^SRCS *= *//p' Makefile)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DF2G_LABVAR=$v -shared \
     $SRCS -o $OUT/libvar$v.so &
done
wait
ls -la $OUT/*.so
