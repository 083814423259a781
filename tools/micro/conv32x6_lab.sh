#!/bin/bash
# lab builds of the library with F2G_LABVAR ablations of conv32x6.hip only (the other objects are the
# product build's) -> tools/micro/libc6v<N>.so, loaded through F2G_LIB_PATH.  Run `make` in csrc first.
set -e
cd "$(dirname "$0")/../../flow2gan_amd/csrc"
OUT=../../tools/micro
OBJS=$(ls *.o | grep -v '^conv32x6.o$')
for v in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DF2G_LABVAR=$v -c conv32x6.hip -o $OUT/c6v$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libc6v$v.so $OBJS $OUT/c6v$v.o && rm -f $OUT/c6v$v.o ) &
done
wait
ls -la $OUT/libc6v*.so
