#!/bin/bash
# builds and runs the fused-MLP timing ablations (run ON the GPU box): F2G_MLPVAR bits 1 no weight
# refills, 2 no output epilogue, 4 no z prologue, 8 no p-slab epilogue / barriers; ring depths
cd "$(dirname "$0")/../.."
SPECS=${SPECS:-"0_16 1_16 2_16 4_16 8_16 15_16 0_8 0_24"}
for spec in $SPECS; do
  v=${spec%_*}; r=${spec#*_}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DF2G_MLPVAR=$v -DF2G_MLP_RING=$r \
      tools/micro/fusedmlp_lab.hip -o tools/micro/fusedmlp_lab_$spec 2>/dev/null &
done
wait
for spec in $SPECS; do ./tools/micro/fusedmlp_lab_$spec; done
