#!/bin/bash
# One PMC pass (MFMA-busy / wave / wait counters) + kernel trace over a bench.py step with the launch lanes
# off; per-kernel means of: duration, effective clock (GRBM_GUI_ACTIVE / duration), matrix-pipe busy share
# (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)).  usage: pmc_busy.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x6}; shift
O=$R/gpurun_out/pmc_busy_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_STREAMS=0
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode "$@" > $O/log.txt 2>&1
python3 $R/tools/pmc_busy.py $O/p > $R/gpurun_out/pmc_busy_$TAG.txt 2>&1
rm -rf $O/p
cat $R/gpurun_out/pmc_busy_$TAG.txt | head -40
