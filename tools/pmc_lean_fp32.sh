#!/bin/bash
# PMC passes over the exact-fp32 stage-2 step (launch lanes off): matrix-pipe counters of the kernels the
# headline runs on -- gemm_lean_kernel (every epilogue instance), gemm_leanw_kernel, the generic
# gemm_kernel, the direct MRD convs.  Per-kernel means + the MFMA-busy share of the kernel's life:
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
# (GRBM_GUI_ACTIVE is summed over the 8 XCDs; v_mfma_f32_32x32x2_f32 holds its SIMD's matrix pipe 64 cycles)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lean_fp32
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export F2G_STREAMS=0
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast-mode > $O/log$i.txt 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/sum_$i.txt 2>&1
done
python3 - $O/sum_1.txt <<'PY'
import re, sys
cur, rows = None, {}
for line in open(sys.argv[1]):
    if not line.startswith("   "):
        cur = line.strip(); rows[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        rows[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
print("# MFMA-busy share per kernel (exact-fp32 step, lanes off): kernel, launches in 2 steps, mean us (GRBM_GUI_ACTIVE / 8 / 2.4 GHz), MFMA busy, issue-stall share of wave time")
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0, 0))[0] * kv[1].get("GRBM_GUI_ACTIVE", (0, 0))[1]):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"][1] == 0:
        continue
    n, gui = c["GRBM_GUI_ACTIVE"]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (gui / 8 * 1024)
    stall = c["SQ_WAIT_INST_ANY"][1] / c["SQ_WAVE_CYCLES"][1]
    print(f"{k[:64]:64s} {n:5d} {gui / 8 / 2400:9.1f} us  mfma_busy {busy:5.3f}  wait_inst/wave {stall:5.3f}")
PY
echo; echo "# raw per-kernel means"; cat $O/sum_1.txt $O/sum_2.txt | grep -A9 "gemm_lean\|gemm_kernel\|conv32" | head -300
rm -rf $O/p1 $O/p2
