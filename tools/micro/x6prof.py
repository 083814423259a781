"""In-kernel interval timing of gemm_x6f_kernel's slab loop (lab build F2G_X6LAB=256: tools/micro/x6lab.sh 256;
run with F2G_LIB_PATH=tools/micro/libx6lab256.so F2G_GEMM=bf16x6)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flow2gan_amd import ops, _lib
lib = _lib.lib
lib.f2g_lab_x6prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 8)()
for R, K, N in ((24064, 384, 1152), (24064, 1152, 384), (12032, 512, 1536), (12032, 1536, 512), (6016, 768, 2304), (6016, 2304, 768)):
    A = torch.randn(R, K, device="cuda"); W = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.02)
    o = torch.empty(R, N, device="cuda")
    for _ in range(3): ops.gemm(ops.mat(A), ops.mat(W), o)
    torch.cuda.synchronize(); lib.f2g_lab_x6prof(out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.gemm(ops.mat(A), ops.mat(W), o)
    e.record(); torch.cuda.synchronize()
    lib.f2g_lab_x6prof(out)
    n = max(1, out[3])
    print(f"R={R:6d} K={K:5d} N={N:5d}  {s.elapsed_time(e) * 100:7.1f} us/launch  path {lib.f2g_gemm_last_path()}  per slab iteration (shader clocks): "
          f"chain {out[0] / n:7.0f}  wait at barrier 1 {out[1] / n:6.0f}  fragments + barrier 2 {out[2] / n:6.0f}   (sum {sum(out[:3]) / n:7.0f}; 48 MFMAs = 1536)   per tile: prologue {out[4] / max(1, out[7]):7.0f}  last chain issue {out[5] / max(1, out[7]):6.0f}  epilogue {out[6] / max(1, out[7]):7.0f}  loop {sum(out[:3]) / max(1, out[7]):8.0f}  ({out[7] // 10} tiles)")
