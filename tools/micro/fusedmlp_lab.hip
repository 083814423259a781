// Stand-alone timing lab for csrc/fusedmlp.hip (no torch): builds the kernel with the F2G_MLPVAR /
// F2G_MLP_RING switches and times it on random data at the three block shapes of mel_24k_base, B = 64.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DF2G_MLPVAR=0 tools/micro/fusedmlp_lab.hip -o tools/micro/fusedmlp_lab_0
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../flow2gan_amd/csrc/fusedmlp.hip"
int f2g_check_launch() { return hipGetLastError() == hipSuccess ? 0 : -2; }
void f2g_set_error(const char* m) { fprintf(stderr, "%s\n", m); }

int main() {
  const int shapes[4][2] = {{6016, 768}, {12032, 512}, {24064, 384}, {6016, 512}};
  printf("var %d ring %d:", F2G_MLPVAR, F2G_MLP_RING);
  for (auto& sh : shapes) {
    const int rows = sh[0], C = sh[1], H = 3 * C;
    std::vector<unsigned short> hz((size_t)rows * C), hw((size_t)2 * C * H);
    for (auto& v : hz) v = 0x3c00 + (rand() & 0x3ff);         // bf16 in [0.0078, 0.0156): random mantissas
    for (auto& v : hw) v = (rand() & 1 ? 0x8000 : 0) | (0x3c00 + (rand() & 0x3ff));
    std::vector<float> hx((size_t)rows * C, 0.5f), hv(H, 0.25f), hc(C, 1.f);
    void *z, *wp; float *x, *out, *b1, *al, *b2, *gm;
    hipMalloc(&z, hz.size() * 2); hipMalloc(&wp, hw.size() * 2);
    hipMalloc(&x, hx.size() * 4); hipMalloc(&out, hx.size() * 4);
    hipMalloc(&b1, H * 4); hipMalloc(&al, H * 4); hipMalloc(&b2, C * 4); hipMalloc(&gm, C * 4);
    hipMemcpy(z, hz.data(), hz.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, hv.data(), H * 4, hipMemcpyHostToDevice); hipMemcpy(al, hv.data(), H * 4, hipMemcpyHostToDevice);
    hipMemcpy(b2, hc.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(gm, hc.data(), C * 4, hipMemcpyHostToDevice);
    f2g_fused_mlp_desc d{};
    d.z = z; d.ldz = C; d.wp = wp; d.b1 = b1; d.alpha = al; d.b2 = b2; d.res = x; d.ldres = C; d.gamma = gm;
    d.out = out; d.ldo = C; d.rows = rows; d.C = C; d.H = H; d.parts = 1;
    for (int i = 0; i < 3; ++i) f2g_fused_mlp(&d, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    const int n = 20;
    for (int i = 0; i < n; ++i) f2g_fused_mlp(&d, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %dx%d %.1f us (%.0f TF)", rows, C, ms / n * 1e3, 4.0 * rows * C * H / (ms / n * 1e-3) / 1e12);
    hipFree(z); hipFree(wp); hipFree(x); hipFree(out); hipFree(b1); hipFree(al); hipFree(b2); hipFree(gm);
  }
  printf("\n");
  return 0;
}
