#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
  __syncthreads();
  // lane l: 16-lane group g = l >> 4, i = l & 15: chunk of row (i / 4), column quad (i % 4), block g
  const int l = threadIdx.x;
  const int g = l >> 4, i = l & 15;
  short* p = sm + g * 1024 + (i / 4) * 64 + (i % 4) * 4;   // rows 64 shorts apart
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l*4+j]); printf("\n"); }
  return 0;
}
