// Batched real FFT of STFT frames through LDS butterflies (reference modules.py:69-78, torch.stft
// with a periodic hann window, center=True), forward and its adjoint (the gradient of
// the frames), for n_fft = 256 ... 4096.
//
// The windowed-DFT GEMM computes an n_fft-point transform with 2*n_fft*(n_fft+2) FLOPs per frame
// (8.4 MFLOP at 2048); the Stockham radix-2 autosort FFT below needs 5*N*log2(N) (0.11 MFLOP) and
// is bound by reading the frame and writing the spectrum.  One block = one frame: the windowed
// samples (or the half spectrum of the gradient) are loaded as complex numbers into LDS, log2(N)
// butterfly stages ping-pong between two LDS arrays, twiddles come from a table computed in double
// precision on the host (one table per size: accuracy ~1e-7 of the frame's largest bin, at least as
// good as the 2048-term fp32 dot products of the GEMM).
//   forward : X[k] = sum_n w[n] x[m*hop + n] e^{-2 pi i k n / N},  k = 0 .. N/2
//             planar rows [Re(0..N/2) | Im(0..N/2)] (fft_to_real, modules.py:31-40) or interleaved
//             [Re0, Im0, Re1, ...] (the MRD's channels-last image, discriminators.py:191-193)
//   adjoint : gframe[n] = w[n] * Re sum_{k=0}^{N/2} (Gr[k] + i Gi[k]) e^{+2 pi i k n / N}
//             (every stored bin is an independent real output: no doubling of interior bins)
#include "common.h"

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <bool ADJ>
__global__ __launch_bounds__(256) void fft_frames_kernel(const f2g_fft_desc d) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int N = d.n_fft, H = N / 2;
  float2* buf0 = sm;
  float2* buf1 = sm + N;
  float2* tw = sm + 2 * N;            // e^{-2 pi i j / N}, j < N/2
  const int tid = threadIdx.x;
  const int row = blockIdx.x;         // frame index = item * F + m
  const int item = row / d.F, m = row - item * d.F;
  for (int j = tid; j < H; j += 256) tw[j] = reinterpret_cast<const float2*>(d.twiddle)[j];
  if (!ADJ) {
    const float* x = d.x + (long long)item * d.x_stride + (long long)m * d.hop;
    for (int n = tid; n < N; n += 256) buf0[n] = make_float2(d.window[n] * x[n], 0.f);
  } else {
    const float* g = d.spec + (long long)row * d.ld_spec;
    for (int k = tid; k < N; k += 256) {
      float2 v = make_float2(0.f, 0.f);
      if (k <= H) v = d.interleaved ? make_float2(g[2 * k], g[2 * k + 1]) : make_float2(g[k], g[H + 1 + k]);
      buf0[k] = v;
    }
  }
  __syncthreads();
  // Stockham radix-2: stage p reads (i, i + N/2), writes (j, j + p) with j = 2*(i - k) + k, k = i % p
  float2* in = buf0;
  float2* out = buf1;
  int tshift = 0;
  for (int t = H; t > 1; t >>= 1) ++tshift;      // log2(H)
  for (int p = 1; p < N; p <<= 1) {
    for (int i = tid; i < H; i += 256) {
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      float2 w = tw[k << tshift];                 // k * (H / p): angle 2 pi k / (2p)
      if (ADJ) w.y = -w.y;
      const float2 u0 = in[i];
      const float2 u1 = cmul(w, in[i + H]);
      out[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      out[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* tmp = in; in = out; out = tmp;
    --tshift;
  }
  if (!ADJ) {
    float* o = d.spec + (long long)row * d.ld_spec;
    for (int k = tid; k <= H; k += 256) {
      const float2 v = in[k];
      if (d.interleaved) { o[2 * k] = v.x; o[2 * k + 1] = v.y; }
      else { o[k] = v.x; o[H + 1 + k] = v.y; }
    }
  } else {
    float* o = d.frames + (long long)row * d.ld_frames;
    for (int n = tid; n < N; n += 256) o[n] = d.window[n] * in[n].x;
  }
}

}  // namespace

extern "C" int f2g_fft_frames(const f2g_fft_desc* d, int32_t adjoint, f2g_stream_t stream) {
  if (!d || !d->window || !d->twiddle || !d->spec) return F2G_EINVAL;
  const int N = d->n_fft;
  if (N < 256 || N > 4096 || (N & (N - 1))) return F2G_EINVAL;
  if (adjoint ? !d->frames : !d->x) return F2G_EINVAL;
  if (d->rows <= 0) return F2G_OK;
  if (d->F <= 0 || d->rows % d->F) return F2G_EINVAL;
  const size_t smem = (size_t)(2 * N + N / 2) * sizeof(float2);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fft_frames_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fft_frames_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    attr_done = true;
  }
  if (adjoint)
    hipLaunchKernelGGL(fft_frames_kernel<true>, dim3(d->rows), dim3(256), smem, (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(fft_frames_kernel<false>, dim3(d->rows), dim3(256), smem, (hipStream_t)stream, *d);
  return f2g_check_launch();
}
