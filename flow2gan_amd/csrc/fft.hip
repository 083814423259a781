// Batched real FFT of STFT frames through LDS butterflies (reference modules.py:69-78 torch.stft
// and :106-115 torch.istft, periodic hann window, center=True): the four
// transforms the path needs, for n_fft = 64 ... 4096.
//
// The windowed-DFT GEMM computes an n_fft-point transform with 2*n_fft*(n_fft+2) FLOPs per frame
// (8.4 MFLOP at 2048, 0.53 at 512); the Stockham radix-2 autosort FFT below needs 5*N*log2(N)
// (0.11 / 0.023 MFLOP) and is bound by reading the frame and writing the spectrum.  The samples
// (or the half spectrum) of a frame are loaded as complex numbers into LDS, log2(N) butterfly
// stages ping-pong between two LDS arrays, twiddles come from a table computed in double precision
// on the host (one table per size: accuracy ~1e-7 of the frame's largest bin, at least as good as
// the n_fft-term fp32 dot products of the GEMM).  Frame ownership: n_fft >= 1024 one frame per
// 256-thread block; n_fft <= 512 one frame per WAVE (four frames per block: a 512-point stage
// is 4 butterflies per lane, and the 6016 ... 24064 frames of a generator branch fill the chip).
//
//   forward (0): X[k] = sum_n w[n] x[m*hop + n] e^{-2 pi i k n / N},  k = 0 .. N/2
//             planar rows [Re(0..N/2) | Im(0..N/2)] (fft_to_real, modules.py:31-40) or interleaved
//             [Re0, Im0, Re1, ...] (the MRD's channels-last image, discriminators.py:191-193)
//   adjoint (1): gframe[n] = w[n] * Re sum_{k=0}^{N/2} (Gr[k] + i Gi[k]) e^{+2 pi i k n / N}
//             (every stored bin is an independent real output: no doubling of interior bins)
//   synthesis (2), the iSTFT's windowed inverse transform (A.2; what f2g_istft_ola overlap-adds):
//             f[n] = w[n] (1/N) Re sum_{k=0}^{N/2} c_k (Yr[k] + i Yi[k]) e^{+2 pi i k n / N},
//             c_0 = c_{N/2} = 1, else 2; Yi[0] and Yi[N/2] are IGNORED (as torch.istft does)
//   synthesis adjoint (3), its gradient: Yr'[k] = (c_k/N) sum_n w[n] g[n] cos,
//             Yi'[k] = -(c_k/N) sum_n w[n] g[n] sin, Yi'[0] = Yi'[N/2] = 0
#include "common.h"

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// MODE as above.  WPF: a wave per frame (blockDim = 256 = 4 frames) instead of a block per frame.
template <int MODE, bool WPF>
__global__ __launch_bounds__(256) void fft_frames_kernel(const f2g_fft_desc d) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  constexpr bool INV = MODE == 1 || MODE == 2;     // spectrum -> samples (conjugate twiddles)
  constexpr int NT = WPF ? 64 : 256;                // threads that share a frame
  const int N = d.n_fft, H = N / 2;
  const bool inter = (d.interleaved & 1) != 0, sbf = (d.interleaved & 2) != 0;   // spec layout / element type
  const int tid = WPF ? (threadIdx.x & 63) : threadIdx.x;
  const int sub = WPF ? (threadIdx.x >> 6) : 0;
  float2* tw = sm;                                  // e^{-2 pi i j / N}, j < N/2 (shared by the block)
  float2* buf0 = sm + H + sub * 2 * N;
  float2* buf1 = buf0 + N;
  int row = WPF ? blockIdx.x * 4 + sub : blockIdx.x;   // frame index = item * F + m
  const bool live = row < d.rows;
  row = live ? row : d.rows - 1;
  const int item = row / d.F, m = row - item * d.F;
  for (int j = threadIdx.x; j < H; j += 256) tw[j] = reinterpret_cast<const float2*>(d.twiddle)[j];
  const float inv_n = 1.f / (float)N;
  if (MODE == 0) {
    if (d.reflect_T > 0) {
      // torch.stft's center / reflect padding applied here (no padded copy of the signal)
      const float* x = d.x + (long long)item * d.x_stride;
      const int T = d.reflect_T, q0 = m * d.hop - H;
      for (int n = tid; n < N; n += NT) {
        int q = q0 + n;
        q = q < 0 ? -q : q;
        q = q >= T ? 2 * T - 2 - q : q;
        buf0[n] = make_float2(d.window[n] * x[q], 0.f);
      }
    } else {
      const float* x = d.x + (long long)item * d.x_stride + (long long)m * d.hop;
      for (int n = tid; n < N; n += NT) buf0[n] = make_float2(d.window[n] * x[n], 0.f);
    }
  } else if (MODE == 3) {
    const float* x = d.frames + (long long)row * d.ld_frames;
    for (int n = tid; n < N; n += NT) buf0[n] = make_float2(d.window[n] * x[n], 0.f);
  } else {
    const float* g = d.spec + (long long)row * d.ld_spec;
    const __bf16* gb = reinterpret_cast<const __bf16*>(d.spec) + (long long)row * d.ld_spec;
    for (int k = tid; k < N; k += NT) {
      float2 v = make_float2(0.f, 0.f);
      if (k <= H) {
        if (sbf) v = make_float2((float)gb[k], (float)gb[H + 1 + k]);      // (planar rows only)
        else v = inter ? make_float2(g[2 * k], g[2 * k + 1]) : make_float2(g[k], g[H + 1 + k]);
        if (MODE == 2) {
          const bool edge = k == 0 || k == H;
          const float c = edge ? inv_n : 2.f * inv_n;
          v = make_float2(c * v.x, edge ? 0.f : c * v.y);
        }
      }
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
    for (int i = tid; i < H; i += NT) {
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      float2 w = tw[k << tshift];                 // k * (H / p): angle 2 pi k / (2p)
      if (INV) w.y = -w.y;
      const float2 u0 = in[i];
      const float2 u1 = cmul(w, in[i + H]);
      out[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      out[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* tmp = in; in = out; out = tmp;
    --tshift;
  }
  if (!live) return;
  if (MODE == 0 || MODE == 3) {
    float* o = d.spec + (long long)row * d.ld_spec;
    __bf16* ob = reinterpret_cast<__bf16*>(d.spec) + (long long)row * d.ld_spec;
    for (int k = tid; k <= H; k += NT) {
      float2 v = in[k];
      if (MODE == 3) {
        const bool edge = k == 0 || k == H;
        const float c = edge ? inv_n : 2.f * inv_n;
        v = make_float2(c * v.x, edge ? 0.f : c * v.y);
      }
      if (sbf) { ob[k] = (__bf16)v.x; ob[H + 1 + k] = (__bf16)v.y; }
      else if (inter) { o[2 * k] = v.x; o[2 * k + 1] = v.y; }
      else { o[k] = v.x; o[H + 1 + k] = v.y; }
    }
    // zero padding of the row up to spec_cols (the GEMMs that reduce over a spectrum row read whole
    // 64-column slabs): written here instead of by a separate fill
    for (int c = N + 2 + tid; c < d.spec_cols; c += NT) {
      if (sbf) ob[c] = (__bf16)0.f;
      else o[c] = 0.f;
    }
  } else {
    float* o = d.frames + (long long)row * d.ld_frames;
    for (int n = tid; n < N; n += NT) o[n] = d.window[n] * in[n].x;
  }
}

template <int MODE>
int launch_fft(const f2g_fft_desc& d, hipStream_t st) {
  const int N = d.n_fft;
  const bool wpf = N <= 512;
  const size_t smem = (size_t)((wpf ? 8 : 2) * N + N / 2) * sizeof(float2);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fft_frames_kernel<MODE, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    attr_done = true;
  }
  if (wpf)
    hipLaunchKernelGGL((fft_frames_kernel<MODE, true>), dim3((d.rows + 3) / 4), dim3(256), smem, st, d);
  else
    hipLaunchKernelGGL((fft_frames_kernel<MODE, false>), dim3(d.rows), dim3(256), smem, st, d);
  return f2g_check_launch();
}

}  // namespace

extern "C" int f2g_fft_frames(const f2g_fft_desc* d, int32_t mode, f2g_stream_t stream) {
  if (!d || !d->twiddle || !d->spec || mode < 0 || mode > 3) return F2G_EINVAL;
  const int N = d->n_fft;
  if (N < 64 || N > 4096 || (N & (N - 1))) return F2G_EINVAL;
  if (!d->window || (mode == 0 ? !d->x : !d->frames)) return F2G_EINVAL;
  if (mode == 0 && d->reflect_T != 0 && d->reflect_T <= N / 2) return F2G_EINVAL;
  if ((d->interleaved & 2) && (d->interleaved & 1)) return F2G_EINVAL;    // bf16 spectra are planar
  if (d->spec_cols > d->ld_spec) return F2G_EINVAL;
  if (d->rows <= 0) return F2G_OK;
  if (d->F <= 0 || d->rows % d->F) return F2G_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: return launch_fft<0>(*d, st);
    case 1: return launch_fft<1>(*d, st);
    case 2: return launch_fft<2>(*d, st);
    default: return launch_fft<3>(*d, st);
  }
}
