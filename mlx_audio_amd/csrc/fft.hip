// mlx_audio.dsp on gfx950: batched STFT / iSTFT for arbitrary n_fft and the fused
// STFT -> |X|^2 -> mel -> log front ends (dsp.py:385-513, 663-738; whisper/audio.py:41-82;
// qwen3_tts.py:64-120).
//
// FFT: mixed-radix (4, 2, 3, 5, then any odd prime) Stockham autosort, complex fp32, staged entirely
// in LDS together with its twiddle table (computed once per workgroup in fp64, so every twiddle is
// correctly rounded).  Two real frames ride one complex transform (even frame -> re, odd frame ->
// im) and are separated with the Hermitian identities, so an n_fft-point real STFT costs half a
// complex FFT.  A workgroup transforms several frame pairs at once so that all 256 lanes have
// butterflies even for n_fft = 400.
#include "common.h"
#include "fft_fast.h"
#include <stdlib.h>

namespace {

struct FftPlan { int N; int nrad; int rad[24]; };

bool make_plan(int N, FftPlan& p) {
  p.N = N; p.nrad = 0;
  int n = N;
  const int pref[4] = {4, 2, 3, 5};
  for (int i = 0; i < 4; ++i)
    while (n % pref[i] == 0) { if (p.nrad >= 24) return false; p.rad[p.nrad++] = pref[i]; n /= pref[i]; }
  for (int q = 7; n > 1; q += 2)
    while (n % q == 0) { if (p.nrad >= 24 || q > 61) return false; p.rad[p.nrad++] = q; n /= q; }
  return true;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// forward (conj=false) or inverse-unscaled (conj=true) FFT of `npairs` length-N complex vectors held
// at src[pair*N + n]; returns the buffer holding the result (src or dst).
__device__ float2* fft_lds(float2* src, float2* dst, const float2* tw, const FftPlan& pl, int npairs, bool conj) {
  const int N = pl.N;
  int Ns = 1;
  for (int s = 0; s < pl.nrad; ++s) {
    const int R = pl.rad[s];
    const int NR = N / R, stride = N / (Ns * R);
    for (int idx = threadIdx.x; idx < npairs * NR; idx += blockDim.x) {
      const int pr = idx / NR, j = idx - pr * NR;
      const int k = j % Ns;
      const float2* in = src + pr * N;
      float2* out = dst + pr * N + (j / Ns) * Ns * R + k;
      if (R == 4) {
        float2 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float2 w = tw[(k * stride * t) % N];
          if (conj) w.y = -w.y;
          v[t] = cmul(in[j + t * NR], w);
        }
        const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 b0 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), b1 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // forward: -i * b1 ; inverse: +i * b1
        const float2 jb = conj ? make_float2(-b1.y, b1.x) : make_float2(b1.y, -b1.x);
        out[0] = make_float2(a0.x + b0.x, a0.y + b0.y);
        out[Ns] = make_float2(a1.x + jb.x, a1.y + jb.y);
        out[2 * Ns] = make_float2(a0.x - b0.x, a0.y - b0.y);
        out[3 * Ns] = make_float2(a1.x - jb.x, a1.y - jb.y);
      } else if (R == 2) {
        float2 w = tw[(k * stride) % N];
        if (conj) w.y = -w.y;
        const float2 v0 = in[j], v1 = cmul(in[j + NR], w);
        out[0] = make_float2(v0.x + v1.x, v0.y + v1.y);
        out[Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
      } else {
        float2 v[61];
        for (int t = 0; t < R; ++t) {
          float2 w = tw[(int)(((long)k * stride * t) % N)];
          if (conj) w.y = -w.y;
          v[t] = cmul(in[j + t * NR], w);
        }
        for (int q = 0; q < R; ++q) {
          float2 acc = make_float2(0.f, 0.f);
          for (int t = 0; t < R; ++t) {
            float2 w = tw[(int)(((long)NR * t * q) % N)];
            if (conj) w.y = -w.y;
            const float2 m = cmul(v[t], w);
            acc.x += m.x; acc.y += m.y;
          }
          out[q * Ns] = acc;
        }
      }
    }
    __syncthreads();
    float2* tmp = src; src = dst; dst = tmp;
    Ns *= R;
  }
  return src;
}

__device__ __forceinline__ void fill_twiddles(float2* tw, int N) {
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    double s, c;
    sincospi(-2.0 * i / (double)N, &s, &c);
    tw[i] = make_float2((float)c, (float)s);
  }
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

struct StftCommon {
  const float* x; int ldx; int L; int n_fft; int hop; const float* window; int pad_mode; int n_frames;
};

// MODE 0: complex spectrum out [B, n_frames, nb, 2];  MODE 1: log-mel out [B, n_frames, n_mels]
template <int MODE>
__global__ __launch_bounds__(256) void stft_kernel(StftCommon c, FftPlan pl, int pairs, float* out, const float* fb, int n_mels,
                                                   int mel_mode, float* gmax, float log_guard) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = c.n_fft, nb = N / 2 + 1;
  float2* tw = (float2*)smem;
  float2* buf0 = tw + N;
  float2* buf1 = buf0 + pairs * N;
  float* pw = (float*)(buf1 + pairs * N);  // MODE 1: [2*pairs][nb]
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * 2 * pairs;
  fill_twiddles(tw, N);
  const float* xb = c.x + (int64_t)b * c.ldx;
  const int off = c.pad_mode ? N / 2 : 0;
  for (int i = threadIdx.x; i < pairs * N; i += blockDim.x) {
    const int pr = i / N, n = i - pr * N;
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = f0 + 2 * pr + h;
      float s = 0.f;
      if (f < c.n_frames) {
        int idx = f * c.hop + n - off;
        bool ok = true;
        if (c.pad_mode == 1) {  // reflect
          if (idx < 0) idx = -idx;
          if (idx >= c.L) idx = 2 * (c.L - 1) - idx;
        } else if (idx < 0 || idx >= c.L) ok = false;  // constant pad / out of range
        if (ok) s = xb[idx] * c.window[n];
      }
      v[h] = s;
    }
    buf0[i] = make_float2(v[0], v[1]);
  }
  __syncthreads();
  float2* Z = fft_lds(buf0, buf1, tw, pl, pairs, false);
  // separate the two real transforms: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / (2i)
  for (int i = threadIdx.x; i < pairs * nb; i += blockDim.x) {
    const int pr = i / nb, k = i - pr * nb;
    const float2 z = Z[pr * N + k], zc = Z[pr * N + ((N - k) % N)];
    float2 xa = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y));
    float2 xb2 = make_float2(0.5f * (z.y + zc.y), 0.5f * (zc.x - z.x));
    if (k == 0 || 2 * k == N) { xa.y = 0.f; xb2.y = 0.f; }
    const int fa = f0 + 2 * pr;
    if (MODE == 0) {
      if (fa < c.n_frames) *(float2*)(out + (((int64_t)b * c.n_frames + fa) * nb + k) * 2) = xa;
      if (fa + 1 < c.n_frames) *(float2*)(out + (((int64_t)b * c.n_frames + fa + 1) * nb + k) * 2) = xb2;
    } else {
      float pa = xa.x * xa.x + xa.y * xa.y, pb = xb2.x * xb2.x + xb2.y * xb2.y;
      if (mel_mode == 1) { pa = sqrtf(pa + 1e-9f); pb = sqrtf(pb + 1e-9f); }
      else if (mel_mode == 3) { pa = sqrtf(pa); pb = sqrtf(pb); }
      pw[(2 * pr) * nb + k] = pa;
      pw[(2 * pr + 1) * nb + k] = pb;
    }
  }
  if (MODE == 1) {
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float lmax = -INFINITY;
    for (int task = wv; task < 2 * pairs * n_mels; task += 4) {
      const int fl = task / n_mels, m = task - fl * n_mels;
      const int f = f0 + fl;
      if (f >= c.n_frames) continue;
      const float* row = fb + (int64_t)m * nb;
      const float* pr = pw + fl * nb;
      float s = 0.f;
      for (int k = lane; k < nb; k += 64) s = fmaf(pr[k], row[k], s);
      s = wave_sum(s);
      if (lane == 0) {
        float y;
        if (mel_mode == 0) y = log10f(fmaxf(s, 1e-10f));
        else if (mel_mode == 1 || mel_mode == 3) y = logf(fmaxf(s, 1e-5f));
        else if (mel_mode == 4) y = logf(s + log_guard);   // NeMo: parakeet/audio.py:78-79
        else y = logf(fmaxf(s, 1e-8f));  // mode 2: Kaldi fbank (dsp.py:994-995)
        out[((int64_t)b * c.n_frames + f) * n_mels + m] = y;
        lmax = fmaxf(lmax, y);
      }
    }
    if (gmax && lane == 0 && lmax > -INFINITY) atomic_max_float(gmax + b, lmax);
  }
}

__global__ void logmel_finish_kernel(float* y, int64_t n, const float* gmax, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  float v = y[(int64_t)b * n + i];
  v = fmaxf(v, gmax[b] - 8.0f);
  y[(int64_t)b * n + i] = (v + 4.0f) / 4.0f;
}

// inverse: spectra of two frames -> one complex inverse FFT -> two windowed real frames in frames_ws
__global__ __launch_bounds__(256) void istft_frames_kernel(const mi355_istft_args a, FftPlan pl, int pairs, int nf_even) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.n_fft, nb = N / 2 + 1;
  float2* tw = (float2*)smem;
  float2* buf0 = tw + N;
  float2* buf1 = buf0 + pairs * N;
  const int b = blockIdx.y, f0 = blockIdx.x * 2 * pairs;
  fill_twiddles(tw, N);
  const float* sb = a.spec + (int64_t)b * a.n_frames * nb * 2;
  for (int i = threadIdx.x; i < pairs * N; i += blockDim.x) {
    const int pr = i / N, n = i - pr * N;
    const int k = (n <= N / 2) ? n : N - n;
    const bool mirror = n > N / 2;
    float2 xa = make_float2(0.f, 0.f), xb = make_float2(0.f, 0.f);
    const int fa = f0 + 2 * pr;
    if (fa < a.n_frames) xa = *(const float2*)(sb + ((int64_t)fa * nb + k) * 2);
    if (fa + 1 < a.n_frames) xb = *(const float2*)(sb + ((int64_t)(fa + 1) * nb + k) * 2);
    if (k == 0 || 2 * k == N) { xa.y = 0.f; xb.y = 0.f; }  // irfft ignores these imaginary parts
    if (mirror) { xa.y = -xa.y; xb.y = -xb.y; }
    buf0[i] = make_float2(xa.x - xb.y, xa.y + xb.x);  // Xa + i*Xb
  }
  __syncthreads();
  float2* z = fft_lds(buf0, buf1, tw, pl, pairs, true);
  const float invn = 1.0f / (float)N;
  for (int i = threadIdx.x; i < pairs * N; i += blockDim.x) {
    const int pr = i / N, n = i - pr * N;
    const float w = a.window[n];
    float va = z[i].x * invn, vb = z[i].y * invn;
    if (a.clamp) { va = fminf(fmaxf(va, -w), w); vb = fminf(fmaxf(vb, -w), w); }
    const int fa = f0 + 2 * pr;
    float* ws = a.frames_ws + ((int64_t)b * nf_even + fa) * N + n;
    if (fa < nf_even) ws[0] = va * w;
    if (fa + 1 < nf_even) ws[N] = vb * w;
  }
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const mi355_istft_args a, int nf_even) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= a.out_len) return;
  const int N = a.n_fft, hop = a.hop;
  const int p = t + a.trim;
  int f_lo = (p - N + hop) / hop; if (p - N + 1 <= 0) f_lo = 0;
  int f_hi = p / hop; if (f_hi > a.n_frames - 1) f_hi = a.n_frames - 1;
  const float* ws = a.frames_ws + (int64_t)b * nf_even * N;
  float acc = 0.f;
  for (int f = f_lo; f <= f_hi; ++f) {
    const int n = p - f * hop;
    if (n >= 0 && n < N) acc += ws[(int64_t)f * N + n];
  }
  const float nv = a.norm[p];
  float o;
  if (a.norm_mode == 0) o = acc / nv;
  else o = (nv > 1e-10f) ? acc / nv : acc;
  a.out[(int64_t)b * a.ld_out + t] = o;
}

int choose_pairs(int N, int extra_per_frame_bytes) {
  int pairs = 8;
  while (pairs > 1 && (size_t)N * 8 + (size_t)pairs * (2 * N * 8 + 2 * extra_per_frame_bytes) > 60 * 1024) pairs >>= 1;
  return pairs;
}

template <typename K>
int set_lds(K kernel, size_t lds, const char* name) {
  if (lds > 64 * 1024) {
    MI355_REQUIRE(lds <= 160 * 1024, "%s: n_fft too large for LDS (%zu B)", name, lds);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MI355_REQUIRE(e == hipSuccess, "%s: cannot reserve %zu B of LDS: %s", name, lds, hipGetErrorString(e));
  }
  return MI355_OK;
}

// Kaldi frame extraction for compute_fbank_kaldi (dsp.py:821-975): frame f = x[f*shift, +win) (snip_edges) or the same over the
// reflected-edge signal (dsp.py:828-838); optional dither noise; DC removal (frame mean); pre-emphasis within the frame (first sample kept);
// window; zero padding to the FFT size.  One workgroup per frame; the frame mean is a block reduction.
// Vocos ISTFTHead front half (codec/models/vocos/vocos.py:126-134): x [B, Fr, 2 nb] (the head Linear's output: log-magnitude | phase)
// -> spec [B, Fr, nb] complex64 = min(exp(m), clip) * (cos p + i sin p).  Elementwise, one thread per bin.
__global__ __launch_bounds__(256) void polar_spec_kernel(const float* x, int64_t x_bstride, int ldx, int Fr, int nb, float clip, float2* spec,
                                                         int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % nb);
  const int64_t fr = i / nb;
  const int b = (int)(fr / Fr);
  const int f = (int)(fr - (int64_t)b * Fr);
  const float* row = x + (int64_t)b * x_bstride + (int64_t)f * ldx;
  const float m = fminf(expf(row[k]), clip);
  const float ph = row[nb + k];
  spec[i] = make_float2(m * cosf(ph), m * sinf(ph));
}

__global__ __launch_bounds__(256) void kaldi_frames_kernel(const mi355_kaldi_frames_args a) {
  __shared__ float red[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  auto sample = [&](int n) -> float {  // n in [0, win): the (dithered) raw sample of this frame
    int idx = f * a.shift + n - a.pad;
    if (idx < 0) idx = -idx;                         // left edge: waveform[1 : pad+1] reversed
    else if (idx >= a.L) idx = 2 * a.L - 1 - idx;    // right edge: waveform[L-1], waveform[L-2], ...
    float v = a.x[idx];
    if (a.noise) v += a.noise[(int64_t)f * a.win + n] * a.dither;
    return v;
  };
  float s = 0.f;
  for (int n = tid; n < a.win; n += 256) s += sample(n);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.win;
  float* out = a.frames + (int64_t)f * a.n_fft;
  for (int n = tid; n < a.n_fft; n += 256) {
    float v = 0.f;
    if (n < a.win) {
      v = sample(n) - mean;
      if (a.preemph != 0.f && n > 0) v -= a.preemph * (sample(n - 1) - mean);
      v *= a.window[n];
    }
    out[n] = v;
  }
}

// ---- the register-resident two-pass kernels (fft_fast.h) for the sizes the in-scope front ends use; MI355_FFT_FAST=0 keeps the LDS Stockham
// kernel for every size (A/B aid).  Persistent grid: as many workgroups as are resident at once (LDS-limited), never more than tiles.
bool fast_enabled() {  // read per call: the A/B test flips it inside one process
  const char* e = getenv("MI355_FFT_FAST");
  return !(e && e[0] == '0');
}
int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
    }
    return cus;
  }();
  return n;
}
template <int N1, int N2, int MODE, int WAVES, bool PREF>
int launch_fast_cfg(const StftCommon& c, int B, float* out, const float* fb, int n_mels, int mel_mode, float* gmax, float guard, hipStream_t st, const char* name) {
  using G = mi355fft::FastGeom<N1, N2>;
  const size_t lds = G::lds_bytes(MODE == 1, WAVES);
  if (int r = set_lds(mi355fft::stft_fast_kernel<N1, N2, MODE, WAVES, PREF>, lds, name)) return r;
  const int tiles_per_item = (c.n_frames + 2 * G::PW - 1) / (2 * G::PW);   // tiles of ONE WAVE
  const int64_t total = (int64_t)tiles_per_item * B;
  MI355_REQUIRE(total < (1ll << 31), "%s: too many tiles", name);
  const int by_lds = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  const int by_regs = ((PREF ? 2 : 3) * 4) / WAVES > 0 ? ((PREF ? 2 : 3) * 4) / WAVES : 1;   // waves per SIMD x 4 SIMDs / waves per workgroup
  const int64_t resident = (int64_t)cu_count() * (by_lds < by_regs ? by_lds : by_regs);
  const int64_t wgs = (total + WAVES - 1) / WAVES;
  const int grid = (int)(wgs < resident ? wgs : resident);
  mi355fft::FastArgs a{c.x, c.ldx, c.L, c.hop, c.window, c.pad_mode, c.n_frames, B, tiles_per_item, (int)total, out, fb, n_mels, mel_mode, gmax, guard};
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((mi355fft::stft_fast_kernel<N1, N2, MODE, WAVES, PREF>), dim3(grid), dim3(WAVES * 64), lds, st, a);
  MI355_LAUNCH_CHECK(name);
  return MI355_OK;
}
template <int N1, int N2, int MODE>
int launch_fast(const StftCommon& c, int B, float* out, const float* fb, int n_mels, int mel_mode, float* gmax, float guard, hipStream_t st, const char* name) {
  const char* e = getenv("MI355_FFT_PREFETCH");   // A/B: 1 = four waves per workgroup with the next tile's samples prefetched into registers
  if (e && e[0] == '1') return launch_fast_cfg<N1, N2, MODE, 4, true>(c, B, out, fb, n_mels, mel_mode, gmax, guard, st, name);
  return launch_fast_cfg<N1, N2, MODE, 6, false>(c, B, out, fb, n_mels, mel_mode, gmax, guard, st, name);
}
// returns -1 when the size has no fast instantiation
template <int MODE>
int try_fast(const StftCommon& c, int B, float* out, const float* fb, int n_mels, int mel_mode, float* gmax, float guard, hipStream_t st, const char* name) {
  if (!fast_enabled() || (MODE == 1 && n_mels > mi355fft::kMaxMels)) return -1;
  switch (c.n_fft) {
    case 400: return (MODE == 1 && !mi355fft::FastGeom<20, 20>::mel_fits(n_mels)) ? -1 : launch_fast<20, 20, MODE>(c, B, out, fb, n_mels, mel_mode, gmax, guard, st, name);
    case 512: return (MODE == 1 && !mi355fft::FastGeom<16, 32>::mel_fits(n_mels)) ? -1 : launch_fast<16, 32, MODE>(c, B, out, fb, n_mels, mel_mode, gmax, guard, st, name);
    case 1024: return (MODE == 1 && !mi355fft::FastGeom<32, 32>::mel_fits(n_mels)) ? -1 : launch_fast<32, 32, MODE>(c, B, out, fb, n_mels, mel_mode, gmax, guard, st, name);
    default: return -1;
  }
}

}  // namespace

extern "C" int mi355_stft(const mi355_stft_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->window && ap->out, "stft: null tensor");
  const mi355_stft_args a = *ap;
  MI355_REQUIRE(a.n_fft >= 2 && a.hop > 0 && a.n_frames > 0 && a.B > 0, "stft: bad shape");
  MI355_REQUIRE(a.pad_mode >= 0 && a.pad_mode <= 2, "stft: pad_mode must be 0 (none), 1 (reflect) or 2 (constant)");
  MI355_REQUIRE(a.pad_mode != 1 || a.L > a.n_fft / 2, "stft: input too short for reflect padding");
  {
    StftCommon c{a.x, a.ldx, a.L, a.n_fft, a.hop, a.window, a.pad_mode, a.n_frames};
    const int r = try_fast<0>(c, a.B, a.out, nullptr, 0, 0, nullptr, 0.f, (hipStream_t)stream, "stft");
    if (r >= 0) return r;
  }
  FftPlan pl;
  MI355_REQUIRE(make_plan(a.n_fft, pl), "stft: n_fft=%d has a prime factor > 61", a.n_fft);
  const int pairs = choose_pairs(a.n_fft, 0);
  const size_t lds = (size_t)a.n_fft * 8 * (1 + 2 * pairs);
  if (int r = set_lds(stft_kernel<0>, lds, "stft")) return r;
  StftCommon c{a.x, a.ldx, a.L, a.n_fft, a.hop, a.window, a.pad_mode, a.n_frames};
  const int blocks = (a.n_frames + 2 * pairs - 1) / (2 * pairs);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(stft_kernel<0>, dim3(blocks, a.B), dim3(256), lds, (hipStream_t)stream, c, pl, pairs, a.out,
                     (const float*)nullptr, 0, 0, (float*)nullptr, 0.f);
  MI355_LAUNCH_CHECK("stft");
  return MI355_OK;
}

extern "C" int mi355_logmel(const mi355_logmel_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->window && ap->fb && ap->out, "logmel: null tensor");
  const mi355_logmel_args a = *ap;
  MI355_REQUIRE(a.n_fft >= 2 && a.hop > 0 && a.n_frames > 0 && a.B > 0 && a.n_mels > 0, "logmel: bad shape");
  MI355_REQUIRE(a.mode >= 0 && a.mode <= 4, "logmel: mode must be 0 (whisper), 1 (qwen3), 2 (kaldi fbank), 3 (vocos) or 4 (nemo)");
  MI355_REQUIRE(a.mode != 4 || a.log_guard > 0.f, "logmel: mode 4 needs a positive log_guard");
  MI355_REQUIRE(a.pad_mode >= 0 && a.pad_mode <= 2, "logmel: bad pad_mode");
  MI355_REQUIRE(a.pad_mode != 1 || a.L > a.n_fft / 2, "logmel: input too short for reflect padding");
  hipStream_t st = (hipStream_t)stream;
  if (a.gmax) {
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)a.gmax, (int)0xff800000u, a.B, st);
    MI355_REQUIRE(e == hipSuccess, "logmel: memset failed: %s", hipGetErrorString(e));
  }
  {
    StftCommon c{a.x, a.ldx, a.L, a.n_fft, a.hop, a.window, a.pad_mode, a.n_frames};
    const int r = try_fast<1>(c, a.B, a.out, a.fb, a.n_mels, a.mode, a.gmax, a.log_guard, st, "logmel");
    if (r >= 0) return r;
  }
  FftPlan pl;
  MI355_REQUIRE(make_plan(a.n_fft, pl), "logmel: n_fft=%d has a prime factor > 61", a.n_fft);
  const int nb = a.n_fft / 2 + 1;
  const int pairs = choose_pairs(a.n_fft, nb * 4);
  const size_t lds = (size_t)a.n_fft * 8 * (1 + 2 * pairs) + (size_t)2 * pairs * nb * 4;
  if (int r = set_lds(stft_kernel<1>, lds, "logmel")) return r;
  StftCommon c{a.x, a.ldx, a.L, a.n_fft, a.hop, a.window, a.pad_mode, a.n_frames};
  const int blocks = (a.n_frames + 2 * pairs - 1) / (2 * pairs);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(stft_kernel<1>, dim3(blocks, a.B), dim3(256), lds, st, c, pl, pairs, a.out, a.fb, a.n_mels, a.mode, a.gmax, a.log_guard);
  MI355_LAUNCH_CHECK("logmel");
  return MI355_OK;
}

extern "C" int mi355_logmel_finish(float* y, int64_t n_per_item, const float* gmax, int32_t B, void* stream) {
  MI355_REQUIRE(y && gmax && n_per_item > 0 && B > 0, "logmel_finish: bad arguments");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(logmel_finish_kernel, dim3((unsigned)((n_per_item + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, y,
                     n_per_item, gmax, B);
  MI355_LAUNCH_CHECK("logmel_finish");
  return MI355_OK;
}

extern "C" int mi355_istft(const mi355_istft_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->spec && ap->window && ap->norm && ap->frames_ws && ap->out, "istft: null tensor");
  const mi355_istft_args a = *ap;
  MI355_REQUIRE(a.n_fft >= 2 && a.n_fft % 2 == 0 && a.hop > 0 && a.n_frames > 0 && a.B > 0 && a.out_len > 0, "istft: bad shape");
  MI355_REQUIRE(a.trim >= 0 && a.trim + a.out_len <= (a.n_frames - 1) * a.hop + a.n_fft, "istft: trim/out_len outside the overlap-add range");
  FftPlan pl;
  MI355_REQUIRE(make_plan(a.n_fft, pl), "istft: n_fft=%d has a prime factor > 61", a.n_fft);
  const int pairs = choose_pairs(a.n_fft, 0);
  const size_t lds = (size_t)a.n_fft * 8 * (1 + 2 * pairs);
  if (int r = set_lds(istft_frames_kernel, lds, "istft")) return r;
  const int nf_even = (a.n_frames + 1) & ~1;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (a.n_frames + 2 * pairs - 1) / (2 * pairs);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(istft_frames_kernel, dim3(blocks, a.B), dim3(256), lds, st, a, pl, pairs, nf_even);
  MI355_LAUNCH_CHECK("istft_frames");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(istft_ola_kernel, dim3((a.out_len + 255) / 256, a.B), dim3(256), 0, st, a, nf_even);
  MI355_LAUNCH_CHECK("istft_ola");
  return MI355_OK;
}

extern "C" int mi355_kaldi_frames(const mi355_kaldi_frames_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->window && ap->frames, "kaldi_frames: null tensor");
  const mi355_kaldi_frames_args a = *ap;
  MI355_REQUIRE(a.L > 0 && a.win > 0 && a.shift > 0 && a.n_fft >= a.win && a.n_frames > 0 && a.pad >= 0, "kaldi_frames: bad shape");
  MI355_REQUIRE(a.pad < a.L && (int64_t)(a.n_frames - 1) * a.shift + a.win - a.pad <= (int64_t)a.L + a.pad, "kaldi_frames: frames run past the reflected edges");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(kaldi_frames_kernel, dim3(a.n_frames), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("kaldi_frames");
  return MI355_OK;
}

extern "C" int mi355_polar_spec(const float* x, int64_t x_bstride, int32_t ldx, int32_t Fr, int32_t nb, int32_t B, float clip, float* spec,
                                void* stream) {
  MI355_REQUIRE(x && spec, "polar_spec: null tensor");
  MI355_REQUIRE(Fr > 0 && nb > 0 && B > 0 && ldx >= 2 * nb, "polar_spec: bad shape");
  const int64_t total = (int64_t)B * Fr * nb;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(polar_spec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_bstride, ldx, Fr, nb, clip,
                     (float2*)spec, total);
  MI355_LAUNCH_CHECK("polar_spec");
  return MI355_OK;
}
