// Harmonic source (SineGen + SourceModuleHnNSF), small-n_fft STFT features and the iSTFT head of the
// iSTFTNet Generator (tts/models/kokoro/istftnet.py:453-709, 797-835; dsp.py:385-513) for gfx950.
//
// Numerics notes (why parts of this file are fp64 / contraction-free):
//  * SineGen's phase reaches 1e5..1e6 rad, so a 1-ulp difference in the x300 interpolation
//    coordinates or in the cumulative sum is a visible phase error.  The coordinate and blend
//    arithmetic therefore reproduces the reference's fp32 op sequence one rounding at a time
//    (no FMA contraction), and the cumulative sum is the same sequential fp32 scan.
//  * the 20-point DFTs are evaluated in fp64 (cost is nil) and rounded once, so the atan2 phase
//    features agree with a correctly-rounded rfft; DC / Nyquist imaginary parts are +0 exactly as
//    pocketfft (numpy, MLX-CPU) produces them.
#include "common.h"

namespace {

// This translation unit is compiled with -ffp-contract=off (see build.py): HIP's default
// -ffp-contract=fast would fuse a*b+c into one rounding and break the op-by-op mirror below.
// (The __fmul_rn/__fadd_rn header intrinsics are plain operators and do NOT prevent contraction.)
#pragma clang fp contract(off)

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

// torch-semantics linear coordinate (tts/models/interpolate.py:94-110), fp32 op by op
__device__ __forceinline__ void lin_coord(int i, float scale, float half_scale, int in_w, int& lo, int& hi, float& fr) {
  float x = mul_rn((float)i, scale);
  x = add_rn(x, half_scale);
  x = sub_rn(x, 0.5f);
  x = fmaxf(x, 0.0f);
  lo = (int)floorf(x);
  hi = min(lo + 1, in_w - 1);
  fr = sub_rn(x, (float)lo);
}

struct SineDims { int L; int small; int big; float sc_dn, hsc_dn, sc_up, hsc_up; };

// tts/models/interpolate.py:41-50: size = max(1, ceil(float(W) * float(scale))) evaluated in doubles;
// 600*F*(1/300) overshoots for some F (e.g. F = 7 -> 15 coarse steps), so this must be exact.
// coarse_f32 (KittenTTS): its SineGen keeps upsample_scale as an mx.array, so the scale factor 1 / upsample_scale reaches interpolate() as a
// float32 (0.0033333334 for 300) and float(W) * float(scale) lands just above 2F: the coarse grid ALWAYS has 2F + 1 points there
// (kitten_tts/istftnet.py:572,595-599), against 2F (mostly) with Kokoro's python double.
__host__ __device__ inline SineDims sine_dims(int len2, int up, int coarse_f32) {
  SineDims d;
  d.L = len2 * up;
  const double s_dn = (double)d.L * (coarse_f32 ? (double)(1.0f / (float)up) : (1.0 / (double)up));
  d.small = (int)ceil(s_dn); if (d.small < 1) d.small = 1;
  d.big = (int)ceil((double)d.small * (double)up); if (d.big < 1) d.big = 1;
  d.sc_dn = (float)((double)d.L / (double)d.small);
  d.hsc_dn = (float)(0.5 * ((double)d.L / (double)d.small));
  d.sc_up = (float)((double)d.small / (double)d.big);
  d.hsc_up = (float)(0.5 * ((double)d.small / (double)d.big));
  return d;
}

// phase_ws[b, h, i] = (cumsum_i(down(rad)) * 2 * pi) * up.  One workgroup per utterance.  The cumulative sum is the reference's sequential fp32
// scan (its rounding feeds a phase that is multiplied by hundreds of radians), but only the additions are sequential: the terms -- the
// down-sampled, harmonic-scaled, wrapped instantaneous frequencies (an fmodf and an interpolation each) -- are computed by all 256 threads into
// LDS a chunk at a time, then one lane per harmonic adds them up out of LDS.
constexpr int kPhaseChunk = 1024;

__global__ __launch_bounds__(256) void sine_phase_kernel(const mi355_sine_source_args a) {
  extern __shared__ __attribute__((aligned(16))) float vterm[];  // [H][kPhaseChunk]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len2 = a.lens2 ? a.lens2[b] : a.L2;
  const SineDims d = sine_dims(len2, a.up, a.coarse_f32);
  const int L = d.L;
  const int small = min(d.small, a.L2 + 1);
  const float* f0 = a.f0 + (int64_t)b * a.ld_f0;
  auto rad = [&](int t, float mult, float ini) {
    const float fn = mul_rn(f0[t / a.up], mult);
    float r = fmodf(fn / a.sr, 1.0f);
    if (r < 0.f) r = add_rn(r, 1.0f);  // python-style modulo (np.mod / mx %) for negative f0
    if (t == 0) r = add_rn(r, ini);
    return r;
  };
  const float pi_f = 3.14159265358979323846f;
  float cum = 0.f;
  for (int c0 = 0; c0 < small; c0 += kPhaseChunk) {
    const int cn = min(kPhaseChunk, small - c0);
    for (int idx = tid; idx < a.H * cn; idx += 256) {
      const int h = idx / cn, i = c0 + (idx - h * cn);
      const float mult = (float)(h + 1);
      const float ini = (h == 0) ? 0.0f : a.rand_ini[(int64_t)b * a.H + h];
      float v;
      if (L == 1) {
        v = rad(0, mult, ini);
      } else {
        int lo, hi; float fr;
        lin_coord(i, d.sc_dn, d.hsc_dn, L, lo, hi, fr);
        v = add_rn(mul_rn(rad(lo, mult, ini), sub_rn(1.0f, fr)), mul_rn(rad(hi, mult, ini), fr));
      }
      vterm[h * kPhaseChunk + (i - c0)] = v;
    }
    __syncthreads();
    if (tid < a.H) {
      float* out = a.phase_ws + ((int64_t)b * a.H + tid) * (a.L2 + 1) + c0;
      const float* vt = vterm + tid * kPhaseChunk;
      for (int i = 0; i < cn; ++i) {
        cum = add_rn(cum, vt[i]);
        out[i] = mul_rn(mul_rn(mul_rn(cum, 2.0f), pi_f), (float)a.up);
      }
    }
    __syncthreads();
  }
}

constexpr int kMergeMaxH = 16;

// EXTREMA = true is the first pass of the KittenTTS variant (l_linear carries activation_quant, kitten_tts/istftnet.py:711-713): only the
// extrema of sine_wavs [L, H] per utterance are produced ({-min, max} in a.quant_ws, integer atomicMax on non-negative floats); the second,
// ordinary pass then quantises each term before the 9-term product.
template <bool EXTREMA>
__global__ __launch_bounds__(256) void sine_merge_kernel(const mi355_sine_source_args a) {
  // the gaussian noise [B, L, H] is the bulk of this kernel's bytes (H floats per sample against one float out): a workgroup's 256 x H block is
  // contiguous, so it comes in as 16-byte loads and is read back per sample at stride H (H = 9: odd, conflict-free)
  __shared__ __attribute__((aligned(16))) float nzs[256 * kMergeMaxH];
  const int t0 = blockIdx.x * 256, tid = threadIdx.x, b = blockIdx.y;
  const int t = t0 + tid;
  const int len2 = a.lens2 ? a.lens2[b] : a.L2;
  const SineDims d = sine_dims(len2, a.up, a.coarse_f32);
  const int L = d.L;
  if (t0 >= L) return;
  {
    const int64_t item = (int64_t)a.L2 * a.up * a.H;            // floats per utterance
    const float* src = a.noise + (int64_t)b * item + (int64_t)t0 * a.H;
    const int nfl = (int)min((int64_t)256 * a.H, item - (int64_t)t0 * a.H);
    if ((((uintptr_t)src) & 15) == 0) {
      for (int i = tid * 4; i < nfl; i += 1024) {
        if (i + 4 <= nfl) *(float4*)(nzs + i) = *(const float4*)(src + i);
        else for (int j = i; j < nfl; ++j) nzs[j] = src[j];
      }
    } else {
      for (int i = tid; i < nfl; i += 256) nzs[i] = src[i];
    }
  }
  __syncthreads();
  float nmn = 0.f, mxv = 0.f;
  const bool live = t < L;
  if (!EXTREMA && !live) return;
  const FakeQuant fq = (!EXTREMA && a.quant_ws) ? FakeQuant(-a.quant_ws[2 * b], a.quant_ws[2 * b + 1]) : FakeQuant(0.f, 0.f);
  if (live) {
  const int small = min(d.small, a.L2 + 1);
  const int big = d.big;
  const float f0v = a.f0[(int64_t)b * a.ld_f0 + t / a.up];
  const float uv = f0v > a.voiced_thr ? 1.0f : 0.0f;
  const float namp = add_rn(mul_rn(uv, a.noise_std), mul_rn(sub_rn(1.0f, uv), a.sine_amp) / 3.0f);
  const float* nz = nzs + tid * a.H;
  int lo = 0, hi = 0; float fr = 0.f;
  if (small > 1) lin_coord(t, d.sc_up, d.hsc_up, small, lo, hi, fr);
  const float omf = sub_rn(1.0f, fr);
  float accv = 0.f;
  for (int h = 0; h < a.H; ++h) {
    float sv = 0.f;
    if (t < big) {
      const float* ph = a.phase_ws + ((int64_t)b * a.H + h) * (a.L2 + 1);
      const float p = small == 1 ? ph[0] : add_rn(mul_rn(ph[lo], omf), mul_rn(ph[hi], fr));
      sv = mul_rn(sinf(p), a.sine_amp);
    }
    float sw = add_rn(mul_rn(sv, uv), mul_rn(namp, nz[h]));
    if (EXTREMA) { nmn = fmaxf(nmn, -sw); mxv = fmaxf(mxv, sw); }
    else if (a.quant_ws) sw = fq(sw);
    accv = add_rn(accv, mul_rn(sw, a.lin_w[h]));
  }
  if (!EXTREMA) a.out[(int64_t)b * a.ld_out + t] = tanhf(add_rn(accv, a.lin_b));
  }
  if (EXTREMA) {
    // 40 000 workgroups x 4 waves hammering the SAME two words per utterance serialised in L2 (the extrema pass took 1.75 ms against 0.2 ms for the
    // merge pass proper): one pair per workgroup (through the noise buffer, which is done with), and only when it RAISES the stored value -- a maximum
    // only grows, a stale read is a smaller value and merely lets the atomic run
    nmn = wave_max(nmn);
    mxv = wave_max(mxv);
    __syncthreads();
    if ((tid & 63) == 0) { nzs[2 * (tid >> 6)] = nmn; nzs[2 * (tid >> 6) + 1] = mxv; }
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < 4; ++i) { nmn = fmaxf(nmn, nzs[2 * i]); mxv = fmaxf(mxv, nzs[2 * i + 1]); }
      volatile float* cur = a.quant_ws + 2 * b;
      if (nmn > cur[0]) atomicMax((int*)a.quant_ws + 2 * b, __float_as_int(nmn));
      if (mxv > cur[1]) atomicMax((int*)a.quant_ws + 2 * b + 1, __float_as_int(mxv));
    }
  }
}

// ---------------------------------------------------------------- small-n_fft STFT -> |X|, angle(X)
constexpr int kMaxSmallFft = 64;

__global__ __launch_bounds__(256) void stft_magphase_kernel(const mi355_stft_magphase_args a) {
  __shared__ double twc[kMaxSmallFft], tws[kMaxSmallFft];
  __shared__ float win[kMaxSmallFft];
  const int N = a.n_fft, nb = N / 2 + 1;
  if (threadIdx.x < N) {
    double s, c;
    sincospi(2.0 * threadIdx.x / (double)N, &s, &c);
    twc[threadIdx.x] = c; tws[threadIdx.x] = s;
    win[threadIdx.x] = a.window[threadIdx.x];
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const int nframes = len / a.hop + 1;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  const float* xb = a.x + (int64_t)b * a.ldx;
  double xw[kMaxSmallFft];
  for (int n = 0; n < N; ++n) {
    int i = f * a.hop + n - N / 2;
    if (i < 0) i = -i;
    if (i >= len) i = 2 * (len - 1) - i;
    xw[n] = (double)mul_rn(xb[i], win[n]);  // frames * w is an fp32 product in the reference
  }
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)f * a.ldy;
  for (int k = 0; k < nb; ++k) {
    double re = 0.0, im = 0.0;
    int m = 0;
    for (int n = 0; n < N; ++n) {
      re += xw[n] * twc[m];
      im -= xw[n] * tws[m];
      m += k; if (m >= N) m -= N;
    }
    float ref = (float)re, imf = (float)im;
    if (k == 0 || 2 * k == N) imf = 0.0f;
    yr[k] = hypotf(ref, imf);
    yr[nb + k] = atan2f(imf, ref);
  }
}

// ---------------------------------------------------------------- iSTFT head
constexpr int kHeadFrames = 48;  // frames per block (plus overlap halo)

__global__ __launch_bounds__(256) void istft_head_kernel(const mi355_istft_head_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.n_fft, nb = N / 2 + 1, hop = a.hop;
  const int halo = (N + hop - 1) / hop;           // frames overlapping one sample
  const int FT = kHeadFrames + halo;               // frames staged per block
  double* twc = (double*)smem;                     // [N]
  double* tws = twc + N;                           // [N]
  float* win = (float*)(tws + N);                  // [N]
  float* spec = win + N;                           // [FT][nb][2]
  float* td = spec + FT * nb * 2;                  // [FT][N] windowed time frames
  const int b = blockIdx.y;
  const int Fr = a.lens ? a.lens[b] : a.Fr;
  const int fa = blockIdx.x * kHeadFrames - halo;  // first staged frame (may be negative)
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    double s, c;
    sincospi(2.0 * i / (double)N, &s, &c);
    twc[i] = c; tws[i] = s; win[i] = a.window[i];
  }
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  for (int i = threadIdx.x; i < FT * nb; i += blockDim.x) {
    const int j = i / nb, k = i - j * nb, f = fa + j;
    float re = 0.f, im = 0.f;
    if (f >= 0 && f < Fr) {
      const float mag = expf(xb[(int64_t)f * a.ldx + k]);
      const float ph = sinf(xb[(int64_t)f * a.ldx + nb + k]);
      re = mul_rn(mag, cosf(ph));
      im = mul_rn(mag, sinf(ph));
    }
    spec[i * 2] = re; spec[i * 2 + 1] = im;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FT * N; i += blockDim.x) {
    const int j = i / N, n = i - j * N;
    const float* sp = spec + j * nb * 2;
    double acc = (double)sp[0];
    if ((N & 1) == 0) acc += ((n & 1) ? -1.0 : 1.0) * (double)sp[(nb - 1) * 2];
    const int kend = (N & 1) ? nb : nb - 1;
    int m = n;  // (k*n) mod N for k = 1
    for (int k = 1; k < kend; ++k) {
      acc += 2.0 * ((double)sp[k * 2] * twc[m] - (double)sp[k * 2 + 1] * tws[m]);
      m += n; if (m >= N) m -= N;
    }
    td[i] = mul_rn((float)(acc / (double)N), win[n]);
  }
  __syncthreads();
  // output samples owned by this block: untrimmed positions p in [blockIdx.x*kHeadFrames*hop, +kHeadFrames*hop)
  const int total = (Fr - 1) * hop;  // trimmed length
  for (int i = threadIdx.x; i < kHeadFrames * hop; i += blockDim.x) {
    const int p = blockIdx.x * kHeadFrames * hop + i;  // untrimmed position
    const int t = p - N / 2;
    if (t < 0 || t >= total) continue;
    int f_lo = (p - N + hop) / hop; if (p - N + 1 <= 0) f_lo = 0;
    int f_hi = p / hop; if (f_hi > Fr - 1) f_hi = Fr - 1;
    float rec = 0.f, ws = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int n = p - f * hop;
      if (n < 0 || n >= N) continue;
      rec = add_rn(rec, td[(f - fa) * N + n]);
      ws = add_rn(ws, mul_rn(win[n], win[n]));
    }
    a.audio[(int64_t)b * a.ld_audio + t] = (ws > 1e-10f) ? rec / ws : rec;
  }
}

// ---------------------------------------------------------------- compile-time (n_fft, hop) variants of the two kernels above
// Kokoro's generator runs n_fft = 20, hop = 5 over 31 681 frames per 6.6 s utterance.  The generic kernels rebuild an fp64 twiddle table with
// sincospi in every workgroup (tens of microseconds of latency on 20 lanes), index it with a runtime (k n) mod N and keep the frame in a
// runtime-indexed array; here the table arrives as a kernel argument (built once on the host), every loop is unrolled so each twiddle is a scalar
// operand, the real-input symmetry x[n] +- x[N-n] halves the products, a workgroup's samples / rows are staged through LDS with coalesced
// accesses and the results leave the same way.  Sums stay in fp64 (rounded once to fp32, like the oracle's rfft / irfft in double).
struct SmallTw { double c[32]; double s[32]; };

template <int N, int HOP>
__global__ __launch_bounds__(256) void stft_magphase_fast_kernel(const mi355_stft_magphase_args a, const SmallTw tw) {
  constexpr int nb = N / 2 + 1, FB = 256, SPAN = (FB - 1) * HOP + N, YS = 2 * nb + 1;
  __shared__ float xs[SPAN];
  __shared__ float ys[FB * YS];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int len = a.lens ? a.lens[b] : a.L;
  const int nframes = len / HOP + 1;
  const int f0 = blockIdx.x * FB;
  if (f0 >= nframes) return;
  const float* xb = a.x + (int64_t)b * a.ldx;
  for (int j = tid; j < SPAN; j += 256) {
    int i = f0 * HOP + j - N / 2;
    if (i < 0) i = -i;
    if (i >= len) i = 2 * (len - 1) - i;
    i = i < 0 ? 0 : (i >= len ? len - 1 : i);   // only reached by frames past the end, which are not stored
    xs[j] = xb[i];
  }
  __syncthreads();
  double xw[N];
#pragma unroll
  for (int n = 0; n < N; ++n) xw[n] = (double)mul_rn(xs[tid * HOP + n], a.window[n]);  // frames * w is an fp32 product in the reference
  double sp[N / 2], sm[N / 2];
#pragma unroll
  for (int n = 1; n < N / 2; ++n) { sp[n] = xw[n] + xw[N - n]; sm[n] = xw[n] - xw[N - n]; }
  float* yrow = ys + tid * YS;
#pragma unroll
  for (int k = 0; k < nb; ++k) {
    double re = (k & 1) ? xw[0] - xw[N / 2] : xw[0] + xw[N / 2];
    double im = 0.0;
#pragma unroll
    for (int n = 1; n < N / 2; ++n) {
      const int m = (k * n) % N;
      re = fma(sp[n], tw.c[m], re);
      im = fma(-sm[n], tw.s[m], im);
    }
    const float ref = (float)re;
    const float imf = (k == 0 || 2 * k == N) ? 0.0f : (float)im;
    yrow[k] = hypotf(ref, imf);
    yrow[nb + k] = atan2f(imf, ref);
  }
  __syncthreads();
  float* yb = a.y + (int64_t)b * a.y_bstride;
  const int nrows = min(FB, nframes - f0);
  for (int i = tid; i < nrows * 2 * nb; i += 256) {
    const int r = i / (2 * nb), c = i - r * (2 * nb);
    yb[(int64_t)(f0 + r) * a.ldy + c] = ys[r * YS + c];
  }
}

template <int N, int HOP>
__global__ __launch_bounds__(256) void istft_head_fast_kernel(const mi355_istft_head_args a, const SmallTw tw) {
  constexpr int nb = N / 2 + 1, HALO = (N + HOP - 1) / HOP, FT = 256, FO = FT - HALO, XS = 2 * nb + 1, TS = N + 1;
  __shared__ float xin[FT * XS];
  __shared__ float td[FT * TS];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int Fr = a.lens ? a.lens[b] : a.Fr;
  const int fa = blockIdx.x * FO - HALO;  // first staged frame (may be negative)
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  for (int i = tid; i < FT * 2 * nb; i += 256) {
    const int j = i / (2 * nb), c = i - j * (2 * nb), f = fa + j;
    xin[j * XS + c] = (f >= 0 && f < Fr) ? xb[(int64_t)f * a.ldx + c] : 0.f;
  }
  __syncthreads();
  {
    const int f = fa + tid;
    const bool valid = f >= 0 && f < Fr;
    const float* xr = xin + tid * XS;
    double re[nb], im[nb];
#pragma unroll
    for (int k = 0; k < nb; ++k) {
      const float mag = expf(xr[k]);
      const float ph = sinf(xr[nb + k]);
      float sn, cs;
      sincosf(ph, &sn, &cs);   // one argument reduction for both (|ph| <= 1); same values as sinf / cosf
      re[k] = valid ? (double)mul_rn(mag, cs) : 0.0;
      im[k] = valid ? (double)mul_rn(mag, sn) : 0.0;
    }
    float* tr = td + tid * TS;
#pragma unroll
    for (int n = 0; n <= N / 2; ++n) {
      const double base = (n & 1) ? re[0] - re[nb - 1] : re[0] + re[nb - 1];
      double A = 0.0, Bs = 0.0;
#pragma unroll
      for (int k = 1; k < nb - 1; ++k) {
        const int m = (k * n) % N;
        A = fma(re[k], tw.c[m], A);
        Bs = fma(im[k], tw.s[m], Bs);
      }
      constexpr double inv_n = 1.0 / (double)N;   // (a multiply instead of 20 fp64 divisions per frame; the fp32 rounding below hides the last bit)
      tr[n] = mul_rn((float)((base + 2.0 * (A - Bs)) * inv_n), a.window[n]);
      if (n > 0 && n < N / 2) tr[N - n] = mul_rn((float)((base + 2.0 * (A + Bs)) * inv_n), a.window[N - n]);
    }
  }
  __syncthreads();
  const int total = (Fr - 1) * HOP;  // trimmed length
  for (int i = tid; i < FO * HOP; i += 256) {
    const int p = blockIdx.x * FO * HOP + i;  // untrimmed position
    const int t = p - N / 2;
    if (t < 0 || t >= total) continue;
    int f_lo = (p - N + HOP) / HOP; if (p - N + 1 <= 0) f_lo = 0;
    int f_hi = p / HOP; if (f_hi > Fr - 1) f_hi = Fr - 1;
    float rec = 0.f, ws = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int n = p - f * HOP;
      if (n < 0 || n >= N) continue;
      rec = add_rn(rec, td[(f - fa) * TS + n]);
      const float w = a.window[n];
      ws = add_rn(ws, mul_rn(w, w));
    }
    a.audio[(int64_t)b * a.ld_audio + t] = (ws > 1e-10f) ? rec / ws : rec;
  }
}

SmallTw make_small_tw(int N) {
  SmallTw tw;
  for (int m = 0; m < 32; ++m) {
    const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)(m % N) / (long double)N;
    tw.c[m] = (double)cosl(ang);
    tw.s[m] = (double)sinl(ang);
  }
  // the exact values at the quarter points (cosl / sinl of a rounded multiple of pi leave ~1e-20 there; the generic kernels' sincospi is exact)
  for (int m = 0; m < N; ++m) {
    if ((4 * m) % N == 0) {
      const int q = (4 * m) / N;
      tw.c[m] = q == 0 ? 1.0 : (q == 2 ? -1.0 : 0.0);
      tw.s[m] = q == 1 ? 1.0 : (q == 3 ? -1.0 : 0.0);
    }
  }
  return tw;
}

// ---------------------------------------------------------------------------------------------- interpolate1d
// tts/models/interpolate.py:61-132: nearest = floor(i * W/size) clipped; linear with torch semantics (half-pixel source coordinate clamped at 0,
// or align_corners).  The coordinate arithmetic is float32 with one rounding per operation, exactly the MLX op sequence (arange * scalar, + scalar,
// - 0.5, maximum): this translation unit is compiled with contraction off, so no step is fused into an FMA (SineGen multiplies these index
// roundings by phase slopes of hundreds of radians).  One thread per output element, rows = N * C.
__global__ __launch_bounds__(256) void interpolate1d_kernel(const mi355_interp1d_args a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.rows * (int64_t)a.size) return;
  const int64_t r = i / a.size;
  const int o = (int)(i - r * a.size);
  const float* xr = a.x + r * a.x_rstride;
  float* yr = a.y + r * a.y_rstride;
  const int W = a.W;
  if (a.mode == 0) {  // nearest
    int idx = a.size == 1 ? 0 : (int)floorf(((float)o * a.scale));
    idx = idx < 0 ? 0 : (idx > W - 1 ? W - 1 : idx);
    yr[o] = xr[idx];
    return;
  }
  if (W == 1) { yr[o] = xr[0]; return; }
  float x;
  if (a.align_corners && a.size > 1) x = ((float)o * a.scale);
  else if (a.size == 1) x = 0.f;
  else {
    x = ((float)o * a.scale);
    if (!a.align_corners) {
      x = x + a.half_scale;
      x = x - 0.5f;
      x = fmaxf(x, 0.f);
    }
  }
  const int lo = (int)floorf(x);
  const int hi = lo + 1 < W - 1 ? lo + 1 : W - 1;
  const float frac = x - (float)lo;
  const float lo_term = xr[lo < W ? lo : W - 1] * (1.0f - frac);
  const float hi_term = xr[hi] * frac;
  yr[o] = lo_term + hi_term;
}

// ---- dynamic uint8 fake quantisation of a module input (tts/models/kitten_tts/quant.py:4-24) ----
// pass 1: y = act(scale * x + shift) (what the consuming conv would have fused) and the tensor's extrema per utterance; min / max are
// order-independent, so the atomics are deterministic.  Both extrema are joined with 0 like the reference, hence {-min, max} >= 0 and
// an integer atomicMax on the float bit patterns orders them.
__device__ __forceinline__ float fq_prologue(const mi355_fake_quant_args& a, float t, int b, int c) {
  if (a.pre_scale) t = t * a.pre_scale[(int64_t)b * a.pre_ld + c] + a.pre_shift[(int64_t)b * a.pre_ld + c];
  if (a.pre_act == MI355_ACT_LEAKY) t = t > 0.f ? t : t * a.pre_slope;
  else if (a.pre_act == MI355_ACT_SNAKE) {
    const float al = a.pre_alpha[c];
    const float sn = sinf(al * t);
    t = t + (1.0f / al) * (sn * sn);
  }
  return t;
}

// VEC: C, ldx, ldy multiples of 4 and 16-byte aligned bases: the workgroup walks (rows, channel quads) with 16-byte accesses
template <bool VEC>
__global__ __launch_bounds__(256) void fq_prepare_kernel(const mi355_fake_quant_args a) {
  const int b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  float* yb = a.y + (int64_t)b * a.y_bstride;
  float nmn = 0.f, mx = 0.f;
  if (VEC) {
    // 256 threads = rps rows x C4 channel quads (rps = 256 / C4 when a row is narrower than the workgroup): one division per thread, none per element
    const int C4 = a.C >> 2, rps = C4 >= 256 ? 1 : 256 / C4, span = C4 >= 256 ? 256 : C4;
    const int ty = C4 >= 256 ? 0 : (int)threadIdx.x / C4, tx = (int)threadIdx.x - ty * span;
    for (int l = blockIdx.x * rps + ty; l < len && ty < rps; l += gridDim.x * rps) {
      for (int c4 = tx; c4 < C4; c4 += span) {
        const float4 v = *(const float4*)(xb + (int64_t)l * a.ldx + 4 * c4);
        float4 t;
        t.x = fq_prologue(a, v.x, b, 4 * c4);
        t.y = fq_prologue(a, v.y, b, 4 * c4 + 1);
        t.z = fq_prologue(a, v.z, b, 4 * c4 + 2);
        t.w = fq_prologue(a, v.w, b, 4 * c4 + 3);
        *(float4*)(yb + (int64_t)l * a.ldy + 4 * c4) = t;
        nmn = fmaxf(fmaxf(nmn, -t.x), fmaxf(fmaxf(-t.y, -t.z), -t.w));
        mx = fmaxf(fmaxf(mx, t.x), fmaxf(fmaxf(t.y, t.z), t.w));
      }
    }
  } else {
    const int64_t n = (int64_t)len * a.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int l = (int)(i / a.C), c = (int)(i - (int64_t)l * a.C);
      const float t = fq_prologue(a, xb[(int64_t)l * a.ldx + c], b, c);
      yb[(int64_t)l * a.ldy + c] = t;
      nmn = fmaxf(nmn, -t);
      mx = fmaxf(mx, t);
    }
  }
  nmn = wave_max(nmn);
  mx = wave_max(mx);
  __shared__ float red[8];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * w] = nmn; red[2 * w + 1] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { nmn = fmaxf(nmn, red[2 * i]); mx = fmaxf(mx, red[2 * i + 1]); }
    atomicMax((int*)a.minmax + 2 * b, __float_as_int(nmn));
    atomicMax((int*)a.minmax + 2 * b + 1, __float_as_int(mx));
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void fq_apply_kernel(const mi355_fake_quant_args a) {
  const int b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  float* yb = a.y + (int64_t)b * a.y_bstride;
  const FakeQuant fq(-a.minmax[2 * b], a.minmax[2 * b + 1]);
  if (VEC) {
    const int C4 = a.C >> 2, rps = C4 >= 256 ? 1 : 256 / C4, span = C4 >= 256 ? 256 : C4;
    const int ty = C4 >= 256 ? 0 : (int)threadIdx.x / C4, tx = (int)threadIdx.x - ty * span;
    for (int l = blockIdx.x * rps + ty; l < len && ty < rps; l += gridDim.x * rps) {
      for (int c4 = tx; c4 < C4; c4 += span) {
        float4* p = (float4*)(yb + (int64_t)l * a.ldy + 4 * c4);
        float4 v = *p;
        v.x = fq(v.x); v.y = fq(v.y); v.z = fq(v.z); v.w = fq(v.w);
        *p = v;
      }
    }
  } else {
    const int64_t n = (int64_t)len * a.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int l = (int)(i / a.C), c = (int)(i - (int64_t)l * a.C);
      float* p = yb + (int64_t)l * a.ldy + c;
      *p = fq(*p);
    }
  }
}

}  // namespace

extern "C" int mi355_fake_quant_u8(const mi355_fake_quant_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y && ap->minmax, "fake_quant_u8: null tensor");
  const mi355_fake_quant_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0 && a.ldx >= a.C && a.ldy >= a.C, "fake_quant_u8: bad shape");
  MI355_REQUIRE(!a.pre_scale == !a.pre_shift, "fake_quant_u8: pre_scale and pre_shift go together");
  MI355_REQUIRE(a.pre_act == MI355_ACT_NONE || a.pre_act == MI355_ACT_LEAKY || a.pre_act == MI355_ACT_SNAKE, "fake_quant_u8: unsupported prologue activation %d", a.pre_act);
  MI355_REQUIRE(a.pre_act != MI355_ACT_SNAKE || a.pre_alpha, "fake_quant_u8: snake needs pre_alpha");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a.minmax, 0, sizeof(float) * 2 * a.B, st);
  MI355_REQUIRE(e == hipSuccess, "fake_quant_u8: memset failed: %s", hipGetErrorString(e));
  const int64_t n = (int64_t)a.L * a.C;
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && (a.x_bstride % 4 == 0) && (a.y_bstride % 4 == 0) &&
                   (((uintptr_t)a.x | (uintptr_t)a.y) & 15) == 0;
  MI355_CLEAR_ERROR();
  if (vec) {
    // a workgroup covers rps = max(1, 256 / (C / 4)) rows per sweep; ~4096 workgroups over the batch, each with several sweeps when the tensor is large
    const int c4 = a.C / 4, rps = c4 >= 256 ? 1 : 256 / c4;
    const unsigned nblk = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)a.L + 8 * rps - 1) / (8 * rps), 4096 / std::max(1, a.B) + 1));
    hipLaunchKernelGGL(fq_prepare_kernel<true>, dim3(nblk, a.B), dim3(256), 0, st, a);
    MI355_LAUNCH_CHECK("fake_quant_u8 (prepare)");
    hipLaunchKernelGGL(fq_apply_kernel<true>, dim3(nblk, a.B), dim3(256), 0, st, a);
    MI355_LAUNCH_CHECK("fake_quant_u8 (apply)");
    return MI355_OK;
  }
  const unsigned nblk = (unsigned)std::min<int64_t>((n + 1023) / 1024, 1024);
  hipLaunchKernelGGL(fq_prepare_kernel<false>, dim3(nblk, a.B), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("fake_quant_u8 (prepare)");
  hipLaunchKernelGGL(fq_apply_kernel<false>, dim3(nblk, a.B), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("fake_quant_u8 (apply)");
  return MI355_OK;
}

extern "C" int mi355_sine_source(const mi355_sine_source_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->f0 && ap->rand_ini && ap->noise && ap->lin_w && ap->phase_ws && ap->out, "sine_source: null tensor");
  const mi355_sine_source_args a = *ap;
  MI355_REQUIRE(a.H > 0 && a.H <= 64 && a.up > 0 && a.L2 > 0 && a.B > 0, "sine_source: bad shape");
  const SineDims d = sine_dims(a.L2, a.up, a.coarse_f32);
  MI355_REQUIRE(d.small <= a.L2 + 1, "sine_source: coarse length %d exceeds workspace", d.small);
  hipStream_t st = (hipStream_t)stream;
  MI355_CLEAR_ERROR();
  MI355_REQUIRE(a.H <= kMergeMaxH, "sine_source: at most %d harmonics", kMergeMaxH);
  hipLaunchKernelGGL(sine_phase_kernel, dim3(a.B), dim3(256), sizeof(float) * a.H * kPhaseChunk, st, a);
  MI355_LAUNCH_CHECK("sine_phase");
  MI355_CLEAR_ERROR();
  if (a.quant_ws) {
    hipError_t e = hipMemsetAsync(a.quant_ws, 0, sizeof(float) * 2 * a.B, st);
    MI355_REQUIRE(e == hipSuccess, "sine_source: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sine_merge_kernel<true>, dim3((d.L + 255) / 256, a.B), dim3(256), 0, st, a);
    MI355_LAUNCH_CHECK("sine_merge (extrema)");
  }
  hipLaunchKernelGGL(sine_merge_kernel<false>, dim3((d.L + 255) / 256, a.B), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("sine_merge");
  return MI355_OK;
}

extern "C" int mi355_stft_magphase(const mi355_stft_magphase_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->window && ap->y, "stft_magphase: null tensor");
  const mi355_stft_magphase_args a = *ap;
  MI355_REQUIRE(a.n_fft >= 2 && a.n_fft <= kMaxSmallFft, "stft_magphase: n_fft must be in [2, %d]", kMaxSmallFft);
  MI355_REQUIRE(a.hop > 0 && a.L > a.n_fft / 2, "stft_magphase: input too short for reflect padding");
  const int nframes = a.L / a.hop + 1;
  MI355_CLEAR_ERROR();
  if (a.n_fft == 20 && a.hop == 5) {
    static const SmallTw tw20 = make_small_tw(20);
    hipLaunchKernelGGL((stft_magphase_fast_kernel<20, 5>), dim3((nframes + 255) / 256, a.B), dim3(256), 0, (hipStream_t)stream, a, tw20);
    MI355_LAUNCH_CHECK("stft_magphase");
    return MI355_OK;
  }
  hipLaunchKernelGGL(stft_magphase_kernel, dim3((nframes + 255) / 256, a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("stft_magphase");
  return MI355_OK;
}

extern "C" int mi355_istft_head(const mi355_istft_head_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->window && ap->audio, "istft_head: null tensor");
  const mi355_istft_head_args a = *ap;
  MI355_REQUIRE(a.n_fft >= 2 && a.n_fft <= kMaxSmallFft && a.hop > 0 && a.Fr > 1, "istft_head: bad shape");
  const int N = a.n_fft, nb = N / 2 + 1;
  if (N == 20 && a.hop == 5) {
    static const SmallTw tw20 = make_small_tw(20);
    constexpr int FO = 256 - (20 + 5 - 1) / 5;
    const int total_untrimmed = (a.Fr - 1) * 5 + 20;
    MI355_CLEAR_ERROR();
    hipLaunchKernelGGL((istft_head_fast_kernel<20, 5>), dim3((total_untrimmed + FO * 5 - 1) / (FO * 5), a.B), dim3(256), 0, (hipStream_t)stream, a, tw20);
    MI355_LAUNCH_CHECK("istft_head");
    return MI355_OK;
  }
  const int halo = (N + a.hop - 1) / a.hop, FT = kHeadFrames + halo;
  const size_t lds = sizeof(double) * 2 * N + sizeof(float) * N + sizeof(float) * FT * nb * 2 + sizeof(float) * FT * N;
  // blocks cover untrimmed positions [0, (Fr-1)*hop + N)
  const int total_untrimmed = (a.Fr - 1) * a.hop + N;
  const int blocks = (total_untrimmed + kHeadFrames * a.hop - 1) / (kHeadFrames * a.hop);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(istft_head_kernel, dim3(blocks, a.B), dim3(256), lds, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("istft_head");
  return MI355_OK;
}

extern "C" int mi355_interpolate1d(const mi355_interp1d_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y, "interpolate1d: null tensor");
  const mi355_interp1d_args a = *ap;
  MI355_REQUIRE(a.rows > 0 && a.W >= 1 && a.size >= 1 && (a.mode == 0 || a.mode == 1), "interpolate1d: bad arguments");
  MI355_CLEAR_ERROR();
  const int64_t n = a.rows * (int64_t)a.size;
  hipLaunchKernelGGL(interpolate1d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("interpolate1d");
  return MI355_OK;
}
