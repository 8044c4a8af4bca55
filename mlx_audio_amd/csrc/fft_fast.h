// fft_fast.h: the register-resident two-pass STFT of csrc/fft.hip for n_fft = N1 x N2 (400 = 20 x 20: Whisper; 512 = 16 x 32: Kaldi fbank;
// 1024 = 32 x 32: Qwen3 speaker mel, Vocos), fused with |X|^2 -> mel -> log (dsp.py:385-433, 519-609; whisper/audio.py:41-82).
//
// Why a second kernel beside the LDS Stockham one (profiles/r3_kernel_stats_whisper_b64_call30.txt: 8.55 ms for 64 thirty-second windows,
// 0.5 % of the HBM roofline): that kernel is one frame pair per workgroup -- 192 000 workgroups each computing 400 twiddles in fp64, walking a
// four-stage mixed-radix schedule with a barrier per stage and pulling the whole 64 KB filterbank through L2 for 160 dot products.  Here:
//   * a PERSISTENT workgroup (256 lanes, grid = a few per CU) sets up ONCE -- inter-pass twiddles (fp64 sincospi, correctly rounded), the window,
//     and the mel filterbank COMPACTED to its non-zero spans (a triangular filterbank of 80 x 201 is ~480 values: rows resident in LDS) -- and
//     then walks tiles of P frame pairs (two real frames ride one complex transform, P = 256 / max(N1, N2));
//   * the N-point transform is two passes of register-resident small DFTs (N1 points per lane, twiddle, N2 points per lane; compile-time
//     twiddles as instruction literals, csrc/fft_tw.h), through ONE in-place LDS buffer with a padded row pitch (N2 + 1) so that both the
//     column access of pass 1 and the row access of pass 2 are bank-conflict free inside a pair: 3 barriers for the transform instead of 1 per radix;
//   * Hermitian split -> power -> mel as an ascending-k sum over the row's span only (adding the zeros outside changes nothing: same value
//     as the dense dot product in that order) -> log -> staged in LDS -> one contiguous store of the tile's [frames, n_mels] block.
//     The spectrum never exists in HBM; HBM traffic = the samples once (overlapping frames hit in L2) + the log-mel block.
#pragma once
#include "fft_tw.h"

namespace mi355fft {

__host__ __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// v * (c + i s) with the trivial factors folded (c, s are compile-time constants after unrolling)
__host__ __device__ __forceinline__ float2 cmul_const(float2 v, float c, float s) {
  if (c == 1.f && s == 0.f) return v;
  if (c == -1.f && s == 0.f) return make_float2(-v.x, -v.y);
  if (c == 0.f && s == -1.f) return make_float2(v.y, -v.x);   // * -i
  if (c == 0.f && s == 1.f) return make_float2(-v.y, v.x);    // * +i
  return make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
}

template <int N> struct Split;
template <> struct Split<8> { static constexpr int A = 4, B = 2; };
template <> struct Split<16> { static constexpr int A = 4, B = 4; };
template <> struct Split<20> { static constexpr int A = 4, B = 5; };
template <> struct Split<32> { static constexpr int A = 4, B = 8; };

// forward DFT of N points in registers, natural order in and out
template <int N>
__host__ __device__ __forceinline__ void dft(float2 (&v)[N]) {
  if constexpr (N == 2) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
  } else if constexpr (N == 4) {
    const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
    const float2 b0 = cadd(v[1], v[3]), b1 = csub(v[1], v[3]);
    const float2 jb = make_float2(b1.y, -b1.x);  // -i * b1
    v[0] = cadd(a0, b0); v[1] = cadd(a1, jb); v[2] = csub(a0, b0); v[3] = csub(a1, jb);
  } else if constexpr (N == 5) {
    constexpr float c1 = 0x1.3c6ef372fe950p-2f, c2 = -0x1.9e3779b97f4a7p-1f;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0x1.e6f0e134454ffp-1f, s2 = 0x1.2cf2304755a5fp-1f;    // sin(2 pi / 5), sin(4 pi / 5)
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 t1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 t2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 u1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 u2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
    v[1] = make_float2(t1.x + u1.y, t1.y - u1.x);   // t1 - i u1
    v[4] = make_float2(t1.x - u1.y, t1.y + u1.x);
    v[2] = make_float2(t2.x + u2.y, t2.y - u2.x);
    v[3] = make_float2(t2.x - u2.y, t2.y + u2.x);
  } else {
    // n = a B + b, k = k1 + A k2:  X[k1 + A k2] = sum_b W_B^{b k2} * ( W_N^{b k1} * sum_a x[a B + b] W_A^{a k1} )
    constexpr int A = Split<N>::A, B = Split<N>::B;
    float2 y[N];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float2 t[A];
#pragma unroll
      for (int a = 0; a < A; ++a) t[a] = v[a * B + b];
      dft<A>(t);
#pragma unroll
      for (int k1 = 0; k1 < A; ++k1) y[k1 * B + b] = cmul_const(t[k1], Tw<N>::c[(b * k1) % N], Tw<N>::s[(b * k1) % N]);
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
      float2 u[B];
#pragma unroll
      for (int b = 0; b < B; ++b) u[b] = y[k1 * B + b];
      dft<B>(u);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = u[k2];
    }
  }
}

// two configurations of the same kernel (A/B: MI355_FFT_PREFETCH): WAVES = 6 per workgroup without the register prefetch (<= 168 registers: three
// waves per SIMD, 12 per CU for n_fft = 400) or WAVES = 4 with the next tile's samples prefetched into registers (two waves per SIMD, 8 per CU)
constexpr int kFbCap = 1536;      // floats of compacted filterbank resident in LDS (80 x 201 slaney: ~480; 128 x 513: ~1150)
constexpr int kMaxMels = 256;

// One WAVE owns a tile of PW frame pairs from the first sample load to the last store: all the data it exchanges between phases (the transform
// buffer, the power rows, the staged output) sit in its own slice of LDS, so the phases are ordered by the wave's own in-order LDS queue plus a
// wavefront-scope fence -- there is NO workgroup barrier inside the tile loop, and the four waves of a workgroup (and the two workgroups of a CU)
// drift apart freely: one is waiting for its samples while another is in its butterflies.  (First form of this kernel, round-5 call 1: 12 pairs
// per 256-lane workgroup with 7 barriers per tile and 81 KB of LDS = ONE workgroup per CU: 0.87 ms for 64 Whisper windows.)
template <int N1, int N2>
struct FastGeom {
  static constexpr int N = N1 * N2, NB = N / 2 + 1;
  static constexpr int PR = N2 + 1;                                                     // row pitch of the in-place buffer (complex elements)
  static constexpr int BASE = N1 * PR;
  static constexpr int PITCH = N2 < 32 ? BASE + ((N2 % 32) - (BASE % 32) + 32) % 32 : BASE;  // pair pitch: = N2 (mod 32) when a pair is narrower than a half wave
  static constexpr int PW = 64 / (N1 > N2 ? N1 : N2);                                   // frame pairs per wave tile: one task per lane in both passes
  static constexpr int NBP = NB | 1;                                                    // odd pitch of the power rows
  static_assert(PW >= 1 && PW * N1 <= 64 && PW * N2 <= 64, "one task per lane per pass");
  // a wave's slice: the in-place transform buffer; the power rows [2 PW][NBP] and the staged output [2 PW][kMaxMels + 1] are written on top of it
  // once the spectrum has been read into registers (they must fit: checked below for the instantiated sizes)
  static constexpr size_t zbytes = (size_t)PW * PITCH * 8;
  static constexpr size_t pw_floats = (size_t)2 * PW * NBP;
  static constexpr size_t wave_bytes(bool) { return zbytes; }
  // does a filterbank of n_mels rows fit (power rows + staged output inside the slice)?  The launcher falls back to the LDS Stockham kernel otherwise
  static constexpr bool mel_fits(int n_mels) { return pw_floats * 4 + (size_t)2 * PW * (size_t)(n_mels | 1) * 4 <= zbytes; }
  // bytes: twiddles | window | (MODE 1: compact filterbank | spans) | per-wave slices
  static constexpr size_t table_bytes(bool mel) { return (size_t)N * 8 + (size_t)N * 4 + (mel ? (size_t)kFbCap * 4 + (size_t)kMaxMels * 12 + 16 : 0); }
  static constexpr size_t lds_bytes(bool mel, int waves) { return table_bytes(mel) + (size_t)waves * wave_bytes(mel); }
};

struct FastArgs {
  const float* x; int ldx; int L; int hop; const float* window; int pad_mode; int n_frames; int B;
  int tiles_per_item; int total_tiles;
  float* out; const float* fb; int n_mels; int mel_mode; float* gmax; float log_guard;
};

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}
// order this wave's LDS traffic: everything issued before is visible to every lane of the wave afterwards (no s_barrier)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: complex spectrum out [B, n_frames, NB, 2];  MODE 1: log-mel out [B, n_frames, n_mels]
template <int N1, int N2, int MODE, int WAVES, bool PREF>
__global__ __attribute__((amdgpu_flat_work_group_size(WAVES * 64, WAVES * 64), amdgpu_waves_per_eu(PREF ? 2 : 3))) void stft_fast_kernel(const FastArgs c) {
  constexpr int kFastThreads = WAVES * 64, kFastWaves = WAVES;
  using G = FastGeom<N1, N2>;
  constexpr int N = G::N, NB = G::NB, PR = G::PR, PITCH = G::PITCH, PW = G::PW, NBP = G::NBP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* tw = (float2*)smem;                 // W_N^j
  float* win = (float*)(tw + N);
  float* fbc = win + N;                       // MODE 1: compacted filterbank rows | spans | flag
  int* span_lo = (int*)(fbc + kFbCap);
  int* span_len = span_lo + kMaxMels;
  int* span_off = span_len + kMaxMels;
  int* flags = span_off + kMaxMels;           // [0]: rows resident in LDS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  char* wbase = smem + G::table_bytes(MODE == 1) + (size_t)wv * G::wave_bytes(MODE == 1);
  float2* z = (float2*)wbase;                 // [PW][PITCH]
  float* pw = (float*)wbase;                  // [2 PW][NBP]: on top of z, written after the spectrum is in registers

  // ---- once per workgroup
  for (int i = tid; i < N; i += kFastThreads) {
    double s, co;
    sincospi(-2.0 * i / (double)N, &s, &co);
    tw[i] = make_float2((float)co, (float)s);
    win[i] = c.window[i];
  }
  if constexpr (MODE == 1) {
    for (int m = wv; m < c.n_mels; m += kFastWaves) {
      int lo = NB, hi = 0;
      for (int k = lane; k < NB; k += 64)
        if (c.fb[(int64_t)m * NB + k] != 0.f) { lo = min(lo, k); hi = max(hi, k + 1); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
      if (lane == 0) { span_lo[m] = hi > lo ? lo : 0; span_len[m] = hi > lo ? hi - lo : 0; }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int m = 0; m < c.n_mels; ++m) { span_off[m] = tot; tot += span_len[m]; }
      flags[0] = tot <= kFbCap;
    }
    __syncthreads();
    if (flags[0]) {
      for (int m = wv; m < c.n_mels; m += kFastWaves)
        for (int j = lane; j < span_len[m]; j += 64) fbc[span_off[m] + j] = c.fb[(int64_t)m * NB + span_lo[m] + j];
    }
  }
  __syncthreads();
  const bool fb_lds = MODE == 1 ? flags[0] != 0 : false;
  const int off = c.pad_mode ? N / 2 : 0;
  const bool vec4 = (N2 % 4 == 0) && (c.hop % 4 == 0) && (c.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(c.x) & 15) == 0);
  int max_b = -1;            // Whisper's running maximum: kept per wave while it stays inside one item, one atomic per item change
  float max_v = -INFINITY;

  constexpr int Q = N / 4, NL = (PW * Q + 63) / 64;   // float4 pieces of a frame, pieces per lane and tile
  float4 ra[PREF ? NL : 1], rb[PREF ? NL : 1];        // PREF: the tile's samples (even frame | odd frame) in flight / landed
  bool have = false;
  auto load_regs = [&](const float* xrow, const int fr0) {
#pragma unroll
    for (int q = 0; q < (PREF ? NL : 0); ++q) {
      const int i = lane + 64 * q;
      if (i < PW * Q) {
        const int pr = i / Q, n = (i - pr * Q) * 4;
        const int idx = (fr0 + 2 * pr) * c.hop + n - off;
        ra[q] = *(const float4*)(xrow + idx);
        rb[q] = *(const float4*)(xrow + idx + c.hop);
      }
    }
  };
  for (int tile = blockIdx.x * kFastWaves + wv; tile < c.total_tiles; tile += gridDim.x * kFastWaves) {
    const int b = tile / c.tiles_per_item;
    const int f0 = (tile - b * c.tiles_per_item) * 2 * PW;
    const float* xb = c.x + (int64_t)b * c.ldx;
    // ---- windowed frames: even frame -> re, odd frame -> im, element n = N2 n1 + n2 at [n1][n2] of the pair's padded matrix
    const bool inner = f0 * c.hop - off >= 0 && (f0 + 2 * PW - 1) * c.hop - off + N <= c.L && f0 + 2 * PW <= c.n_frames;
    unsigned nzbits = 0;   // OR of the magnitude bits of every sample this lane loads for the tile: zero <=> the whole tile is silence (+0 / -0)
    auto note4 = [&](const float4 v) {
      nzbits |= (__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) << 1;
    };
    if (inner && vec4) {  // every sample of the tile is inside the signal and 16-byte aligned: four samples of both frames per lane and step
      if constexpr (!PREF) {   // straight from memory, two steps in flight
#pragma unroll 2
        for (int i = lane; i < PW * Q; i += 64) {
          const int pr = i / Q, n = (i - pr * Q) * 4;
          const int n1 = n / N2, n2 = n - n1 * N2;
          const int idx = (f0 + 2 * pr) * c.hop + n - off;
          const float4 a = *(const float4*)(xb + idx), bq = *(const float4*)(xb + idx + c.hop), w = *(const float4*)(win + n);
          note4(a); note4(bq);
          float2* dst = z + pr * PITCH + n1 * PR + n2;
          dst[0] = make_float2(a.x * w.x, bq.x * w.x);
          dst[1] = make_float2(a.y * w.y, bq.y * w.y);
          dst[2] = make_float2(a.z * w.z, bq.z * w.z);
          dst[3] = make_float2(a.w * w.w, bq.w * w.w);
        }
      } else {
      if (!have) load_regs(xb, f0);   // first tile of this wave / the one before was an edge tile
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const int i = lane + 64 * q;
        if (i < PW * Q) {
          const int pr = i / Q, n = (i - pr * Q) * 4;
          const int n1 = n / N2, n2 = n - n1 * N2;
          const float4 a = ra[q], bq = rb[q], w = *(const float4*)(win + n);
          note4(a); note4(bq);
          float2* dst = z + pr * PITCH + n1 * PR + n2;
          dst[0] = make_float2(a.x * w.x, bq.x * w.x);
          dst[1] = make_float2(a.y * w.y, bq.y * w.y);
          dst[2] = make_float2(a.z * w.z, bq.z * w.z);
          dst[3] = make_float2(a.w * w.w, bq.w * w.w);
        }
      }
      }
      // the NEXT tile's samples start their trip from HBM now and land while this tile is in its butterflies and mel rows
      have = false;
      const int nt = tile + (int)gridDim.x * kFastWaves;
      if (PREF && nt < c.total_tiles) {
        const int nb_ = nt / c.tiles_per_item, nf0 = (nt - nb_ * c.tiles_per_item) * 2 * PW;
        if (nf0 * c.hop - off >= 0 && (nf0 + 2 * PW - 1) * c.hop - off + N <= c.L && nf0 + 2 * PW <= c.n_frames) {
          load_regs(c.x + (int64_t)nb_ * c.ldx, nf0);
          have = true;
        }
      }
    } else {
      have = false;
#pragma unroll 4
      for (int i = lane; i < PW * N; i += 64) {
        const int pr = i / N, n = i - pr * N;
        const int n1 = n / N2, n2 = n - n1 * N2;
        float v[2];
        if (inner) {
          const int idx = (f0 + 2 * pr) * c.hop + n - off;
          v[0] = xb[idx];
          v[1] = xb[idx + c.hop];
        } else {
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
              if (ok) s = xb[idx];
            }
            v[h] = s;
          }
        }
        const float w = win[n];
        nzbits |= (__float_as_uint(v[0]) | __float_as_uint(v[1])) << 1;
        z[pr * PITCH + n1 * PR + n2] = make_float2(v[0] * w, v[1] * w);
      }
    }
    // ---- a tile of pure silence (Whisper pads every window to 30 s with zeros; a 10 s utterance is two thirds silence): its spectrum is exactly
    // zero, so every mel row is the clamp floor's logarithm -- the same expression the full path would evaluate on a zero sum, hence bit-identical --
    // and the transforms are skipped.  Not taken for mode 1 (sqrt(|X|^2 + 1e-9) is not zero on silence).
    if (__builtin_amdgcn_ballot_w64(nzbits != 0) == 0 && (MODE == 0 || c.mel_mode != 1)) {
      const int nf = min(2 * PW, c.n_frames - f0);
      if constexpr (MODE == 0) {
        float2* dst = (float2*)(c.out + ((int64_t)b * c.n_frames + f0) * NB * 2);
        for (int i = lane; i < nf * NB; i += 64) dst[i] = make_float2(0.f, 0.f);
      } else {
        const float floor_v = c.mel_mode == 0 ? 1e-10f : (c.mel_mode == 3 ? 1e-5f : 1e-8f);
        const float arg = c.mel_mode == 4 ? 0.f + c.log_guard : fmaxf(0.f, floor_v);
        const float y0 = __builtin_amdgcn_logf(arg) * (c.mel_mode == 0 ? 0.30102999566398120f : 0.69314718055994531f);
        if (b != max_b) {
          if (c.gmax && max_b >= 0 && lane == 0 && max_v > -INFINITY) atomic_max_f32(c.gmax + max_b, max_v);
          max_b = b; max_v = -INFINITY;
        }
        if (nf > 0) max_v = fmaxf(max_v, y0);
        float* dst = c.out + ((int64_t)b * c.n_frames + f0) * c.n_mels;
        for (int i = lane; i < nf * c.n_mels; i += 64) dst[i] = y0;
      }
      wave_sync();   // the buffer is rewritten by the next tile's loads
      continue;
    }
    wave_sync();
    // ---- pass 1: lane (pair, n2): N1-point DFT down column n2, inter-pass twiddle W_N^{n2 k1}, back into the same column
    if (lane < PW * N2) {
      const int pr = lane / N2, n2 = lane - pr * N2;
      float2* col = z + pr * PITCH + n2;
      float2 v[N1];
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) v[n1] = col[n1 * PR];
      dft<N1>(v);
      col[0] = v[0];
#pragma unroll
      for (int k1 = 1; k1 < N1; ++k1) {
        const float2 w = tw[n2 * k1];   // (N1 - 1) (N2 - 1) < N: no reduction needed
        col[k1 * PR] = make_float2(v[k1].x * w.x - v[k1].y * w.y, v[k1].x * w.y + v[k1].y * w.x);
      }
    }
    wave_sync();
    // ---- pass 2: lane (pair, k1): N2-point DFT along row k1; X[k1 + N1 k2] goes back in natural order (all reads before any write)
    {
      float2 u[N2];
      const bool act = lane < PW * N1;
      const int pr = act ? lane / N1 : 0, k1 = act ? lane - pr * N1 : 0;
      if (act) {
        const float2* row = z + pr * PITCH + k1 * PR;
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) u[n2] = row[n2];
        dft<N2>(u);
      }
      wave_sync();
      if (act) {
        float2* dst = z + pr * PITCH + k1;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) dst[N1 * k2] = u[k2];
      }
    }
    wave_sync();
    // ---- the two real transforms: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / (2i).  MODE 1: the powers of a lane's bins wait in
    // registers until every lane has read its spectrum values, then go back on top of the transform buffer as the rows the mel stage reads
    constexpr int NS = (PW * NB + 63) / 64;
    float pa_r[MODE == 1 ? NS : 1], pb_r[MODE == 1 ? NS : 1];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int i = lane + 64 * q;
      if (i < PW * NB) {
        const int pr = i / NB, k = i - pr * NB;
        const float2 zk = z[pr * PITCH + k], zc = z[pr * PITCH + (k == 0 ? 0 : N - k)];
        float2 xa = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
        float2 xb2 = make_float2(0.5f * (zk.y + zc.y), 0.5f * (zc.x - zk.x));
        if (k == 0 || 2 * k == N) { xa.y = 0.f; xb2.y = 0.f; }
        const int fa = f0 + 2 * pr;
        if constexpr (MODE == 0) {
          if (fa < c.n_frames) *(float2*)(c.out + (((int64_t)b * c.n_frames + fa) * NB + k) * 2) = xa;
          if (fa + 1 < c.n_frames) *(float2*)(c.out + (((int64_t)b * c.n_frames + fa + 1) * NB + k) * 2) = xb2;
        } else {
          float pa = xa.x * xa.x + xa.y * xa.y, pb = xb2.x * xb2.x + xb2.y * xb2.y;
          if (c.mel_mode == 1) { pa = sqrtf(pa + 1e-9f); pb = sqrtf(pb + 1e-9f); }
          else if (c.mel_mode == 3) { pa = sqrtf(pa); pb = sqrtf(pb); }
          pa_r[q] = pa; pb_r[q] = pb;
        }
      }
    }
    wave_sync();
    if constexpr (MODE == 1) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int i = lane + 64 * q;
        if (i < PW * NB) {
          const int pr = i / NB, k = i - pr * NB;
          pw[(2 * pr) * NBP + k] = pa_r[q];
          pw[(2 * pr + 1) * NBP + k] = pb_r[q];
        }
      }
      wave_sync();
    }
    if constexpr (MODE == 1) {
      // ---- mel rows over their spans, log, staged as [frame][n_mels | 1] on top of the (now dead) transform buffer
      float* stage = pw + G::pw_floats;   // behind the power rows
      const int n_mels = c.n_mels, SP = n_mels | 1;
      if (b != max_b) {
        if (c.gmax && max_b >= 0 && lane == 0 && max_v > -INFINITY) atomic_max_f32(c.gmax + max_b, max_v);
        max_b = b; max_v = -INFINITY;
      }
      float lmax = -INFINITY;
      for (int task = lane; task < 2 * PW * n_mels; task += 64) {
        const int m = task / (2 * PW), fl = task - m * (2 * PW);
        if (f0 + fl >= c.n_frames) continue;
        const int lo = span_lo[m], len = span_len[m];
        const float* prow = pw + fl * NBP + lo;
        float s = 0.f;
        if (fb_lds) {   // two running sums (even / odd k): the LDS round trips of consecutive terms overlap instead of chaining through one fma
          const float* frow = fbc + span_off[m];
          float s1 = 0.f;
          int j = 0;
          for (; j + 1 < len; j += 2) { s = fmaf(prow[j], frow[j], s); s1 = fmaf(prow[j + 1], frow[j + 1], s1); }
          if (j < len) s = fmaf(prow[j], frow[j], s);
          s += s1;
        } else {
          const float* frow = c.fb + (int64_t)m * NB + lo;
          for (int j = 0; j < len; ++j) s = fmaf(prow[j], frow[j], s);
        }
        // log through v_log_f32 (log2, 1 ulp; the argument is clamped to a normal number first) times the constant: the library log10f / logf
        // spend ~25 instructions per value on denormal and special-case handling this argument range cannot reach
        const float floor_v = c.mel_mode == 0 ? 1e-10f : ((c.mel_mode == 1 || c.mel_mode == 3) ? 1e-5f : 1e-8f);   // mode 2: Kaldi fbank (dsp.py:994-995)
        const float arg = c.mel_mode == 4 ? s + c.log_guard : fmaxf(s, floor_v);                                  // mode 4: NeMo, ln(mel + guard)
        const float y = __builtin_amdgcn_logf(arg) * (c.mel_mode == 0 ? 0.30102999566398120f : 0.69314718055994531f);
        stage[fl * SP + m] = y;
        lmax = fmaxf(lmax, y);
      }
      if (c.gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        max_v = fmaxf(max_v, lmax);
      }
      wave_sync();
      const int nf = min(2 * PW, c.n_frames - f0);
      float* dst = c.out + ((int64_t)b * c.n_frames + f0) * n_mels;
      for (int i = lane; i < nf * n_mels; i += 64) {
        const int fl = i / n_mels, m = i - fl * n_mels;
        dst[i] = stage[fl * SP + m];
      }
      wave_sync();
    }
  }
  if constexpr (MODE == 1) {
    if (c.gmax && max_b >= 0 && lane == 0 && max_v > -INFINITY) atomic_max_f32(c.gmax + max_b, max_v);
  }
}

}  // namespace mi355fft
