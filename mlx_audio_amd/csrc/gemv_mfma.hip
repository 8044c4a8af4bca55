// Skinny GEMM for 5..8 rows per decode step on the matrix pipe (gfx950): y[m, n] = epilogue(sum_k norm(x)[m, k] * W[n, k]).
//
// Same contract as mi355_gemv (gemv.hip; replaces nn.Linear at sequence length 1 for batches of 5..8 sequences: Whisper TextDecoder
// stt/models/whisper/whisper.py:347-416, 498; Qwen3-TTS talker / code predictor tts/models/qwen3_tts/talker.py:230-330, 503-764).  The FMA
// kernel does 8 fp32 FMAs per weight element at 8 rows and ran the weight stream at ~1 TB/s against ~3.7 TB/s at 1 row: its waves sit in
// VALU / LDS work and latency-bound staging instead of keeping loads in flight (profiles/r1_gemv_launch_periods_call23_27.txt).  Here the products
// go to v_mfma_f32_16x16x32 (bf16 or fp16, the weights' own type):
//   * A operand = a 16-row tile of W, straight from HBM: lane (i = lane & 15, g = lane >> 4) loads the 32 contiguous bytes W[n0 + i][k0 + 16 g ..
//     + 16) of a 64-wide k step -- full 128-byte lines per row, no conversion, no LDS -- and feeds its two 16-byte halves to two MFMAs (the
//     contraction index of an MFMA is a free relabelling as long as A and B agree: half h of group g carries k = k0 + 16 g + 8 h + e);
//   * B operand = the (normalised) input rows, split ONCE per workgroup into hi + lo images of the weights' 16-bit type (about 16 mantissa bits
//     for bf16 -- the same split conv_gemm runs prefill with -- and 22 for fp16) and laid out in LDS in fragment order
//     [k step][half][group][row] x 16 bytes, so every ds_read_b128 of a wave covers 512 contiguous bytes: conflict-free;
//   * the four waves of a workgroup take interleaved k steps of ONE 16-column tile (split-K: N / 16 workgroups x 4 waves keep enough loads in
//     flight even for N = 1024) and add their 16 x 16 partial tiles through LDS; thread (n = t & 15, m = t >> 4) finishes one output: bias,
//     activation, LayerScale, residual, SwiGLU pairs, split destinations (q -> buffer, k | v -> KV-cache slot);
//   * fused LayerNorm / RMSNorm: statistics from the registers the staging loads already hold (one read of x), two-pass numerics.
// Rows m >= M of the 16-wide MFMA column space alias rows m & 7: their outputs are never read.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kKC = 2048;   // input columns staged per chunk: 2 images x 8 rows x 2 bytes x kKC = 64 KB of LDS
constexpr int kD = 8;       // weight prefetch depth in k steps (8 x 32 bytes per lane = 16 KB per wave in flight: a whole K = 2048 slice)

__device__ __forceinline__ float mfma_act(float v, int act, float slope) {
  switch (act) {
    case MI355_ACT_LEAKY: return v > 0.f ? v : v * slope;
    case MI355_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case MI355_ACT_SILU: return v / (1.0f + expf(-v));
    case MI355_ACT_GELU_TANH: return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    case MI355_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case MI355_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// hi + lo images of two fp32 values (low half = first value)
template <bool F16>
__device__ __forceinline__ void split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  if constexpr (F16) {
    hi = pack_f16x2(a, b);
    const float ha = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi & 0xffffu)), hb = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi >> 16));
    lo = pack_f16x2(a - ha, b - hb);
  } else {
    hi = pack_bf16x2(a, b);
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16x2(a - ha, b - hb);
  }
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma_16(const uint4 a, const uint4 b, const f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const mi355_gemv_args a) {
  extern __shared__ __attribute__((aligned(16))) uint4 planes[];  // [2 images][steps of the chunk][2 halves][4 groups][8 rows] 16-byte pieces
  __shared__ float red[4][256];
  __shared__ float st_part[4][8];
  __shared__ float st_rstd[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int K = a.K, M = a.M;
  const int KC = K < kKC ? K : kKC;            // multiple of 64
  const int img = KC * 8 / 8;                  // 16-byte pieces per image: 8 rows x KC / 8
  const int steps_total = K >> 6;
  const int gi = lane >> 4, li = lane & 15;    // MFMA k group / row of W (A) resp. column = input row (B)
  const int nrow = n0 + li < a.N ? n0 + li : a.N - 1;   // tail tile: clamped rows recompute the last row, never stored
  const uint16_t* wrow = a.w + (int64_t)nrow * a.ldw + 16 * gi;

  // ---- weight stream: this wave's k steps are wave, wave + 4, ... ; kD of them are always in flight, issued before x is even staged
  uint4 ring[kD][2];
  auto issue = [&](const int s, uint4 (&dst)[2]) {
    const uint16_t* p = wrow + ((int64_t)s << 6);
    dst[0] = *(const uint4*)p;
    dst[1] = *(const uint4*)(p + 8);
  };
#pragma unroll
  for (int d = 0; d < kD; ++d)
    if (wave + 4 * d < steps_total) issue(wave + 4 * d, ring[d]);

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int it = 0;  // index of this wave's next step: s = wave + 4 * it
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = K - k0 < KC ? K - k0 : KC;  // multiple of 64
    // ---- stage the chunk: thread t owns the 8-column groups q = t, t + 256, ... of EVERY row (norm weight / bias loaded once per column)
    __syncthreads();  // the previous chunk's readers are done
    {
      constexpr int NQ = kKC / 8 / 256;  // 1
      float4 xa[NQ][8], xb[NQ][8], wa[NQ], wb2[NQ], ba[NQ], bb2[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int q = tid + 256 * j, k = q * 8;
        wa[j] = wb2[j] = make_float4(1.f, 1.f, 1.f, 1.f);
        ba[j] = bb2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kc && a.norm && a.norm_weight) { wa[j] = *(const float4*)(a.norm_weight + k0 + k); wb2[j] = *(const float4*)(a.norm_weight + k0 + k + 4); }
        if (k < kc && a.norm && a.norm_bias) { ba[j] = *(const float4*)(a.norm_bias + k0 + k); bb2[j] = *(const float4*)(a.norm_bias + k0 + k + 4); }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          xa[j][m] = xb[j][m] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k < kc && m < M) {
            const float* p = a.x + (int64_t)m * a.ldx + k0 + k;
            xa[j][m] = *(const float4*)p;
            xb[j][m] = *(const float4*)(p + 4);
          }
        }
      }
      if (a.norm) {  // whole rows are in this chunk (host-side eligibility: K <= kKC): two-pass statistics from the registers
        float s[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          s[m] = 0.f;
#pragma unroll
          for (int j = 0; j < NQ; ++j) s[m] += ((xa[j][m].x + xa[j][m].y) + (xa[j][m].z + xa[j][m].w)) + ((xb[j][m].x + xb[j][m].y) + (xb[j][m].z + xb[j][m].w));
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) s[m] = wave_sum(s[m]);
        if (lane < 8) {
          float v = s[0];
#pragma unroll
          for (int m = 1; m < 8; ++m) v = lane == m ? s[m] : v;
          st_part[wave][lane] = v;
        }
        __syncthreads();
        float mean[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) mean[m] = a.norm == 1 ? ((st_part[0][m] + st_part[1][m]) + (st_part[2][m] + st_part[3][m])) / (float)K : 0.f;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          float qv = 0.f;
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            if ((tid + 256 * j) * 8 < kc) {
              const float d0 = xa[j][m].x - mean[m], d1 = xa[j][m].y - mean[m], d2 = xa[j][m].z - mean[m], d3 = xa[j][m].w - mean[m];
              const float d4 = xb[j][m].x - mean[m], d5 = xb[j][m].y - mean[m], d6 = xb[j][m].z - mean[m], d7 = xb[j][m].w - mean[m];
              qv += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
            }
          }
          s[m] = wave_sum(qv);
        }
        if (lane < 8) {
          float v = s[0];
#pragma unroll
          for (int m = 1; m < 8; ++m) v = lane == m ? s[m] : v;
          st_part[wave][lane] = v;
        }
        __syncthreads();
        if (tid < 8) {
          const float var = ((st_part[0][tid] + st_part[1][tid]) + (st_part[2][tid] + st_part[3][tid])) / (float)K;
          st_rstd[tid] = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float mu = mean[m], rs = st_rstd[m];
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            xa[j][m] = make_float4((xa[j][m].x - mu) * rs * wa[j].x + ba[j].x, (xa[j][m].y - mu) * rs * wa[j].y + ba[j].y,
                                   (xa[j][m].z - mu) * rs * wa[j].z + ba[j].z, (xa[j][m].w - mu) * rs * wa[j].w + ba[j].w);
            xb[j][m] = make_float4((xb[j][m].x - mu) * rs * wb2[j].x + bb2[j].x, (xb[j][m].y - mu) * rs * wb2[j].y + bb2[j].y,
                                   (xb[j][m].z - mu) * rs * wb2[j].z + bb2[j].z, (xb[j][m].w - mu) * rs * wb2[j].w + bb2[j].w);
          }
        }
      }
      // hi / lo images in fragment order: piece (step, half, group, row) = ((step * 2 + half) * 4 + group) * 8 + row
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int q = tid + 256 * j;
        if (q * 8 < kc) {
          const int step = q >> 3, r = q & 7, g = r >> 1, h = r & 1;
          const int base = ((step * 2 + h) * 4 + g) * 8;
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            uint4 hi, lo;
            split2<F16>(xa[j][m].x, xa[j][m].y, hi.x, lo.x);
            split2<F16>(xa[j][m].z, xa[j][m].w, hi.y, lo.y);
            split2<F16>(xb[j][m].x, xb[j][m].y, hi.z, lo.z);
            split2<F16>(xb[j][m].z, xb[j][m].w, hi.w, lo.w);
            planes[base + m] = hi;
            planes[img + base + m] = lo;
          }
        }
      }
    }
    __syncthreads();
    // ---- this wave's k steps inside the chunk
    const int s_end = (k0 + kc) >> 6;
    for (;;) {
      const int s = wave + 4 * it;
      if (s >= s_end) break;
      const int slot = it % kD;
      uint4 w0, w1;
      // static ring indexing (registers cannot be indexed dynamically): one unrolled case per slot
#pragma unroll
      for (int d = 0; d < kD; ++d) {
        if (slot == d) {
          w0 = ring[d][0];
          w1 = ring[d][1];
          if (s + 4 * kD < steps_total) issue(s + 4 * kD, ring[d]);
        }
      }
      const int ls = s - (k0 >> 6);                       // step inside the chunk
      const int p0 = ((ls * 2 + 0) * 4 + gi) * 8 + (li & 7);
      const int p1 = ((ls * 2 + 1) * 4 + gi) * 8 + (li & 7);
      const uint4 h0 = planes[p0], h1 = planes[p1], l0 = planes[img + p0], l1 = planes[img + p1];
      acc = mfma_16<F16>(w0, h0, acc);
      acc = mfma_16<F16>(w1, h1, acc);
      acc = mfma_16<F16>(w0, l0, acc);
      acc = mfma_16<F16>(w1, l1, acc);
      ++it;
    }
  }
  // ---- split-K: the four waves' 16 x 16 partial tiles through LDS.  D layout: lane holds column (lane & 15) = input row m, rows 4 (lane >> 4) + r = n
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * gi + r) * 16 + li] = acc[r];
  __syncthreads();
  const int i = tid & 15, m = tid >> 4;  // consecutive threads -> consecutive output columns n of one input row m
  const int n = n0 + i;
  if (m >= M || n >= a.N) return;
  const float v0 = (red[0][i * 16 + m] + red[1][i * 16 + m]) + (red[2][i * 16 + m] + red[3][i * 16 + m]);
  if (a.glu) {  // rows come in (gate, up) pairs: the even thread of a pair finishes both
    if (i & 1) return;
    const float v1 = (red[0][(i + 1) * 16 + m] + red[1][(i + 1) * 16 + m]) + (red[2][(i + 1) * 16 + m] + red[3][(i + 1) * 16 + m]);
    const float g = v0 + (a.bias ? a.bias[n] : 0.f), u = v1 + (a.bias ? a.bias[n + 1] : 0.f);
    a.y[(int64_t)m * a.ldy + (n >> 1)] = (g / (1.0f + expf(-g))) * u * a.out_scale;
    return;
  }
  float v = mfma_act(v0 + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
  if (a.res) v += a.res[(int64_t)m * a.ldr + n];
  if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);
  else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
}

// ---------------------------------------------------------------------------------------------- streaming variant (no fused norm, any K % 64 == 0)
// Without a fused norm nothing forces the workgroup to see whole rows, so each WAVE stages only the x columns of ITS OWN k steps, in a private 2 KB
// LDS window, and there is no workgroup barrier before the final split-K reduction: lane (row = lane >> 3, part = lane & 7) loads the 8 floats
// x[row][64 s + 8 part .. + 8) of step s (prefetched kDX steps ahead next to the weight ring), splits them into the hi / lo images and writes one
// 16-byte piece each in the same fragment order as above; the LDS queue of a wave is in order, a wave-level fence separates write and read.
// Opt-in (MI355_GEMV_MFMA_STREAM, see stream_mode below): measured slower than the FMA kernel on the K > 2048 down projections.
constexpr int kDX = 4;

template <bool F16>
__global__ __launch_bounds__(256) void gemv_mfma_stream_kernel(const mi355_gemv_args a) {
  __shared__ __attribute__((aligned(16))) uint4 win[4][2][64];   // [wave][image][piece]
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int K = a.K, M = a.M;
  const int steps_total = K >> 6;
  const int gi = lane >> 4, li = lane & 15;
  const int nrow = n0 + li < a.N ? n0 + li : a.N - 1;
  const uint16_t* wrow = a.w + (int64_t)nrow * a.ldw + 16 * gi;
  const int xr = lane >> 3, xp = lane & 7;                       // staging role: input row / 8-column part of the step
  const float* xrow = a.x + (int64_t)(xr < M ? xr : 0) * a.ldx + 8 * xp;
  const int wpiece = ((xp & 1) * 4 + (xp >> 1)) * 8 + xr;        // (half, group, row) of the piece this lane writes

  uint4 ring[kD][2];
  float4 xring[kDX][2];
  auto issue_w = [&](const int s, uint4 (&dst)[2]) {
    const uint16_t* p = wrow + ((int64_t)s << 6);
    dst[0] = *(const uint4*)p;
    dst[1] = *(const uint4*)(p + 8);
  };
  auto issue_x = [&](const int s, float4 (&dst)[2]) {
    if (xr < M) {
      const float* p = xrow + ((int64_t)s << 6);
      dst[0] = *(const float4*)p;
      dst[1] = *(const float4*)(p + 4);
    } else {
      dst[0] = dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
#pragma unroll
  for (int d = 0; d < kD; ++d)
    if (wave + 4 * d < steps_total) issue_w(wave + 4 * d, ring[d]);
#pragma unroll
  for (int d = 0; d < kDX; ++d)
    if (wave + 4 * d < steps_total) issue_x(wave + 4 * d, xring[d]);

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0;; ++it) {
    const int s = wave + 4 * it;
    if (s >= steps_total) break;
    uint4 w0, w1;
    float4 x0, x1;
    const int slot = it % kD, xslot = it % kDX;
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      if (slot == d) {
        w0 = ring[d][0];
        w1 = ring[d][1];
        if (s + 4 * kD < steps_total) issue_w(s + 4 * kD, ring[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < kDX; ++d) {
      if (xslot == d) {
        x0 = xring[d][0];
        x1 = xring[d][1];
        if (s + 4 * kDX < steps_total) issue_x(s + 4 * kDX, xring[d]);
      }
    }
    uint4 hi, lo;
    split2<F16>(x0.x, x0.y, hi.x, lo.x);
    split2<F16>(x0.z, x0.w, hi.y, lo.y);
    split2<F16>(x1.x, x1.y, hi.z, lo.z);
    split2<F16>(x1.z, x1.w, hi.w, lo.w);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous step's fragment reads are done (in-order LDS queue of the wave)
    __builtin_amdgcn_wave_barrier();
    win[wave][0][wpiece] = hi;
    win[wave][1][wpiece] = lo;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int p0 = (0 * 4 + gi) * 8 + (li & 7), p1 = (1 * 4 + gi) * 8 + (li & 7);
    const uint4 h0 = win[wave][0][p0], h1 = win[wave][0][p1], l0 = win[wave][1][p0], l1 = win[wave][1][p1];
    acc = mfma_16<F16>(w0, h0, acc);
    acc = mfma_16<F16>(w1, h1, acc);
    acc = mfma_16<F16>(w0, l0, acc);
    acc = mfma_16<F16>(w1, l1, acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * gi + r) * 16 + li] = acc[r];
  __syncthreads();
  const int i = tid & 15, m = tid >> 4;
  const int n = n0 + i;
  if (m >= M || n >= a.N) return;
  const float v0 = (red[0][i * 16 + m] + red[1][i * 16 + m]) + (red[2][i * 16 + m] + red[3][i * 16 + m]);
  if (a.glu) {
    if (i & 1) return;
    const float v1 = (red[0][(i + 1) * 16 + m] + red[1][(i + 1) * 16 + m]) + (red[2][(i + 1) * 16 + m] + red[3][(i + 1) * 16 + m]);
    const float g = v0 + (a.bias ? a.bias[n] : 0.f), u = v1 + (a.bias ? a.bias[n + 1] : 0.f);
    a.y[(int64_t)m * a.ldy + (n >> 1)] = (g / (1.0f + expf(-g))) * u * a.out_scale;
    return;
  }
  float v = mfma_act(v0 + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
  if (a.res) v += a.res[(int64_t)m * a.ldr + n];
  if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);
  else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
}

// ---------------------------------------------------------------------------------------------- K split over workgroups (no fused norm, K > 2048)
// The down projections (N = 1024 .. 2048, K = 3072 .. 8192) have few column tiles and a long K: N / 16 workgroups walking 2-4 chunks one after the
// other leave the chip empty (measured: slower than the FMA kernel, call 24).  Here a workgroup owns ONE (tile, 2048-column chunk) pair -- the grid is
// tiles x chunks -- and the chunks of a tile meet through a scratch record and a ticket: the workgroup that draws the last ticket adds the records
// in chunk order (deterministic) and runs the epilogue.  Nobody waits for anybody.
template <bool F16>
__global__ __launch_bounds__(256) void gemv_mfma_ksplit_kernel(const mi355_gemv_args a, const int nch) {
  __shared__ __attribute__((aligned(16))) uint4 planes[2 * kKC * 8 / 8];  // [2 images][32 steps][2 halves][4 groups][8 rows] 16-byte pieces: 64 KB
  __shared__ float red[4][256];
  __shared__ int ticket_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, c = blockIdx.y;
  const int n0 = tile * 16;
  const int K = a.K, M = a.M;
  const int k0 = c * kKC;
  const int kc = K - k0 < kKC ? K - k0 : kKC;   // multiple of 64
  const int steps = kc >> 6;
  constexpr int img = kKC;                      // 16-byte pieces per image
  const int gi = lane >> 4, li = lane & 15;
  const int nrow = n0 + li < a.N ? n0 + li : a.N - 1;
  const uint16_t* wrow = a.w + (int64_t)nrow * a.ldw + k0 + 16 * gi;
  uint4 ring[kD][2];
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    ring[d][0] = ring[d][1] = make_uint4(0u, 0u, 0u, 0u);
    if (wave + 4 * d < steps) {
      const uint16_t* p = wrow + ((int64_t)(wave + 4 * d) << 6);
      ring[d][0] = *(const uint4*)p;
      ring[d][1] = *(const uint4*)(p + 8);
    }
  }
  {  // stage the chunk: thread t owns the 8-column group q = t of every row
    const int q = tid, k = q * 8;
    if (k < kc) {
      const int step = q >> 3, r = q & 7, g = r >> 1, h = r & 1;
      const int base = ((step * 2 + h) * 4 + g) * 8;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (m < M) {
          const float* p = a.x + (int64_t)m * a.ldx + k0 + k;
          xa = *(const float4*)p;
          xb = *(const float4*)(p + 4);
        }
        uint4 hi, lo;
        split2<F16>(xa.x, xa.y, hi.x, lo.x);
        split2<F16>(xa.z, xa.w, hi.y, lo.y);
        split2<F16>(xb.x, xb.y, hi.z, lo.z);
        split2<F16>(xb.z, xb.w, hi.w, lo.w);
        planes[base + m] = hi;
        planes[img + base + m] = lo;
      }
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < kD; ++d) {
    const int sl = wave + 4 * d;
    if (sl < steps) {
      const int p0 = ((sl * 2 + 0) * 4 + gi) * 8 + (li & 7);
      const int p1 = ((sl * 2 + 1) * 4 + gi) * 8 + (li & 7);
      acc = mfma_16<F16>(ring[d][0], planes[p0], acc);
      acc = mfma_16<F16>(ring[d][1], planes[p1], acc);
      acc = mfma_16<F16>(ring[d][0], planes[img + p0], acc);
      acc = mfma_16<F16>(ring[d][1], planes[img + p1], acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * gi + r) * 16 + li] = acc[r];
  __syncthreads();
  const int i = tid & 15, m = tid >> 4;   // consecutive threads -> consecutive output columns n of one input row m
  const int n = n0 + i;
  float v0 = (red[0][i * 16 + m] + red[1][i * 16 + m]) + (red[2][i * 16 + m] + red[3][i * 16 + m]);
  if (nch > 1) {
    float* rec = a.split_ws + ((int64_t)tile * nch + c) * 256;
    rec[tid] = v0;
    __threadfence();
    __syncthreads();
    if (tid == 0) ticket_s = atomicAdd(a.split_cnt + tile, 1);
    __syncthreads();
    if (ticket_s != nch - 1) return;
    __threadfence();
    const float* recs = a.split_ws + (int64_t)tile * nch * 256;
    v0 = 0.f;
    for (int cc = 0; cc < nch; ++cc) v0 += __hip_atomic_load(recs + cc * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) a.split_cnt[tile] = 0;   // leave the ticket zeroed for the next launch
  }
  if (m >= M || n >= a.N) return;
  float v = mfma_act(v0 + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
  if (a.res) v += a.res[(int64_t)m * a.ldr + n];
  if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);
  else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
}

}  // namespace

// 1 = this call qualifies for the matrix-pipe kernel (mi355_gemv dispatches here unless MI355_GEMV_MFMA=0)
// MI355_GEMV_MFMA_STREAM: 0 (default) = streaming variant off: K > 2048 stays on the FMA kernel; 1 = for images the staged kernel cannot take
// (no fused norm, K > 2048); 2 = for every call without a fused norm.  Parity-green in both modes (GPU call 35), but with N / 16 workgroups a wave
// streams its whole K slice serially: Qwen3-TTS-1.7B ran 8.55 ms per frame with mode 1 against 7.81 with the FMA kernel on those images (24
// steps per wave at K = 6144), so it is an A/B harness until it splits K over more waves.
static int stream_mode() {
  static const int mode = getenv("MI355_GEMV_MFMA_STREAM") ? atoi(getenv("MI355_GEMV_MFMA_STREAM")) : 0;
  return mode;
}

static bool ksplit_on() {
  static const bool on = getenv("MI355_GEMV_KSPLIT") != nullptr && getenv("MI355_GEMV_KSPLIT")[0] == '1';
  return on;
}

static bool use_stream(const mi355_gemv_args& a) {
  if (a.norm) return false;
  return stream_mode() == 2 || (stream_mode() == 1 && a.K > kKC);
}

int mi355_gemv_mfma_eligible(const mi355_gemv_args& a) {
  static const bool off = getenv("MI355_GEMV_MFMA") != nullptr && getenv("MI355_GEMV_MFMA")[0] == '0';
  if (off) return 0;
  if (a.M < 5 || a.M > 8 || a.rope_cos) return 0;
  if (a.wdtype != MI355_W_BF16 && a.wdtype != MI355_W_F16) return 0;
  if (a.K % 64 || a.K < 64 || a.ldw % 8 || ((uintptr_t)a.w) % 16 || a.ldx % 4 || ((uintptr_t)a.x) % 16) return 0;
  // the staged kernel holds one chunk: its chunk loop's workgroup barriers were measured to cost more than they save (r1 call 26);
  // MI355_GEMV_MFMA_CHUNKED=1 lets no-norm images with K > 2048 take the chunk loop anyway (A/B knob)
  static const bool chunked = getenv("MI355_GEMV_MFMA_CHUNKED") != nullptr && getenv("MI355_GEMV_MFMA_CHUNKED")[0] == '1';
  // K split over workgroups (gemv_mfma_ksplit_kernel): OPT-IN (MI355_GEMV_KSPLIT=1).  Measured in call 29: 19.7 / 43.5 us per launch on the Qwen3
  // down projections against 13.9 / 20.1 us for the FMA kernel -- the agent-scope fence in front of the ticket writes the L2 back, which costs
  // more than the extra workgroups bring (the same finding as the key-split attention of round 1)
  if (ksplit_on() && a.K > kKC && !a.norm && !a.glu && a.split_ws && a.split_cnt) return 1;
  if (a.K > kKC && !use_stream(a) && !(chunked && !a.norm)) return 0;
  if (a.glu && (a.N % 2)) return 0;
  return 1;
}

int mi355_gemv_mfma_launch(const mi355_gemv_args& a, hipStream_t st) {
  static bool attr_set[2] = {false, false};  // benign race: the attribute is idempotent
  const bool f16 = a.wdtype == MI355_W_F16;
  const dim3 grid((a.N + 15) / 16);
  if (ksplit_on() && a.K > kKC && !a.norm && !a.glu && a.split_ws && a.split_cnt && !use_stream(a)) {
    const int nch = (a.K + kKC - 1) / kKC;
    MI355_CLEAR_ERROR();
    if (f16) hipLaunchKernelGGL(gemv_mfma_ksplit_kernel<true>, dim3(grid.x, nch), dim3(256), 0, st, a, nch);
    else hipLaunchKernelGGL(gemv_mfma_ksplit_kernel<false>, dim3(grid.x, nch), dim3(256), 0, st, a, nch);
    MI355_LAUNCH_CHECK("gemv(mfma, K split over workgroups)");
    return MI355_OK;
  }
  if (use_stream(a)) {
    MI355_CLEAR_ERROR();
    if (f16) hipLaunchKernelGGL(gemv_mfma_stream_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemv_mfma_stream_kernel<false>, grid, dim3(256), 0, st, a);
    MI355_LAUNCH_CHECK("gemv(mfma stream)");
    return MI355_OK;
  }
  const size_t lds = (size_t)(a.K < kKC ? a.K : kKC) * 32;
  if (!attr_set[f16]) {
    hipError_t e = f16 ? hipFuncSetAttribute((const void*)gemv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kKC * 32)
                       : hipFuncSetAttribute((const void*)gemv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kKC * 32);
    MI355_REQUIRE(e == hipSuccess, "gemv(mfma): cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set[f16] = true;
  }
  MI355_CLEAR_ERROR();
  if (f16) hipLaunchKernelGGL(gemv_mfma_kernel<true>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(gemv_mfma_kernel<false>, grid, dim3(256), lds, st, a);
  MI355_LAUNCH_CHECK("gemv(mfma)");
  return MI355_OK;
}
