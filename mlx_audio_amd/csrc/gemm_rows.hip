// Skinny GEMM for 9..64 rows per decode step on the matrix pipe (gfx950): y[m, n] = epilogue(sum_k norm(x)[m, k] * W[n, k]).
//
// Same contract as mi355_gemv (gemv.hip).  It replaces nn.Linear at sequence length 1 for a BATCH of sequences -- the reference's batched generation
// (tts/models/qwen3_tts/qwen3_tts.py:1651-2060 batch_generate, talker.py:229-336; BASELINE config[3] names 64 utterances): at 64 rows the
// weights of a step are still read exactly once, so the step stays a weight stream (2 bytes per weight, 256 MFMA flops per weight at hi + lo),
// and the bytes per generated token fall 8x against the 8-row kernels.
//
//   * A operand = a 16-row tile of W straight from HBM (as gemv_mfma.hip): lane (i = lane & 15, g = lane >> 4) loads the 32 contiguous bytes
//     W[n0 + i][k0 + 16 g .. + 16) of a 64-wide k step -- full 128-byte lines per row -- and feeds its two 16-byte halves to MFMAs.
//   * B operand = the (normalised) input rows, R = 16 / 32 / 64 of them, split into hi + lo images of the weights' 16-bit type (the split the
//     prefill GEMMs and the 5..8-row kernel use: ~16 mantissa bits for bf16, ~22 for fp16) in LDS fragment order [step][half][group][row] x 16 bytes.
//     64 rows x K columns of two images do not fit LDS, so K is walked in chunks of KC = 1024 / (R / 16) columns (64 KB of LDS, two workgroups
//     per CU).  The four waves of a workgroup take interleaved k steps (split K) -- and a wave stages exactly the columns of ITS OWN steps, all
//     rows: a wave's LDS window is private, so the main loop has NO workgroup barrier; the LDS queue of a wave is in order and a wave-level fence
//     separates the writes of a chunk from its fragment reads.
//   * A workgroup owns T column tiles at a time (tile = (group * T + j) * gridDim.x + blockIdx.x): one staged chunk feeds T x (R / 16) x 4 MFMAs per
//     k step.  Weights of the next P chunks (16 KB per wave) and the input rows of the next chunk are always in flight.
//   * the four partial tiles meet in LDS after the last chunk; thread (n = t & 15, m = t >> 4) finishes the outputs: bias, activation, LayerScale,
//     residual, SwiGLU pairs, split destinations (q -> buffer, k | v -> KV-cache slot, 16-bit slots included).
//   * fused LayerNorm / RMSNorm: a statistics pass over the rows (one read of x from L2, fp64 sums) precedes the chunk loop, any K.
// Rows m >= M of the MFMA column space are staged as zeros and never stored.
#include <stdlib.h>
#include "common.h"

namespace {

template <int MR> struct rows_cfg {
  static constexpr int R = 16 * MR;          // input rows staged (MFMA column groups of 16)
  static constexpr int KC = 1024 / MR;       // columns per chunk: R x KC x 4 bytes = 64 KB
  static constexpr int SPW = (KC / 64) / 4;  // k steps per wave and chunk: 4 / MR
  static constexpr int T = MR == 1 ? 1 : 2;  // column tiles per workgroup pass
  static constexpr int P = 2;                // chunks of weights in flight per wave (MR = 4: 4 x 2 KB, else 8 x 2 KB; 8 waves per CU)
};

__device__ __forceinline__ float rows_act(float v, int act, float slope) {
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
__device__ __forceinline__ void rows_split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
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
__device__ __forceinline__ f32x4 rows_mfma(const uint4 a, const uint4 b, const f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LSTM mode (mi355_lstm_seq with gate-interleaved Wh rows, lstm_seq.hip): the GEMM is one recurrence step pre = h @ Wh^T; its epilogue adds the
// x-projection row, applies the gates (the reference's Metal kernel: codec/models/encodec/encodec.py:89-134 -- chunks i | f | g | o, sigmoid as
// 1 / (1 + exp(-|x|)) mirrored for x < 0, precise tanh), updates c in place and stores the new h to the OTHER h buffer (a.y) and to row t of the output
struct rows_lstm_t {
  const float* xproj; int64_t xproj_bstride;   // row t of the x-projection of sequence m: xproj + m * xproj_bstride, gate blocks of H
  float* c;                                     // [B, H]
  float* out; int64_t out_bstride;              // row t of the output of sequence m
  int H;
};

__device__ __forceinline__ float rows_lstm_sigmoid(const float x) {
  const float y = 1.0f / (1.0f + expf(-fabsf(x)));
  return x < 0.f ? 1.0f - y : y;
}

template <int MR, bool F16, bool LSTM = false>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const mi355_gemv_args a, const int ntiles, const int dbg, const rows_lstm_t L) {
  using C = rows_cfg<MR>;
  constexpr int R = C::R, KC = C::KC, SPW = C::SPW, T = C::T, P = C::P;
  constexpr int IMGW = SPW * 8 * R;     // 16-byte pieces per image of ONE wave's window
  extern __shared__ __attribute__((aligned(16))) uint4 planes[];   // [wave][image][step of the wave][half][group][row]: 4 x 2 x IMGW pieces = 64 KB
  __shared__ float st_mean[64], st_rstd[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, M = a.M, G = gridDim.x;
  const int nch = (K + KC - 1) / KC;
  const int gi = lane >> 4, li = lane & 15;   // MFMA k group / row of W (A) resp. input row inside a row group (B)
  const int srow = lane & 7, sq = lane >> 3;  // staging role: row inside an 8-row block / 8-column piece of the step
  uint4* const win = planes + wave * (2 * IMGW);

  // ---- fused norm: statistics of every row (wave w takes rows w, w + 4, ...), fp64 sums of one pass
  if (a.norm && !(dbg & 32)) {
    for (int m = wave; m < M; m += 4) {
      const float* xr = a.x + (int64_t)m * a.ldx;
      double s = 0.0, ss = 0.0;
      for (int k = 4 * lane; k < K; k += 256) {
        const float4 v = *(const float4*)(xr + k);
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      }
      s = wave_sum_d(s);
      ss = wave_sum_d(ss);
      if (lane == 0) {
        const double mean = a.norm == 1 ? s / K : 0.0;
        const double var = ss / K - mean * mean;
        const float vf = (float)(var > 0.0 ? var : 0.0);
        st_mean[m] = (float)mean;
        st_rstd[m] = a.norm == 1 ? 1.0f / sqrtf(vf + a.norm_eps) : rsqrtf(vf + a.norm_eps);
      }
    }
    __syncthreads();
  }

  for (int tg = 0; tg * T * G < ntiles; ++tg) {
    const uint16_t* wrow[T];
    bool tv[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int tile = (tg * T + j) * G + blockIdx.x;
      tv[j] = tile < ntiles;
      const int n0 = (tv[j] ? tile : ntiles - 1) * 16;
      const int nrow = n0 + li < a.N ? n0 + li : a.N - 1;   // tail tile: clamped rows recompute the last row, never stored
      wrow[j] = a.w + (int64_t)nrow * a.ldw + 16 * gi;
    }
    f32x4 acc[T][MR];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < MR; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ring[P][T][SPW][2];
    float4 xr[SPW][2 * MR][2];   // the next chunk's input rows: pass (u, rb) -> row 8 rb + srow, columns (wave + 4 u) * 64 + 8 sq .. + 8

    auto issue_w = [&](const int c, uint4 (&dst)[T][SPW][2]) {
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int u = 0; u < SPW; ++u) {
          const int k = c * KC + (wave + 4 * u) * 64;
          if (tv[j] && k < K && !(dbg & 16)) {
            const uint16_t* p = wrow[j] + k;
            dst[j][u][0] = *(const uint4*)p;
            dst[j][u][1] = *(const uint4*)(p + 8);
          }
        }
    };
    auto issue_x = [&](const int c) {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        const int k = c * KC + (wave + 4 * u) * 64 + 8 * sq;
#pragma unroll
        for (int rb = 0; rb < 2 * MR; ++rb) {
          const int m = 8 * rb + srow;
          xr[u][rb][0] = xr[u][rb][1] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < M && k < K && !(dbg & 1)) {
            const float* p = a.x + (int64_t)m * a.ldx + k;
            xr[u][rb][0] = *(const float4*)p;
            xr[u][rb][1] = *(const float4*)(p + 4);
          }
        }
      }
    };
    // the chunk held in xr -> hi / lo pieces of this wave's window
    auto stage = [&](const int c) {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        const int k = c * KC + (wave + 4 * u) * 64 + 8 * sq;
        float4 w0 = make_float4(1.f, 1.f, 1.f, 1.f), w1 = w0, b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (a.norm && k < K) {
          if (a.norm_weight) { w0 = *(const float4*)(a.norm_weight + k); w1 = *(const float4*)(a.norm_weight + k + 4); }
          if (a.norm_bias) { b0 = *(const float4*)(a.norm_bias + k); b1 = *(const float4*)(a.norm_bias + k + 4); }
        }
        const int base = ((u * 2 + (sq & 1)) * 4 + (sq >> 1)) * R;   // piece (step u, half, group, row 0)
#pragma unroll
        for (int rb = 0; rb < 2 * MR; ++rb) {
          const int m = 8 * rb + srow;
          float4 v0 = xr[u][rb][0], v1 = xr[u][rb][1];
          if (a.norm) {
            const float mu = st_mean[m < M ? m : 0], rs = m < M ? st_rstd[m] : 0.f;
            v0 = make_float4((v0.x - mu) * rs * w0.x + b0.x, (v0.y - mu) * rs * w0.y + b0.y, (v0.z - mu) * rs * w0.z + b0.z, (v0.w - mu) * rs * w0.w + b0.w);
            v1 = make_float4((v1.x - mu) * rs * w1.x + b1.x, (v1.y - mu) * rs * w1.y + b1.y, (v1.z - mu) * rs * w1.z + b1.z, (v1.w - mu) * rs * w1.w + b1.w);
            if (m >= M) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          uint4 hi, lo;
          if (dbg & 2) {
            hi = __builtin_bit_cast(uint4, v0); lo = __builtin_bit_cast(uint4, v1);
          } else {
          rows_split2<F16>(v0.x, v0.y, hi.x, lo.x);
          rows_split2<F16>(v0.z, v0.w, hi.y, lo.y);
          rows_split2<F16>(v1.x, v1.y, hi.z, lo.z);
          rows_split2<F16>(v1.z, v1.w, hi.w, lo.w);
          }
          if (!(dbg & 4)) {
          win[base + m] = hi;
          win[IMGW + base + m] = lo;
          }
        }
      }
    };

#pragma unroll
    for (int p = 0; p < P; ++p)
      if (p < nch) issue_w(p, ring[p]);
    issue_x(0);

    for (int c0 = 0; c0 < nch; c0 += P) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int c = c0 + p;
        if (c < nch) {
          wave_lds_fence();   // the previous chunk's fragment reads are done (in-order LDS queue of the wave)
          stage(c);
          if (c + 1 < nch) issue_x(c + 1);
          wave_lds_fence();
#pragma unroll
          for (int u = 0; u < SPW; ++u) {
            if (c * KC + (wave + 4 * u) * 64 < K && !(dbg & 8)) {
              uint4 w0[T], w1[T];
#pragma unroll
              for (int j = 0; j < T; ++j) { w0[j] = ring[p][j][u][0]; w1[j] = ring[p][j][u][1]; }
              const int pb0 = ((u * 2 + 0) * 4 + gi) * R + li, pb1 = ((u * 2 + 1) * 4 + gi) * R + li;
#pragma unroll
              for (int r = 0; r < MR; ++r) {
                const uint4 h0 = win[pb0 + 16 * r], h1 = win[pb1 + 16 * r], l0 = win[IMGW + pb0 + 16 * r], l1 = win[IMGW + pb1 + 16 * r];
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j][r] = rows_mfma<F16>(w0[j], h0, acc[j][r]);
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j][r] = rows_mfma<F16>(w1[j], h1, acc[j][r]);
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j][r] = rows_mfma<F16>(w0[j], l0, acc[j][r]);
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j][r] = rows_mfma<F16>(w1[j], l1, acc[j][r]);
              }
            }
          }
          if (c + P < nch) issue_w(c + P, ring[p]);
        }
      }
    }

    // ---- split K: the four waves' partial tiles through LDS (aliases the windows).  D layout: lane holds column li = input row, rows 4 gi + r = n
    __syncthreads();
    float* const red = (float*)planes;   // [wave][tile j][row group][256]
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((wave * T + j) * MR + r) * 256 + (4 * gi + e) * 16 + li] = acc[j][r][e];
    __syncthreads();
    const int i = tid & 15, ml = tid >> 4;   // consecutive threads -> consecutive output columns n of one input row
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int tile = (tg * T + j) * G + blockIdx.x;
      const int n = tile * 16 + i;
      if (tile >= ntiles || n >= a.N) continue;
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        const int m = 16 * r + ml;
        if (m >= M) continue;
        auto part = [&](const int ii) {
          const int o = (j * MR + r) * 256 + ii * 16 + ml;
          return (red[o] + red[T * MR * 256 + o]) + (red[2 * T * MR * 256 + o] + red[3 * T * MR * 256 + o]);
        };
        if constexpr (LSTM) {   // rows of W come as (i, f, g, o) of one hidden unit: the first thread of a quad finishes the unit
          if (i & 3) continue;
          const int j = n >> 2;
          const float* xp = L.xproj + (int64_t)m * L.xproj_bstride + j;
          const float gi_ = rows_lstm_sigmoid(part(i) + xp[0]), gf = rows_lstm_sigmoid(part(i + 1) + xp[L.H]);
          const float gg = tanhf(part(i + 2) + xp[2 * L.H]), go = rows_lstm_sigmoid(part(i + 3) + xp[3 * L.H]);
          const int64_t ci = (int64_t)m * L.H + j;
          const float cn = gf * L.c[ci] + gi_ * gg;
          const float hn = go * tanhf(cn);
          L.c[ci] = cn;
          a.y[(int64_t)m * a.ldy + j] = hn;
          L.out[(int64_t)m * L.out_bstride + j] = hn;
          continue;
        }
        const float v0 = part(i);
        if (a.glu) {   // rows of W come in (gate, up) pairs: the even thread of a pair finishes both
          if (i & 1) continue;
          const float g = v0 + (a.bias ? a.bias[n] : 0.f), u = part(i + 1) + (a.bias ? a.bias[n + 1] : 0.f);
          a.y[(int64_t)m * a.ldy + (n >> 1)] = (g / (1.0f + expf(-g))) * u * a.out_scale;
          continue;
        }
        float v = rows_act(v0 + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
        if (a.res) v += a.res[(int64_t)m * a.ldr + n];
        if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);
        else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
      }
    }
    __syncthreads();   // the next tile group restages the windows
  }
}

template <int MR, bool F16, bool LSTM = false>
int launch_rows(const mi355_gemv_args& a, hipStream_t st, const rows_lstm_t L = rows_lstm_t{}) {
  static bool attr_set = false;  // benign race: the attribute is idempotent
  constexpr size_t lds = 64 * 1024;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_rows_kernel<MR, F16, LSTM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MI355_REQUIRE(e == hipSuccess, "gemv(rows): cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int ntiles = (a.N + 15) / 16;
  static const int wgs_env = getenv("MI355_GEMM_ROWS_WGS") ? atoi(getenv("MI355_GEMM_ROWS_WGS")) : 0;   // A/B knob: workgroups of the launch
  const int cap = wgs_env > 0 ? wgs_env : 512;   // two 64 KB workgroups per CU
  const int grid = ntiles < cap ? ntiles : cap;
  MI355_CLEAR_ERROR();
  static const int dbg = getenv("MI355_GEMM_ROWS_DBG") ? atoi(getenv("MI355_GEMM_ROWS_DBG")) : 0;   // ablation bits (timing experiments only)
  hipLaunchKernelGGL((gemm_rows_kernel<MR, F16, LSTM>), dim3(grid), dim3(256), lds, st, a, ntiles, dbg, L);
  MI355_LAUNCH_CHECK("gemv(9..64 rows, matrix pipe)");
  return MI355_OK;
}

}  // namespace

// 1 = this call runs on the 9..64-row kernel
int mi355_gemm_rows_eligible(const mi355_gemv_args& a) {
  if (a.M < 9 || a.M > 64) return 0;
  if (a.wdtype != MI355_W_BF16 && a.wdtype != MI355_W_F16) return 0;
  if (a.K % 64 || a.K < 64 || a.ldw % 8 || ((uintptr_t)a.w) % 16 || a.ldx % 4 || ((uintptr_t)a.x) % 16) return 0;
  if (a.rope_cos || a.x_ids) return 0;
  if (a.glu && (a.N % 2)) return 0;
  return 1;
}

int mi355_gemm_rows_launch(const mi355_gemv_args& a, hipStream_t st) {
  const bool f16 = a.wdtype == MI355_W_F16;
  if (a.M <= 16) return f16 ? launch_rows<1, true>(a, st) : launch_rows<1, false>(a, st);
  if (a.M <= 32) return f16 ? launch_rows<2, true>(a, st) : launch_rows<2, false>(a, st);
  return f16 ? launch_rows<4, true>(a, st) : launch_rows<4, false>(a, st);
}

// One step of mi355_lstm_seq on gate-interleaved Wh rows (lstm_seq.hip): a = { x = h_in [B, H], w = Wh [4 H, H] (row 4 j + g), y = h_out, ldy = H, M = B, K = H,
// N = 4 H }; any B in 1..64 (rows >= B of the MFMA column space are staged as zeros and never stored)
int mi355_gemm_rows_lstm_step(const mi355_gemv_args& a, const float* xproj, int64_t xproj_bstride, float* c, float* out, int64_t out_bstride, hipStream_t st) {
  const rows_lstm_t L{xproj, xproj_bstride, c, out, out_bstride, a.K};
  const bool f16 = a.wdtype == MI355_W_F16;
  if (a.M <= 16) return f16 ? launch_rows<1, true, true>(a, st, L) : launch_rows<1, false, true>(a, st, L);
  if (a.M <= 32) return f16 ? launch_rows<2, true, true>(a, st, L) : launch_rows<2, false, true>(a, st, L);
  return f16 ? launch_rows<4, true, true>(a, st, L) : launch_rows<4, false, true>(a, st, L);
}
