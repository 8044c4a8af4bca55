// Skinny GEMM for the autoregressive decode steps (1..8 rows): y[m, n] = epilogue(sum_k x[m, k] * W[n, k])  (gfx950).
//
// Replaces nn.Linear / nn.Embedding.as_linear at sequence length 1 under the Whisper TextDecoder
// (stt/models/whisper/whisper.py:347-416, 498), the Qwen3-TTS talker / code predictor (tts/models/qwen3_tts/talker.py:
// 230-330, 503-764) and the CSM Llama backbone / depth decoder (lm/models/llama.py:46-198, tts/models/sesame/sesame.py:361-404).
// These steps read every weight once per generated token and do 2*M flops per weight: HBM-bound by a wide margin, so
// the MFMA tile kernel (conv_gemm) is the wrong tool -- a 64-row tile with <= 8 live rows and N/128 workgroups cannot pull
// bandwidth.  Here W stays in its natural row-major 16-bit layout [N, ldw] (bf16 or fp16, the checkpoint dtype); one
// wavefront owns NC output columns, lanes stride K in 16-byte (8-element) pieces so every weight load is a fully coalesced
// 1 KB wave access, x is staged in LDS once per workgroup in K-chunks, products accumulate in fp32 FMAs (exact on the
// 16-bit weights) and a wave-level butterfly finishes each column.  Epilogue: bias, activation, residual, scale, and the
// optional fused SwiGLU (interleaved gate / up rows: y[n/2] = silu(acc[n]) * acc[n+1]).  Prologue (optional): LayerNorm / RMSNorm of
// the input rows, recomputed per workgroup; the output columns can be split over two destinations (q | k,v -> buffer | KV-cache slot).
// Weight element types: bf16, fp16, or fp8 (OCP e4m3fn bytes + one power-of-two scale per row, mi355_pack_rowmajor_fp8_host): a lane's 16-byte
// piece then carries 16 elements, decoded exactly by moving the byte into binary16 position (common.h cvt_w16).
// Dispatch: calls with 5..8 rows, 16-bit weights and K <= 2048 (K % 64 == 0) go to the matrix-pipe kernel of gemv_mfma.hip (the FMA kernel runs
// the weight stream at ~1.4 TB/s at 8 rows against ~3.7 TB/s at 1 row); everything else runs here.  What bounds the small images is the
// dependent chain inside a launch (x load -> statistics -> staging -> reduction), hence: statistics from the staged LDS copy (one read of x),
// a staging thread owns columns (norm weight / bias loaded once per column: fewer registers, more waves in flight), weights issued first.
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace {

// Workgroups of 256 threads (one wave per SIMD) the chip holds at once for kernel `kern` on the CURRENT device: workgroups per CU = waves per
// SIMD = floor(512 / registers allocated in granules of 8), at most 8 (the occupancy API answered 4 for the 224-register instantiation, which
// the hardware runs at 2: round-2 call 5), times the CU count.  Cached per (kernel address, device): the launchers below take the kernel as a
// generic-lambda argument, and every instantiation of one kernel template decays to the same function-pointer TYPE, so a `static` inside the
// lambda is shared by all of them (the first variant launched would size every other variant's grid).
int resident_workgroups(const void* kern) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_pair(kern, dev);
  const auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 2, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kern) == hipSuccess && fa.numRegs > 0) {
    per_cu = 512 / (((fa.numRegs + 7) / 8) * 8);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  }
  return cache[key] = per_cu * cus;
}


__device__ __forceinline__ float gemv_act(float v, int act, float slope) {
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

// the epilogue of NC complete column sums starting at column n0 (bias, activation, LayerScale, residual, SwiGLU, rotary pair, split destinations),
// executed by whichever lane holds them
template <int MT, int NC>
__device__ __forceinline__ void gemv_epilogue(const mi355_gemv_args& a, float (&acc)[NC][MT], const int n0) {
  if (a.glu) {  // columns come in (gate, up) pairs: NC is even on this path
#pragma unroll
    for (int c = 0; c + 1 < NC; c += 2) {
      const int n = n0 + c;
      if (n + 1 >= a.N) break;
      const float bg = a.bias ? a.bias[n] : 0.f, bu = a.bias ? a.bias[n + 1] : 0.f;
      const float wg = a.wscale ? a.wscale[n] * kFp8Unbias : 1.f, wu = a.wscale ? a.wscale[n + 1] * kFp8Unbias : 1.f;  // fp8 images only
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= a.M) break;
        const float g = acc[c][m] * wg + bg, u = acc[c + 1][m] * wu + bu;
        a.y[(int64_t)m * a.ldy + (n >> 1)] = (g / (1.0f + expf(-g))) * u * a.out_scale;
      }
    }
    return;
  }
  if constexpr (NC == 2) {
    if (a.rope_cos && n0 + 1 < a.N && n0 < a.rope_cols) {  // the wave's two columns are one interleaved rotary pair of a q or k head
      const int i = (n0 % a.rope_dh) >> 1;
      const float cs = a.rope_cos[i], sn = a.rope_sin[i];
      const float b0 = a.bias ? a.bias[n0] : 0.f, b1 = a.bias ? a.bias[n0 + 1] : 0.f;
      const float w0 = a.wscale ? a.wscale[n0] * kFp8Unbias : 1.f, w1 = a.wscale ? a.wscale[n0 + 1] * kFp8Unbias : 1.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= a.M) break;
        const float x0 = acc[0][m] * w0 + b0, x1 = acc[1][m] * w1 + b1;
        float y0, y1;
        rope_pair(x0, x1, cs, sn, y0, y1);    // (x * cos) + (rotate(x) * sin), the roundings of head_norm_rope_kernel
        if (a.y2 && n0 >= a.split) {
          store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n0 - a.split), y0 * a.out_scale, a.y2_dtype);
          store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n0 + 1 - a.split), y1 * a.out_scale, a.y2_dtype);
        } else {
          a.y[(int64_t)m * a.ldy + n0] = y0 * a.out_scale;
          a.y[(int64_t)m * a.ldy + n0 + 1] = y1 * a.out_scale;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int n = n0 + c;
    if (n >= a.N) break;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const float cs = a.colscale ? a.colscale[n] : 1.f;
    const float ws = a.wscale ? a.wscale[n] * kFp8Unbias : 1.f;  // fp8 images only (a power of two: exact)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m >= a.M) break;
      float v = gemv_act(acc[c][m] * ws + bias, a.post_act, a.post_slope) * cs;
      if (a.res) v += a.res[(int64_t)m * a.ldr + n];
      if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);  // e.g. q -> y, k|v -> the KV-cache slot
      else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
    }
  }
}

// wave-level reduction of the per-lane partial sums + the epilogue in lane 0
template <int MT, int NC>
__device__ __forceinline__ void gemv_finish(const mi355_gemv_args& a, float (&acc)[NC][MT], const int n0, const int lane) {
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[c][m] = wave_sum_fast(acc[c][m]);   // called by whole waves
  if (lane != 0) return;
  gemv_epilogue<MT, NC>(a, acc, n0);
}

template <int MT, int NC, int WT>
__global__ __launch_bounds__(256, (MT >= 8 && NC == 2 && WT != MI355_W_FP8) ? 3 : 1) void gemv_kernel(const mi355_gemv_args a) {
  constexpr int EPL = mi355_wt<WT>::EPL;                         // weight elements per 16-byte lane piece (8, or 16 for fp8)
  constexpr int ESZ = 16 / EPL;                                  // bytes per weight element
  constexpr int SL = 64 * EPL;                                   // elements per k-slice (one 1 KB wave load per column)
  constexpr int KC = MT >= 8 ? 1024 : (MT == 4 ? 2048 : 4096);  // x elements staged per row and chunk: MT * KC * 4 B <= 32 KB of LDS
  constexpr int IPC = KC / SL;                                   // k-slices per chunk
  constexpr int D = 4;                                           // weight prefetch depth, in k-slices
  __shared__ __attribute__((aligned(16))) float xs[MT * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * NC;
  float acc[NC][MT];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
  const uint8_t* wrow[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int n = n0 + c < a.N ? n0 + c : a.N - 1;  // clamp: tail columns recompute the last row, never stored
    wrow[c] = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
  }
  // The weight stream is a software pipeline of its own: D k-slices (SL elements = 64 lanes x 16 B each) per column are always in
  // flight, issued before the x chunk they will meet is even staged -- the HBM latency of the weights overlaps the L2 latency of x and
  // the workgroup barriers around the LDS staging instead of adding to them.
  const int n_it = (a.K + SL - 1) / SL;  // (tail waves, n0 >= N, run the same loop on the clamped last row: the barriers stay block-uniform)
  uint4 ring[D][NC];
  auto issue = [&](int it, uint4 (&dst)[NC]) {
    const int k = it * SL + lane * EPL;
#pragma unroll
    for (int c = 0; c < NC; ++c) dst[c] = k < a.K ? *(const uint4*)(wrow[c] + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < n_it) issue(d, ring[d]);
  // fused input normalisation (LayerNorm / RMSNorm of the M input rows): every workgroup recomputes the row statistics out of L2 instead of
  // a separate norm launch writing and re-reading the normalised rows.  Rows of up to 2048 elements are read ONCE into registers (mean, then
  // the centred second moment from the same registers: two-pass numerics, one L2 round trip); longer rows take a second read.
  __shared__ float st_mean[8], st_rstd[8];
  // single = the whole row fits one staged chunk: the statistics are then computed from the LDS copy inside the staging step -- ONE L2 round
  // trip for x instead of three (two dependent row reads per wave for the statistics at 8 rows, then the staging read).  Decode-step
  // launches are bound by exactly this dependent chain, not by bytes (profiles/r1_decode_runner_ab_call21.txt).
  const bool single = a.norm && a.K <= KC && !a.norm_two_reads;
  if (a.norm && !single) {
    // both rows of this wave are in flight before the first reduction (rows m and m + 4: one round trip, not two)
    const int m0 = wave, m1 = wave + 4;
    constexpr int NR = MT > 4 ? 2 : 1;  // rows per wave
    if (a.K <= 2048) {
      float4 buf[NR][8];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int m = r ? m1 : m0;
        const float* xr = a.x + (int64_t)(m < a.M ? m : 0) * a.ldx;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = lane * 4 + i * 256;
          buf[r][i] = (m < a.M && k < a.K) ? *(const float4*)(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int m = r ? m1 : m0;
        if (m >= a.M) continue;  // wave-uniform
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (buf[r][i].x + buf[r][i].y) + (buf[r][i].z + buf[r][i].w);
        const float mean = a.norm == 1 ? wave_sum(s) / (float)a.K : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (lane * 4 + i * 256 < a.K) {
            const float d0 = buf[r][i].x - mean, d1 = buf[r][i].y - mean, d2 = buf[r][i].z - mean, d3 = buf[r][i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
        }
        const float var = wave_sum(q) / (float)a.K;
        if (lane == 0) { st_mean[m] = mean; st_rstd[m] = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps); }
      }
    } else {
      for (int m = wave; m < a.M; m += 4) {
        const float* xr = a.x + (int64_t)m * a.ldx;
        float s = 0.f, q = 0.f;
        for (int k = lane * 4; k < a.K; k += 256) { const float4 t = *(const float4*)(xr + k); s += (t.x + t.y) + (t.z + t.w); }
        const float mean = a.norm == 1 ? wave_sum(s) / (float)a.K : 0.f;
        for (int k = lane * 4; k < a.K; k += 256) {
          const float4 t = *(const float4*)(xr + k);
          const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
          q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float var = wave_sum(q) / (float)a.K;
        if (lane == 0) { st_mean[m] = mean; st_rstd[m] = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps); }
      }
    }
  }
  for (int base = 0; base < n_it; base += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int it = base + d;
      if (it >= n_it) break;
      if (it % IPC == 0) {  // next x chunk (block-uniform: every wave walks the same k sequence)
        const int k0 = (it / IPC) * KC;
        const int kc = a.K - k0 < KC ? a.K - k0 : KC;  // multiple of 8 (K % 8 == 0)
        __syncthreads();
        {  // all of this thread's loads of the chunk go out before the first one is consumed (a rolled load -> store loop serialises one
           // L2 latency per iteration).  A thread owns KQ column positions (k = tid * 4 + j * 1024) of EVERY row: the norm weight / bias of a
           // column is loaded once for all rows (8 + 1 + 1 float4 at 8 rows instead of 8 + 8 + 8: the staging no longer sets the register
           // count, and the register count sets how many workgroups a CU keeps in flight across these latency-bound phases).
          constexpr int KQ = KC / 1024;
          float4 tb[KQ][MT], wb[KQ], bb[KQ];
#pragma unroll
          for (int j = 0; j < KQ; ++j) {
            const int k = tid * 4 + j * 1024;
            wb[j] = make_float4(1.f, 1.f, 1.f, 1.f);
            bb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kc && a.norm && a.norm_weight) wb[j] = *(const float4*)(a.norm_weight + k0 + k);
            if (k < kc && a.norm && a.norm_bias) bb[j] = *(const float4*)(a.norm_bias + k0 + k);
#pragma unroll
            for (int m = 0; m < MT; ++m)
              tb[j][m] = (k < kc && m < a.M) ? *(const float4*)(a.x + (int64_t)m * a.ldx + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          if (single) {  // raw rows -> LDS, statistics from LDS (same summation order as the register path), then normalise below from tb
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
              const int k = tid * 4 + j * 1024;
              if (k < kc) {
#pragma unroll
                for (int m = 0; m < MT; ++m) *(float4*)(xs + m * KC + k) = tb[j][m];
              }
            }
            __syncthreads();
            for (int m = wave; m < a.M; m += 4) {
              const float* xr = xs + m * KC;
              float s = 0.f, q = 0.f;
              for (int k = lane * 4; k < a.K; k += 256) { const float4 t = *(const float4*)(xr + k); s += (t.x + t.y) + (t.z + t.w); }
              const float mean = a.norm == 1 ? wave_sum(s) / (float)a.K : 0.f;
              for (int k = lane * 4; k < a.K; k += 256) {
                const float4 t = *(const float4*)(xr + k);
                const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
              }
              const float var = wave_sum(q) / (float)a.K;
              if (lane == 0) { st_mean[m] = mean; st_rstd[m] = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps); }
            }
            __syncthreads();
          }
#pragma unroll
          for (int j = 0; j < KQ; ++j) {
            const int k = tid * 4 + j * 1024;
            if (k < kc) {
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                float4 t = tb[j][m];
                if (a.norm && m < a.M) {
                  const float mu = st_mean[m], rs = st_rstd[m];
                  t = make_float4((t.x - mu) * rs * wb[j].x + bb[j].x, (t.y - mu) * rs * wb[j].y + bb[j].y, (t.z - mu) * rs * wb[j].z + bb[j].z,
                                  (t.w - mu) * rs * wb[j].w + bb[j].w);
                }
                *(float4*)(xs + m * KC + k) = t;
              }
            }
          }
        }
        __syncthreads();
      }
      const int kl = (it % IPC) * SL + lane * EPL;  // position inside the staged chunk
      if (it * SL + lane * EPL < a.K) {
        float wf[NC][EPL];
#pragma unroll
        for (int c = 0; c < NC; ++c) cvt_w16<WT>(ring[d][c], wf[c]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          float xv[EPL];
#pragma unroll
          for (int j4 = 0; j4 < EPL / 4; ++j4) {
            const float4 t = *(const float4*)(xs + m * KC + kl + 4 * j4);
            xv[4 * j4] = t.x; xv[4 * j4 + 1] = t.y; xv[4 * j4 + 2] = t.z; xv[4 * j4 + 3] = t.w;
          }
#pragma unroll
          for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < EPL; ++j) acc[c][m] = fmaf(xv[j], wf[c][j], acc[c][m]);
        }
      }
      if (it + D < n_it) issue(it + D, ring[d]);
    }
  }
  if (n0 >= a.N) return;
  gemv_finish<MT, NC>(a, acc, n0, lane);
}

// Resident-x variant for 4..8 input rows: at M = 8 the staging of x (M * K * 4 bytes per workgroup) costs more L2 traffic than the weight
// rows a 4-wave workgroup consumes, so the rows are staged ONCE per workgroup (dynamic LDS, the fused norm applied on the way in) and the
// waves then walk many column groups (grid-stride) with no barrier inside the loop -- only the weight stream touches memory.
template <int MT, int NC, int WT>
__global__ __launch_bounds__(256) void gemv_res_kernel(const mi355_gemv_args a, const int ngroups) {
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  constexpr int D = 4;
  extern __shared__ __attribute__((aligned(16))) float xr_s[];  // [MT][K]
  __shared__ float st_mean[8], st_rstd[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  if (a.norm) {
    for (int m = wave; m < a.M; m += 4) {
      const float* xr = a.x + (int64_t)m * a.ldx;
      float s = 0.f;
      for (int k = lane * 4; k < K; k += 256) { const float4 t = *(const float4*)(xr + k); s += (t.x + t.y) + (t.z + t.w); }
      const float mean = a.norm == 1 ? wave_sum(s) / (float)K : 0.f;
      float q = 0.f;
      for (int k = lane * 4; k < K; k += 256) {
        const float4 t = *(const float4*)(xr + k);
        const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float var = wave_sum(q) / (float)K;
      if (lane == 0) { st_mean[m] = mean; st_rstd[m] = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps); }
    }
    __syncthreads();
  }
  for (int e = tid * 4; e < MT * K; e += 1024) {
    const int m = e / K, k = e - m * K;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M) {
      t = *(const float4*)(a.x + (int64_t)m * a.ldx + k);
      if (a.norm) {
        const float mu = st_mean[m], rs = st_rstd[m];
        float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.norm_weight) w4 = *(const float4*)(a.norm_weight + k);
        if (a.norm_bias) b4 = *(const float4*)(a.norm_bias + k);
        t = make_float4((t.x - mu) * rs * w4.x + b4.x, (t.y - mu) * rs * w4.y + b4.y, (t.z - mu) * rs * w4.z + b4.z, (t.w - mu) * rs * w4.w + b4.w);
      }
    }
    *(float4*)(xr_s + e) = t;
  }
  __syncthreads();
  const int n_it = (K + SL - 1) / SL;
  for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
    const int n0 = g * NC;
    float acc[NC][MT];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
    const uint8_t* wrow[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = n0 + c < a.N ? n0 + c : a.N - 1;
      wrow[c] = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
    }
    uint4 ring[D][NC];
    auto issue = [&](int it, uint4 (&dst)[NC]) {
      const int k = it * SL + lane * EPL;
#pragma unroll
      for (int c = 0; c < NC; ++c) dst[c] = k < K ? *(const uint4*)(wrow[c] + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n_it) issue(d, ring[d]);
    for (int base = 0; base < n_it; base += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int it = base + d;
        if (it >= n_it) break;
        const int k = it * SL + lane * EPL;
        if (k < K) {
          float wf[NC][EPL];
#pragma unroll
          for (int c = 0; c < NC; ++c) cvt_w16<WT>(ring[d][c], wf[c]);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float xv[EPL];
#pragma unroll
            for (int j4 = 0; j4 < EPL / 4; ++j4) {
              const float4 t = *(const float4*)(xr_s + m * K + k + 4 * j4);
              xv[4 * j4] = t.x; xv[4 * j4 + 1] = t.y; xv[4 * j4 + 2] = t.z; xv[4 * j4 + 3] = t.w;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int j = 0; j < EPL; ++j) acc[c][m] = fmaf(xv[j], wf[c][j], acc[c][m]);
          }
        }
        if (it + D < n_it) issue(it + D, ring[d]);
      }
    }
    gemv_finish<MT, NC>(a, acc, n0, lane);
  }
}

// ---------------------------------------------------------------------------------------------------- one input row (M = 1)
// The single-sequence decode step (CSM generate_frame: 16 backbone + 31 x 4 depth-decoder layer steps per frame, sesame.py:361-404) is a chain
// of ~650 of these per frame; profiles/r2_kernel_stats_csm_bygrid_call10.txt has the chunked kernel above at 5.8 us (q|k|v, 3 MB), 14.2 us
// (gate|up, 34-67 MB) and 6.7 us (o-proj / down) per launch -- bound by what happens around the weight stream: every 8-column workgroup
// re-stages x through LDS behind three barriers and recomputes the norm statistics, and a wave's life is two k-slices long.
//
// gemv1_res_kernel (K <= 4 slices, i.e. 2048 elements at 16 bits): a wave is autonomous.  Lane l multiplies the weights at k = it * SL + l * EPL
// with the x values at the same k -- ITS OWN 8 (16) floats of each slice -- so x lives in registers, the fused RMSNorm / LayerNorm statistics are
// one in-register pass plus a wave reduction, and there is no LDS and no barrier anywhere.  Waves walk column groups grid-stride with the next
// group's weights already in flight (double buffer), so the stream does not stop at group boundaries.
//
// gemv1_splitk_kernel (K > 2048: the down projections): 4 columns per workgroup, the 4 waves split K; a lane streams its slices of x with the
// weights (same prefetch ring), partial sums meet in LDS once.  16 KB of weights in flight per wave keeps HBM busy with only N / 4 workgroups.
// 16 bytes of the weight stream under the launch's cache policy (NT: global_load_dwordx4 ... nt)
template <bool NT>
__device__ __forceinline__ uint4 ldw16(const void* p) {
  if constexpr (NT) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    return *(const uint4*)p;
  }
}

// HALF: K <= 1024 (half the slices: half the weight ring and x registers -- the 1024-wide depth decoder / code predictor then keep 4 waves per SIMD
// resident instead of 2, see launch_gemv1)
template <int NC, int WT, bool NT = false, bool HALF = false>
__global__ __launch_bounds__(256) void gemv1_res_kernel(const mi355_gemv_args a, const int ngroups) {
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  constexpr int NIT = 2048 / (64 * 8) / (EPL / 8) / (HALF ? 2 : 1);  // slices covering K <= 2048 (4 at 16 bits, 2 for fp8 whose slice is 1024 elements)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K;
  const float* xrow = a.x_ids ? a.x + ((int64_t)a.x_ids[0] + a.x_id_offset) * a.ldx : a.x;   // optional fused embedding lookup
  float xr[NIT][EPL];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * SL + lane * EPL;
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) {
      const float4 t = k < K ? *(const float4*)(xrow + k + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[it][4 * j4] = t.x; xr[it][4 * j4 + 1] = t.y; xr[it][4 * j4 + 2] = t.z; xr[it][4 * j4 + 3] = t.w;
    }
  }
  uint4 ring[2][NIT][NC];
  auto issue = [&](int g, uint4 (&dst)[NIT][NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = g * NC + c < a.N ? g * NC + c : a.N - 1;  // clamp: tail columns recompute the last row, never stored
      const uint8_t* wrow = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = it * SL + lane * EPL;
        dst[it][c] = k < K ? ldw16<NT>(wrow + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  int g = blockIdx.x * 4 + wave;
  const int gstep = gridDim.x * 4;
  if (g < ngroups) issue(g, ring[0]);   // the weight stream starts before the statistics: its HBM latency overlaps the norm
  // the norm weights are requested before the statistics too (one L2 round trip less on the kernel's dependent chain)
  float4 nwq[NIT][EPL / 4];
  if (a.norm) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = it * SL + lane * EPL;
#pragma unroll
      for (int j4 = 0; j4 < EPL / 4; ++j4)
        nwq[it][j4] = (a.norm_weight && k < K) ? *(const float4*)(a.norm_weight + k + 4 * j4) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
  }
  if (a.norm) {
    float mean = 0.f;
    if (a.norm == 1) {
      float s = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < EPL; ++j) s += xr[it][j];
      mean = wave_sum_fast(s) / (float)K;
    }
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const float d = (it * SL + lane * EPL + j < K) ? xr[it][j] - mean : 0.f;
        q += d * d;
      }
    const float var = wave_sum_fast(q) / (float)K;
    const float rstd = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = it * SL + lane * EPL;
      if (k < K) {
#pragma unroll
        for (int j4 = 0; j4 < EPL / 4; ++j4) {
          const float4 w4 = nwq[it][j4];
          const float4 b4 = a.norm_bias ? *(const float4*)(a.norm_bias + k + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
          xr[it][4 * j4] = (xr[it][4 * j4] - mean) * rstd * w4.x + b4.x;
          xr[it][4 * j4 + 1] = (xr[it][4 * j4 + 1] - mean) * rstd * w4.y + b4.y;
          xr[it][4 * j4 + 2] = (xr[it][4 * j4 + 2] - mean) * rstd * w4.z + b4.z;
          xr[it][4 * j4 + 3] = (xr[it][4 * j4 + 3] - mean) * rstd * w4.w + b4.w;
        }
      }
    }
  }
  int buf = 0;
  for (; g < ngroups; g += gstep, buf ^= 1) {
    if (g + gstep < ngroups) {
      if (buf == 0) issue(g + gstep, ring[1]); else issue(g + gstep, ring[0]);
    }
    float acc[NC][1];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = 0.f;
    auto consume = [&](uint4 (&src)[NIT][NC]) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          float wf[EPL];
          cvt_w16<WT>(src[it][c], wf);
#pragma unroll
          for (int j = 0; j < EPL; ++j) acc[c][0] = fmaf(xr[it][j], wf[j], acc[c][0]);
        }
      }
    };
    if (buf == 0) consume(ring[0]); else consume(ring[1]);
    gemv_finish<1, NC>(a, acc, g * NC, lane);
  }
}


// gemv1_stream_kernel: gemv1_res_kernel's arithmetic with a memory schedule that has ONE dependent round trip.  The old loop's loads sat under
// `k < K` / `g + gstep < ngroups` branches; at every control-flow join the compiler's wait-count pass has to assume the path with the fewest
// younger loads, so each group's data was awaited with s_waitcnt vmcnt(0) -- the prefetched NEXT group included -- and every group then paid its
// own epilogue round trip (bias / residual loads -> store) in lane 0: ~3.5 us per 8 KB group and wave, whatever the bandwidth.  Here
//  * every load is unconditional (out-of-range k / groups are clamped to a valid address, x is zeroed instead): the loop body is straight-line
//    code and the waits are exact (vmcnt = the loads of the younger group);
//  * a wave owns a CONTIGUOUS run of <= 64 groups and parks group i's finished sums in lane i; after the last group the lanes run the
//    epilogue side by side: its loads and stores are coalesced over the run and are awaited once per wave, not once per group.
template <int NC, int WT, bool NT, bool HALF>
__global__ __launch_bounds__(256) void gemv1_stream_kernel(const mi355_gemv_args a, const int ngroups, const int gpw) {
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  constexpr int NIT = 2048 / (64 * 8) / (EPL / 8) / (HALF ? 2 : 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K;
  const int g0 = (blockIdx.x * 4 + wave) * gpw;
  if (g0 >= ngroups) return;                     // wave-uniform; the kernel has no barrier
  const int G = ngroups - g0 < gpw ? ngroups - g0 : gpw;
  const float* xrow = a.x_ids ? a.x + ((int64_t)a.x_ids[0] + a.x_id_offset) * a.ldx : a.x;
  int koff[NIT];          // element offset of this lane's piece of slice it (clamped into the row), and whether it is real
  bool kin[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * SL + lane * EPL;
    kin[it] = k < K;
    koff[it] = kin[it] ? k : 0;
  }
  float xr[NIT][EPL];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) {
      const float4 t = *(const float4*)(xrow + koff[it] + 4 * j4);
      xr[it][4 * j4] = t.x; xr[it][4 * j4 + 1] = t.y; xr[it][4 * j4 + 2] = t.z; xr[it][4 * j4 + 3] = t.w;
    }
  }
  float4 nwq[NIT][EPL / 4];
  // the norm weights are requested with x, UNCONDITIONALLY (a launch without them re-reads x instead and never uses the values): a load under a
  // branch whose other side supplies constants compiles to register copies inside the branch, i.e. an s_waitcnt before the weight stream starts
  const bool has_nw = a.norm && a.norm_weight;
  const float* nwp = has_nw ? a.norm_weight : xrow;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) nwq[it][j4] = *(const float4*)(nwp + koff[it] + 4 * j4);
  uint4 ring0[NIT][NC], ring1[NIT][NC];
  auto issue = [&](int gi, uint4 (&dst)[NIT][NC]) {
    const int g = g0 + (gi < G ? gi : G - 1);    // past the run: the last group again (an L1 / L2 hit), never a branch
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = g * NC + c < a.N ? g * NC + c : a.N - 1;
      const uint8_t* wrow = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
#pragma unroll
      for (int it = 0; it < NIT; ++it) dst[it][c] = ldw16<NT>(wrow + (int64_t)koff[it] * ESZ);
    }
  };
  issue(0, ring0);
  issue(1, ring1);   // unconditional (G == 1: group 0 again, merged with the request in flight): under `if (G > 1)` the loop's waits degrade to vmcnt(1 / 0)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    if (!kin[it]) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) xr[it][j] = 0.f;
    }
  if (a.norm) {
    float mean = 0.f;
    if (a.norm == 1) {
      float s = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < EPL; ++j) s += xr[it][j];
      mean = wave_sum_fast(s) / (float)K;
    }
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const float d = kin[it] ? xr[it][j] - mean : 0.f;
        q += d * d;
      }
    const float var = wave_sum_fast(q) / (float)K;
    const float rstd = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int j4 = 0; j4 < EPL / 4; ++j4) {
        const float4 w4 = has_nw ? nwq[it][j4] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        xr[it][4 * j4] = kin[it] ? (xr[it][4 * j4] - mean) * rstd * w4.x + b4.x : 0.f;
        xr[it][4 * j4 + 1] = kin[it] ? (xr[it][4 * j4 + 1] - mean) * rstd * w4.y + b4.y : 0.f;
        xr[it][4 * j4 + 2] = kin[it] ? (xr[it][4 * j4 + 2] - mean) * rstd * w4.z + b4.z : 0.f;
        xr[it][4 * j4 + 3] = kin[it] ? (xr[it][4 * j4 + 3] - mean) * rstd * w4.w + b4.w : 0.f;
      }
    }
    if (a.norm_bias) {   // a LayerNorm's bias comes late, under one branch (loads behind the stream's first two groups: that stack's first wait is vmcnt(0))
      float4 nb[NIT][EPL / 4];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j4 = 0; j4 < EPL / 4; ++j4) nb[it][j4] = *(const float4*)(a.norm_bias + koff[it] + 4 * j4);
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j4 = 0; j4 < EPL / 4; ++j4) {
          xr[it][4 * j4] += kin[it] ? nb[it][j4].x : 0.f;
          xr[it][4 * j4 + 1] += kin[it] ? nb[it][j4].y : 0.f;
          xr[it][4 * j4 + 2] += kin[it] ? nb[it][j4].z : 0.f;
          xr[it][4 * j4 + 3] += kin[it] ? nb[it][j4].w : 0.f;
        }
    }
  }
  float mine[NC][1];      // lane i: the sums of group g0 + i
#pragma unroll
  for (int c = 0; c < NC; ++c) mine[c][0] = 0.f;
  auto consume = [&](uint4 (&src)[NIT][NC], int gi) {
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float wf[EPL];
        cvt_w16<WT>(src[it][c], wf);
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[c] = fmaf(xr[it][j], wf[j], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float t = wave_sum_fast(acc[c]);
      mine[c][0] = lane == gi ? t : mine[c][0];
    }
  };
  // the steady state requests two groups ahead; the last pair is consumed without requesting anything (a repeat of the last group there would be
  // a second trip to HBM for a non-temporal stream).  Only an odd run of >= 3 groups re-reads one group (its last), once.
  int gi = 0;
  for (; gi + 2 < G; gi += 2) {
    consume(ring0, gi);
    issue(gi + 2, ring0);
    __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler sinks these requests behind the second group's arithmetic: nothing would overlap)
    consume(ring1, gi + 1);
    issue(gi + 3, ring1);
    __builtin_amdgcn_sched_barrier(0);
  }
  consume(ring0, gi);
  if (gi + 1 < G) consume(ring1, gi + 1);
  if (lane < G) gemv_epilogue<1, NC>(a, mine, (g0 + lane) * NC);
}

// gemv1_attn_kernel: the o-proj of a decode step with the attention of that step as its PROLOGUE (M == 1, K = heads * dh <= 2048, at most 64 cached
// positions: the depth decoder of CSM attends over <= 32 positions, 124 times per frame).  Every workgroup recomputes the whole attention row -- a
// few thousand multiply-adds on K | V rows that sit in L2 -- while its slice of the weight stream is already in flight, and then runs the
// register-resident GEMV of gemv1_res_kernel on it: one launch per layer less, and the scores never touch memory.  x = the (rotated) query row.
template <int WT, bool NT>
__global__ __launch_bounds__(256) void gemv1_attn_kernel(const mi355_gemv_args a, const int ngroups) {
  constexpr int NC = 1;
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  constexpr int NIT = 2048 / (64 * 8) / (EPL / 8);
  __shared__ float s_q[2048];        // query, then (same storage) the attention output
  __shared__ float s_p[32 * 64];     // scores / probabilities [heads][64]
  __shared__ float s_o[4 * 2048];    // partial outputs of the position shares
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, H = a.attn_heads, G = a.attn_kv_heads, dh = a.attn_dh, Tk = a.attn_Tk, rep = H / G;
  uint4 ring[2][NIT][NC];
  auto issue = [&](int g, uint4 (&dst)[NIT][NC]) {
    const int n = g < a.N ? g : a.N - 1;
    const uint8_t* wrow = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = it * SL + lane * EPL;
      dst[it][0] = k < K ? ldw16<NT>(wrow + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  int g = blockIdx.x * 4 + wave;
  const int gstep = gridDim.x * 4;
  if (g < ngroups) issue(g, ring[0]);   // the weight stream starts first: its HBM latency covers the attention prologue
  // four consecutive cache elements as floats (16-byte / 8-byte load by element type)
  auto kv4 = [&](const void* base, const int64_t idx) -> float4 {
    if (a.attn_kv_dtype == MI355_KV_F32) return *(const float4*)((const float*)base + idx);
    const uint2 u = *(const uint2*)((const uint16_t*)base + idx);
    if (a.attn_kv_dtype == MI355_KV_BF16)
      return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u), __builtin_bit_cast(float, u.y << 16),
                         __builtin_bit_cast(float, u.y & 0xffff0000u));
    return make_float4((float)__builtin_bit_cast(_Float16, (uint16_t)(u.x & 0xffffu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(u.x >> 16)),
                       (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y & 0xffffu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(u.y >> 16)));
  };
  for (int i = tid; i < K; i += 256) s_q[i] = a.x[i];
  __syncthreads();
  // scores: one cache row (t, g) per quad of lanes, each lane a quarter of the dh channels; the quad's partial dots with the rep query heads of kv
  // head g meet through two xor steps.  All of a pass's loads are in flight together: one L2 round trip per 64 rows.
  {
    const int qd = tid & 3, rowi = tid >> 2, seg = dh >> 2;   // seg = channels per lane (16 or 32)
    for (int r0 = 0; r0 < Tk * G; r0 += 64) {
      const int r = r0 + rowi;
      const bool ok = r < Tk * G;
      const int t = ok ? r / G : 0, gk = ok ? r - t * G : 0;
      const int64_t kb = (int64_t)t * a.attn_ld + gk * dh + qd * seg;
      float4 kk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (4 * j < seg) kk[j] = kv4(a.attn_k, kb + 4 * j);
      for (int rr = 0; rr < rep; ++rr) {
        const int h = gk * rep + rr;
        const float* qp = s_q + h * dh + qd * seg;
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (4 * j < seg) {
            sc = fmaf(qp[4 * j], kk[j].x, sc); sc = fmaf(qp[4 * j + 1], kk[j].y, sc);
            sc = fmaf(qp[4 * j + 2], kk[j].z, sc); sc = fmaf(qp[4 * j + 3], kk[j].w, sc);
          }
        sc += __shfl_xor(sc, 1, 64);
        sc += __shfl_xor(sc, 2, 64);
        if (ok && qd == 0) s_p[h * 64 + t] = sc * a.attn_scale;
      }
    }
  }
  __syncthreads();
  // softmax per head: one wave per head round, lane = key position (Tk <= 64)
  for (int h = wave; h < H; h += 4) {
    const float v = lane < Tk ? s_p[h * 64 + lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < Tk ? expf(v - m) : 0.f;
    const float den = wave_sum(e);
    if (lane < Tk) s_p[h * 64 + lane] = e / den;
  }
  __syncthreads();
  // out[h, d] = sum_t p[h, t] v[t, h / rep, d]: a thread owns four channels of one kv head for a share of the positions (TS shares so that all 256
  // threads work); the shares meet in s_o, the result overwrites the query
  {
    const int items = G * (dh >> 2);              // (g, d4) items: 64 (depth decoder) .. 128
    const int TS = 256 / items >= 4 ? 4 : (256 / items >= 1 ? 256 / items : 1);   // <= 4 shares: s_o holds four rows
    const int it = tid % items, ts = tid / items;
    const int gk = it / (dh >> 2), d4 = (it - gk * (dh >> 2)) * 4;
    float4 o[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) o[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ts < TS) {
      for (int t = ts; t < Tk; t += TS) {
        const float4 vv = kv4(a.attn_v, (int64_t)t * a.attn_ld + gk * dh + d4);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
          if (rr < rep) {
            const float pw = s_p[(gk * rep + rr) * 64 + t];
            o[rr].x = fmaf(pw, vv.x, o[rr].x); o[rr].y = fmaf(pw, vv.y, o[rr].y); o[rr].z = fmaf(pw, vv.z, o[rr].z); o[rr].w = fmaf(pw, vv.w, o[rr].w);
          }
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        if (rr < rep) *(float4*)(s_o + ts * 2048 + (gk * rep + rr) * dh + d4) = o[rr];
    }
    __syncthreads();
    for (int i = tid; i < K; i += 256) {
      float sum = s_o[i];
      for (int u = 1; u < TS; ++u) sum += s_o[u * 2048 + i];
      s_q[i] = sum;
    }
  }
  __syncthreads();
  float xr[NIT][EPL];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = it * SL + lane * EPL;
#pragma unroll
    for (int j = 0; j < EPL; ++j) xr[it][j] = k + j < K ? s_q[k + j] : 0.f;
  }
  int buf = 0;
  for (; g < ngroups; g += gstep, buf ^= 1) {
    if (g + gstep < ngroups) {
      if (buf == 0) issue(g + gstep, ring[1]); else issue(g + gstep, ring[0]);
    }
    float acc[NC][1];
    acc[0][0] = 0.f;
    auto consume = [&](uint4 (&src)[NIT][NC]) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        float wf[EPL];
        cvt_w16<WT>(src[it][0], wf);
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[0][0] = fmaf(xr[it][j], wf[j], acc[0][0]);
      }
    };
    if (buf == 0) consume(ring[0]); else consume(ring[1]);
    gemv_finish<1, NC>(a, acc, g, lane);
  }
}

template <int WT, bool NT = false>
__global__ __launch_bounds__(256) void gemv1_splitk_old_kernel(const mi355_gemv_args a) {
  constexpr int NC = 4, D = 4;
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  __shared__ float part[4][NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * NC;
  const int n_it = (a.K + SL - 1) / SL;
  const int per = (n_it + 3) / 4;                       // slices per wave
  const int it0 = wave * per, it1 = it0 + per < n_it ? it0 + per : n_it;
  const uint8_t* wrow[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int n = n0 + c < a.N ? n0 + c : a.N - 1;
    wrow[c] = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
  }
  const float* xrow = a.x_ids ? a.x + ((int64_t)a.x_ids[0] + a.x_id_offset) * a.ldx : a.x;
  uint4 wring[D][NC];
  float4 xring[D][EPL / 4];
  auto issue = [&](int it, int d) {
    const int k = it * SL + lane * EPL;
#pragma unroll
    for (int c = 0; c < NC; ++c) wring[d][c] = k < a.K ? ldw16<NT>(wrow[c] + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) xring[d][j4] = k < a.K ? *(const float4*)(xrow + k + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (it0 + d < it1) issue(it0 + d, d);
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  for (int base = it0; base < it1; base += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int it = base + d;
      if (it >= it1) break;
      float xv[EPL];
#pragma unroll
      for (int j4 = 0; j4 < EPL / 4; ++j4) {
        xv[4 * j4] = xring[d][j4].x; xv[4 * j4 + 1] = xring[d][j4].y; xv[4 * j4 + 2] = xring[d][j4].z; xv[4 * j4 + 3] = xring[d][j4].w;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float wf[EPL];
        cvt_w16<WT>(wring[d][c], wf);
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[c] = fmaf(xv[j], wf[j], acc[c]);
      }
      if (it + D < it1) issue(it + D, d);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = wave_sum_fast(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) part[wave][c] = acc[c];
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < NC && n0 + c < a.N) {
    const int n = n0 + c;
    const float sum = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];   // fixed order: run-to-run identical
    const float ws = a.wscale ? a.wscale[n] * kFp8Unbias : 1.f;
    float v = gemv_act(sum * ws + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
    if (a.res) v += a.res[n];
    if (a.y2 && n >= a.split) store_kv_elem(a.y2, n - a.split, v * a.out_scale, a.y2_dtype);
    else a.y[n] = v * a.out_scale;
  }
}

// gemv1_splitk_kernel, second form: the first one's `k < K ? load : 0` guards compiled to a branch per load, several of them with their own
// s_waitcnt vmcnt(0) -- up to eight SERIAL HBM round trips before the first multiply (7.0-8.7 us for the 16.8 MB down projection of the CSM depth
// decoder, whose bytes are 2.7 us at the chip's rate).  Here every load is unconditional (clamped address, x zeroed instead), the epilogue's
// operands are requested first, and the slice loop is straight-line code: one round trip.
template <int WT, bool NT = false>
__global__ __launch_bounds__(256) void gemv1_splitk_kernel(const mi355_gemv_args a) {
  constexpr int NC = 4, D = 4;
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;
  __shared__ float part[4][NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * NC;
  const int ec = threadIdx.x < NC && n0 + (int)threadIdx.x < a.N ? n0 + (int)threadIdx.x : -1;   // the column this thread finishes
  // its operands are requested first, by every thread for a clamped column (no lane-divergent branch in front of the stream)
  const int el = n0 + (lane & 3) < a.N ? n0 + (lane & 3) : a.N - 1;
  const float e_bias = a.bias ? a.bias[el] : 0.f;
  const float e_cs = a.colscale ? a.colscale[el] : 1.f;
  const float e_res = a.res ? a.res[el] : 0.f;
  const float e_ws = a.wscale ? a.wscale[el] : 1.0f / kFp8Unbias;
  const int n_it = (a.K + SL - 1) / SL;
  const int per = (n_it + 3) / 4;                       // slices per wave
  const int it0 = wave * per, it1 = it0 + per < n_it ? it0 + per : n_it;
  const uint8_t* wrow[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int n = n0 + c < a.N ? n0 + c : a.N - 1;
    wrow[c] = (const uint8_t*)a.w + (int64_t)n * a.ldw * ESZ;
  }
  const float* xrow = a.x_ids ? a.x + ((int64_t)a.x_ids[0] + a.x_id_offset) * a.ldx : a.x;
  uint4 wring[D][NC];
  float4 xring[D][EPL / 4];
  bool okring[D];
  auto issue = [&](int it, int d) {
    const int k = it * SL + lane * EPL;
    okring[d] = it < it1 && k < a.K;
    const int kc = okring[d] ? k : 0;                   // past the wave's slices / the row: a valid address, the product is zeroed through x
#pragma unroll
    for (int c = 0; c < NC; ++c) wring[d][c] = ldw16<NT>(wrow[c] + (int64_t)kc * ESZ);
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) xring[d][j4] = *(const float4*)(xrow + kc + 4 * j4);
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(it0 + d, d);
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.f;
  auto consume = [&](int d) {
    float xv[EPL];
#pragma unroll
    for (int j4 = 0; j4 < EPL / 4; ++j4) {
      xv[4 * j4] = okring[d] ? xring[d][j4].x : 0.f; xv[4 * j4 + 1] = okring[d] ? xring[d][j4].y : 0.f;
      xv[4 * j4 + 2] = okring[d] ? xring[d][j4].z : 0.f; xv[4 * j4 + 3] = okring[d] ? xring[d][j4].w : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float wf[EPL];
      cvt_w16<WT>(wring[d][c], wf);
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[c] = fmaf(xv[j], wf[j], acc[c]);
    }
  };
  int base = it0;
  for (; base + D < it1; base += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(d);
      issue(base + D + d, d);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) consume(d);               // the last block: nothing left to request
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = wave_sum_fast(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) part[wave][c] = acc[c];
  }
  __syncthreads();
  if (ec >= 0) {
    const int c = threadIdx.x;
    const float sum = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];   // fixed order: run-to-run identical
    float ews = e_ws;
    asm volatile("" : "+v"(ews));   // keeps the scale's multiply (and with it the load's s_waitcnt) down here instead of hoisted in front of the stream
    const float ws = a.wscale ? ews * kFp8Unbias : 1.f;
    float v = gemv_act(sum * ws + e_bias, a.post_act, a.post_slope) * e_cs;
    if (a.res) v += e_res;
    if (a.y2 && ec >= a.split) store_kv_elem(a.y2, ec - a.split, v * a.out_scale, a.y2_dtype);
    else a.y[ec] = v * a.out_scale;
  }
}

template <int WT>
int launch_gemv1(const mi355_gemv_args& a, hipStream_t st) {
  constexpr int EPL = mi355_wt<WT>::EPL;
  const int kres = 2048 * (EPL / 8) / (EPL / 8);  // K the register-resident kernel covers (2048 elements for every weight type)
  MI355_CLEAR_ERROR();
  if (a.attn_k) {   // attention prologue: validated by mi355_gemv
    int blocks = (a.N + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    static const int nt_env0 = getenv("MI355_GEMV_NT") ? atoi(getenv("MI355_GEMV_NT")) : -1;
    if (nt_env0 >= 0 ? nt_env0 != 0 : a.w_policy == 1) hipLaunchKernelGGL((gemv1_attn_kernel<WT, true>), dim3(blocks), dim3(256), 0, st, a, a.N);
    else hipLaunchKernelGGL((gemv1_attn_kernel<WT, false>), dim3(blocks), dim3(256), 0, st, a, a.N);
    MI355_LAUNCH_CHECK("gemv(M=1, attention prologue)");
    return MI355_OK;
  }
  if (a.K <= kres) {
    const bool two = a.glu || a.rope_cos || a.N >= 4096;
    const int ngroups = two ? (a.N + 1) / 2 : a.N;
    static const int nt_env = getenv("MI355_GEMV_NT") ? atoi(getenv("MI355_GEMV_NT")) : -1;   // A/B knob: 0 / 1 overrides every launch's policy
    const bool nt = nt_env >= 0 ? nt_env != 0 : a.w_policy == 1;
    const bool half = a.K <= 1024;
    // ONE resident round: the grid is what the chip holds at this instantiation's register count (the two-column kernel at K = 2048 takes 224
    // registers = 2 workgroups per CU; launched as 1024 workgroups it ran two full rounds, each with its own x load -> statistics -> first HBM
    // round trip: 13.9 us for the 33.5 MB gate|up image of the depth decoder, profiles/r2_kernel_stats_csm_bygrid_call11_m1_kernels.txt); the rest
    // of the columns come grid-stride with their weights prefetched
    auto go = [&](auto kern) {
      const int resident = resident_workgroups((const void*)kern);   // per kernel ADDRESS and device (see resident_workgroups)
      int blocks = (ngroups + 3) / 4;
      if (blocks > resident) blocks = resident;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, a, ngroups);
    };
    static const bool old1 = getenv("MI355_GEMV1_OLD") != nullptr && getenv("MI355_GEMV1_OLD")[0] == '1';   // A/B knob: the grid-stride kernel
    auto gs = [&](auto kern) {
      const int resident = resident_workgroups((const void*)kern);
      // one resident round of waves, every wave the same contiguous run of groups (<= 64: one lane per group for the epilogue)
      int gpw = (ngroups + resident * 4 - 1) / (resident * 4);
      if (gpw > 64) gpw = 64;
      const int blocks = (ngroups + 4 * gpw - 1) / (4 * gpw);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, a, ngroups, gpw);
    };
    if (!old1) {
      if (half) {
        if (two && nt) gs(gemv1_stream_kernel<2, WT, true, true>);
        else if (two) gs(gemv1_stream_kernel<2, WT, false, true>);
        else if (nt) gs(gemv1_stream_kernel<1, WT, true, true>);
        else gs(gemv1_stream_kernel<1, WT, false, true>);
      } else {
        if (two && nt) gs(gemv1_stream_kernel<2, WT, true, false>);
        else if (two) gs(gemv1_stream_kernel<2, WT, false, false>);
        else if (nt) gs(gemv1_stream_kernel<1, WT, true, false>);
        else gs(gemv1_stream_kernel<1, WT, false, false>);
      }
      MI355_LAUNCH_CHECK("gemv(M=1, register-resident x, streamed groups)");
      return MI355_OK;
    }
    if (half) {
      if (two && nt) go(gemv1_res_kernel<2, WT, true, true>);
      else if (two) go(gemv1_res_kernel<2, WT, false, true>);
      else if (nt) go(gemv1_res_kernel<1, WT, true, true>);
      else go(gemv1_res_kernel<1, WT, false, true>);
    } else {
      if (two && nt) go(gemv1_res_kernel<2, WT, true>);
      else if (two) go(gemv1_res_kernel<2, WT>);
      else if (nt) go(gemv1_res_kernel<1, WT, true>);
      else go(gemv1_res_kernel<1, WT>);
    }
    MI355_LAUNCH_CHECK("gemv(M=1, register-resident x)");
    return MI355_OK;
  }
  static const int nt_env2 = getenv("MI355_GEMV_NT") ? atoi(getenv("MI355_GEMV_NT")) : -1;
  static const bool old2 = getenv("MI355_GEMV1_OLD") != nullptr && getenv("MI355_GEMV1_OLD")[0] == '1';
  const bool nt2 = nt_env2 >= 0 ? nt_env2 != 0 : a.w_policy == 1;
  if (old2 && nt2) hipLaunchKernelGGL((gemv1_splitk_old_kernel<WT, true>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
  else if (old2) hipLaunchKernelGGL((gemv1_splitk_old_kernel<WT>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
  else if (nt2) hipLaunchKernelGGL((gemv1_splitk_kernel<WT, true>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemv1_splitk_kernel<WT>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("gemv(M=1, split K)");
  return MI355_OK;
}

template <int MT, int WT>
int launch_gemv(const mi355_gemv_args& a, hipStream_t st) {
  // one column per wave keeps the most wavefronts (and weight bytes) in flight; two columns per wave halve the LDS reads of x per weight
  // byte, which is what binds at 8 rows (16 LDS bytes per weight byte at NC = 1) and for very wide outputs
  static const int nc_env = getenv("MI355_GEMV_NC") ? atoi(getenv("MI355_GEMV_NC")) : 0;  // A/B knob (1 | 2); SwiGLU pairs always need 2
  const int nc = (a.glu || a.rope_cos) ? 2 : (nc_env == 1 || nc_env == 2) ? nc_env : ((MT >= 8 || a.N >= 16384) ? 2 : 1);
  if constexpr (MT >= 4) {
    const size_t lds = (size_t)MT * a.K * sizeof(float);
    // Measured (profiles/r1_kernel_stats_qwen3_1p7b_b8_v4_resident.txt): 22.0 us vs 18.9 us per launch for the chunked kernel at M = 8 -- these
    // launches are bound by their dependent chain (statistics -> staging -> first weight slice), not by the re-staging of x, and the chunked
    // kernel overlaps the weight latency with the staging.  Kept as an opt-in (MI355_GEMV_RESIDENT=1) for wide-N shapes.
    static const bool use_res = getenv("MI355_GEMV_RESIDENT") != nullptr;
    if (lds <= 96 * 1024 && use_res) {
      static bool attr_set = false;  // benign race: the attribute is idempotent
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemv_res_kernel<MT, 2, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        MI355_REQUIRE(e == hipSuccess, "gemv: cannot reserve LDS: %s", hipGetErrorString(e));
        attr_set = true;
      }
      const int ngroups = (a.N + 1) / 2;
      const int per_cu = lds <= 24 * 1024 ? 4 : (lds <= 48 * 1024 ? 3 : (lds <= 72 * 1024 ? 2 : 1));
      int blocks = (ngroups + 3) / 4;
      if (blocks > 256 * per_cu) blocks = 256 * per_cu;
      MI355_CLEAR_ERROR();
      hipLaunchKernelGGL((gemv_res_kernel<MT, 2, WT>), dim3(blocks), dim3(256), lds, st, a, ngroups);
      MI355_LAUNCH_CHECK("gemv(resident)");
      return MI355_OK;
    }
  }
  MI355_CLEAR_ERROR();
  if (nc == 2) hipLaunchKernelGGL((gemv_kernel<MT, 2, WT>), dim3((a.N + 7) / 8), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemv_kernel<MT, 1, WT>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("gemv");
  return MI355_OK;
}

template <int WT>
int launch_gemv_m(const mi355_gemv_args& a, hipStream_t st) {
  static const bool m1_old = getenv("MI355_GEMV_M1") != nullptr && getenv("MI355_GEMV_M1")[0] == '0';  // A/B knob: the chunked kernel at one row
  // split K needs a plain epilogue (no SwiGLU pairs / rotary pairs: those shapes have K <= 2048 in every model of the path) and no fused norm
  if (a.M == 1 && (!m1_old || a.x_ids) && (a.K <= 2048 || (!a.norm && !a.glu && !a.rope_cos))) return launch_gemv1<WT>(a, st);
  if (a.M == 1) return launch_gemv<1, WT>(a, st);
  if (a.M == 2) return launch_gemv<2, WT>(a, st);
  if (a.M <= 4) return launch_gemv<4, WT>(a, st);
  return launch_gemv<8, WT>(a, st);
}

}  // namespace

// gemv_mfma.hip / gemv_mfma_fp8.hip: the matrix-pipe kernels for 5..8 rows (16-bit and fp8 weight images)
int mi355_gemv_mfma_eligible(const mi355_gemv_args& a);
int mi355_gemv_mfma_launch(const mi355_gemv_args& a, hipStream_t st);
int mi355_gemv_mfma_fp8_eligible(const mi355_gemv_args& a);
int mi355_gemv_mfma_fp8_launch(const mi355_gemv_args& a, hipStream_t st);
// gemm_rows.hip: 9..64 rows (a batch of sequences per decode step), 16-bit weight images
int mi355_gemm_rows_eligible(const mi355_gemv_args& a);
int mi355_gemm_rows_launch(const mi355_gemv_args& a, hipStream_t st);

extern "C" int mi355_gemv(const mi355_gemv_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->w && ap->y, "gemv: null tensor");
  mi355_gemv_args a = *ap;
  MI355_REQUIRE(a.M >= 1 && a.M <= 64, "gemv: M must be in [1, 64] (got %d); use conv_gemm for taller inputs", a.M);
  MI355_REQUIRE(a.M <= 8 || mi355_gemm_rows_eligible(a),
                "gemv: 9..64 rows need 16-bit weights, K %% 64 == 0, 16-byte aligned rows and no fused rope / gathered input (M = %d, K = %d, wdtype = %d)",
                a.M, a.K, a.wdtype);
  MI355_REQUIRE(!a.x_ids || (a.M == 1 && (a.K <= 2048 || (!a.norm && !a.glu && !a.rope_cos))), "gemv: the gathered input exists for one row on the M = 1 kernels");
  MI355_REQUIRE(a.N > 0 && a.K > 0 && a.K % 8 == 0, "gemv: K must be a positive multiple of 8");
  MI355_REQUIRE(a.ldw % 8 == 0 && a.ldw >= a.K && ((uintptr_t)a.w) % 16 == 0, "gemv: weight rows must be 16-byte aligned");
  MI355_REQUIRE(a.ldx % 4 == 0 && ((uintptr_t)a.x) % 16 == 0, "gemv: x rows must be 16-byte aligned");
  MI355_REQUIRE(a.wdtype == MI355_W_BF16 || a.wdtype == MI355_W_F16 || a.wdtype == MI355_W_FP8, "gemv: wdtype must be MI355_W_BF16, MI355_W_F16 or MI355_W_FP8");
  MI355_REQUIRE(a.wdtype != MI355_W_FP8 || (a.wscale && a.K % 16 == 0 && a.ldw % 16 == 0), "gemv: fp8 weights need wscale and K, ldw multiples of 16");
  MI355_REQUIRE(a.wdtype == MI355_W_FP8 || !a.wscale, "gemv: wscale belongs to fp8 weights only");
  MI355_REQUIRE(!a.glu || (a.N % 2 == 0 && !a.res && !a.colscale && a.post_act == MI355_ACT_NONE && !a.y2), "gemv: glu needs an even N and a plain epilogue");
  MI355_REQUIRE(a.norm >= 0 && a.norm <= 2, "gemv: norm must be 0 (none), 1 (LayerNorm) or 2 (RMSNorm)");
  MI355_REQUIRE(!a.norm || (a.K % 4 == 0 && (!a.norm_weight || ((uintptr_t)a.norm_weight) % 16 == 0) && (!a.norm_bias || ((uintptr_t)a.norm_bias) % 16 == 0)),
                "gemv: norm weight / bias must be 16-byte aligned");
  MI355_REQUIRE(!a.y2 || (a.split > 0 && a.split < a.N), "gemv: split must be inside (0, N) when y2 is given");
  MI355_REQUIRE(a.y2_dtype >= MI355_KV_F32 && a.y2_dtype <= MI355_KV_F16, "gemv: y2_dtype must be MI355_KV_F32, MI355_KV_BF16 or MI355_KV_F16");
  MI355_REQUIRE(!a.rope_cos || (a.rope_sin && a.rope_dh > 0 && a.rope_dh % 2 == 0 && a.rope_cols > 0 && a.rope_cols <= a.N && a.rope_cols % a.rope_dh == 0 &&
                                !a.glu && !a.res && !a.colscale && a.post_act == MI355_ACT_NONE && a.M <= 4 && (!a.y2 || a.split % 2 == 0)),
                "gemv: fused rope needs a plain epilogue, <= 4 rows, whole heads and an even split");
  MI355_REQUIRE(!a.attn_k || (a.attn_v && a.M == 1 && a.K <= 2048 && !a.norm && !a.glu && !a.rope_cos && !a.x_ids && a.attn_Tk >= 1 && a.attn_Tk <= 64 &&
                               a.attn_heads >= 1 && a.attn_heads <= 32 && a.attn_kv_heads >= 1 && a.attn_heads % a.attn_kv_heads == 0 &&
                               a.attn_heads * a.attn_dh == a.K && a.attn_ld >= a.attn_kv_heads * a.attn_dh && a.attn_ld % 4 == 0 &&
                               (a.attn_dh == 64 || a.attn_dh == 128) && a.attn_heads / a.attn_kv_heads <= 8 && a.attn_kv_heads * (a.attn_dh / 4) <= 256 &&
                               ((uintptr_t)a.attn_k) % 16 == 0 && ((uintptr_t)a.attn_v) % 8 == 0 &&
                               a.attn_kv_dtype >= MI355_KV_F32 && a.attn_kv_dtype <= MI355_KV_F16),
                "gemv: the attention prologue needs one row, K = heads * dh <= 2048, <= 32 heads, 1..64 cached positions and a plain input side");
  MI355_REQUIRE(a.w_policy == 0 || a.w_policy == 1, "gemv: w_policy must be 0 (default) or 1 (non-temporal)");
  if (a.out_scale == 0.f) a.out_scale = 1.f;
  static const bool two_reads = getenv("MI355_GEMV_TWO_READS") != nullptr;  // A/B knob: statistics from a separate read of x (the older schedule)
  a.norm_two_reads = two_reads ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (a.M > 8) return mi355_gemm_rows_launch(a, st);
  if (mi355_gemv_mfma_eligible(a)) return mi355_gemv_mfma_launch(a, st);
  if (mi355_gemv_mfma_fp8_eligible(a)) return mi355_gemv_mfma_fp8_launch(a, st);
  if (a.wdtype == MI355_W_FP8) return launch_gemv_m<MI355_W_FP8>(a, st);
  return a.wdtype == MI355_W_F16 ? launch_gemv_m<MI355_W_F16>(a, st) : launch_gemv_m<MI355_W_BF16>(a, st);
}

// fp32 [rows, cols] (host) -> row-major 16-bit image for mi355_gemv (and for embedding tables kept in the checkpoint dtype)
extern "C" int mi355_pack_rowmajor16_host(const float* w, int64_t n, int32_t dtype, uint16_t* out) {
  MI355_REQUIRE(w && out && n >= 0, "pack_rowmajor16: bad arguments");
  MI355_REQUIRE(dtype == MI355_W_BF16 || dtype == MI355_W_F16, "pack_rowmajor16: dtype must be MI355_W_BF16 or MI355_W_F16");
  for (int64_t i = 0; i < n; ++i) out[i] = dtype == MI355_W_F16 ? host_f32_to_f16(w[i]) : host_f32_to_bf16(w[i]);
  return MI355_OK;
}

// fp32 [rows, cols] (host) -> fp8 e4m3fn bytes + one power-of-two scale per row (see the header)
extern "C" int mi355_pack_rowmajor_fp8_host(const float* w, int64_t rows, int64_t cols, uint8_t* out, float* scale) {
  MI355_REQUIRE(w && out && scale && rows >= 0 && cols > 0 && cols % 16 == 0, "pack_rowmajor_fp8: bad arguments (cols must be a multiple of 16)");
  for (int64_t r = 0; r < rows; ++r) {
    const float* wr = w + r * cols;
    float amax = 0.f;
    for (int64_t c = 0; c < cols; ++c) {
      const float a = __builtin_fabsf(wr[c]);
      MI355_REQUIRE(a == a && a <= 3.0e38f, "pack_rowmajor_fp8: non-finite weight in row %lld", (long long)r);
      if (a > amax) amax = a;
    }
    float s = 1.0f;
    if (amax > 0.f) {
      int e;
      const float m = __builtin_frexpf(amax / 448.0f, &e);  // amax / 448 = m * 2^e, m in [0.5, 1)
      s = __builtin_ldexpf(1.0f, m == 0.5f ? e - 1 : e);    // 2^ceil(log2(amax / 448))
    }
    scale[r] = s;
    for (int64_t c = 0; c < cols; ++c) out[r * cols + c] = host_f32_to_e4m3(wr[c] / s);  // power-of-two divide: exact
  }
  return MI355_OK;
}
