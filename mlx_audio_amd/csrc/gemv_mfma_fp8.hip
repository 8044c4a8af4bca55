// Skinny GEMM for 5..8 rows per decode step with fp8 weight images on the fp8 matrix pipe (gfx950): y[m, n] = epilogue(sum_k norm(x)[m, k] * W[n, k]).
//
// BASELINE.json config[4] asks for fp8 GEMMs on CSM; mi355_pack_rowmajor_fp8_host gives every Linear an OCP e4m3fn byte image with one power-of-two
// scale per output row.  The FMA kernel (gemv.hip) decodes those bytes to fp32 and spends 8 FMAs per weight element at 8 rows; here the bytes go
// to v_mfma_f32_16x16x32_fp8_fp8 untouched:
//   * A operand = a 16-row tile of W straight from HBM: lane (i = lane & 15, g = lane >> 4) loads the 16 contiguous bytes W[n0 + i][k0 + 16 g .. + 16)
//     of a 64-wide k step and feeds its two 8-byte halves to two MFMAs (the contraction index is a free relabelling as long as A and B agree);
//   * B operand = the (normalised) input rows.  fp8 has 4 significant bits, so a row is scaled by a power of two into [-240, 240] and split into
//     NT = 4 e4m3 TERMS of decreasing weight: x / s = h0 + h1 / 16 + h2 / 256 + h3 / 4096 (each remainder is exact in fp32 and is re-scaled by 16 before
//     it is rounded again), ~16 significant bits in total -- the accuracy of the bf16 hi + lo split the 16-bit kernel uses.  Each term has its own
//     accumulator (the terms differ by a power of two that cannot ride inside an e4m3 operand); they are combined per 2048-column chunk together with
//     the chunk's row scale of x, and the row scale of W is applied in the epilogue.  The images live in LDS in fragment order [term][k step][half][group][row] x 8 bytes;
//   * split-K over the four waves of a workgroup, fused LayerNorm / RMSNorm, bias / activation / LayerScale / residual / SwiGLU / split destinations
//     exactly as gemv_mfma.hip.
// Same contract as the 16-bit matrix-pipe kernel (5..8 rows, K % 64 == 0; a fused norm needs K <= 2048); MI355_GEMV_MFMA_FP8=0 keeps the FMA kernel.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kKC8 = 2048;  // input columns staged: 4 terms x 8 rows x 1 byte x kKC8 = 64 KB of LDS
constexpr int kD8 = 8;      // weight prefetch depth in k steps (16 bytes per lane and step)
constexpr int NT = 4;       // e4m3 terms of the input rows

__device__ __forceinline__ float fp8_act(float v, int act, float slope) {
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

// 8 fp32 values -> NT e4m3 terms (8 bytes each); the values are already scaled into the e4m3 range
__device__ __forceinline__ void split_fp8(float (&v)[8], uint2 (&out)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    out[t] = make_uint2((uint32_t)lo, (uint32_t)hi);
    if (t + 1 < NT) {
      v[0] = (v[0] - __builtin_amdgcn_cvt_f32_fp8(lo, 0)) * 16.f; v[1] = (v[1] - __builtin_amdgcn_cvt_f32_fp8(lo, 1)) * 16.f;
      v[2] = (v[2] - __builtin_amdgcn_cvt_f32_fp8(lo, 2)) * 16.f; v[3] = (v[3] - __builtin_amdgcn_cvt_f32_fp8(lo, 3)) * 16.f;
      v[4] = (v[4] - __builtin_amdgcn_cvt_f32_fp8(hi, 0)) * 16.f; v[5] = (v[5] - __builtin_amdgcn_cvt_f32_fp8(hi, 1)) * 16.f;
      v[6] = (v[6] - __builtin_amdgcn_cvt_f32_fp8(hi, 2)) * 16.f; v[7] = (v[7] - __builtin_amdgcn_cvt_f32_fp8(hi, 3)) * 16.f;
    }
  }
}

__device__ __forceinline__ f32x4 mfma_fp8(const uint2 a, const uint2 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}

// TPW column tiles per workgroup (grid-stride: tile = blockIdx.x + j * gridDim.x) and a chunk loop over K: the split of the input rows into e4m3
// terms -- ~2.5 k VALU instructions per thread, the dominant cost of the first version, in which every one of the N / 16 workgroups of a launch
// repeated it -- is done once per 2048-column chunk per WORKGROUP and amortised over its TPW tiles; K > 2048 (the down projections) walks the
// chunks with the next (tile, chunk) segment's weights already in flight.  Each chunk carries its own power-of-two row scales, so a segment's
// term accumulators are folded into a float total right away.
template <int TPW>
__global__ __launch_bounds__(256) void gemv_mfma_fp8_kernel(const mi355_gemv_args a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) uint2 planes[];  // [NT][32 steps][2 halves][4 groups][8 rows] 8-byte pieces of ONE chunk
  __shared__ float red[4][256];
  __shared__ float st_part[4][8];
  __shared__ float st_val[8];     // the chunk's power-of-two scale per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, M = a.M;
  const int nch = (K + kKC8 - 1) / kKC8;
  constexpr int img = (kKC8 / 64) * 64;        // 8-byte pieces per term image of a chunk
  const int gi = lane >> 4, li = lane & 15;
  const uint8_t* wrow[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = blockIdx.x + j * gridDim.x;
    const int n0 = (tile < ntiles ? tile : ntiles - 1) * 16;
    const int nrow = n0 + li < a.N ? n0 + li : a.N - 1;   // tail rows / tiles recompute a valid row, never stored
    wrow[j] = (const uint8_t*)a.w + (int64_t)nrow * a.ldw + 16 * gi;
  }
  // a SEGMENT = (chunk c, tile j): this wave's k steps of the chunk are c * 32 + wave + 4 d, d = 0..7 (a chunk has at most 32 steps)
  auto issue_seg = [&](int c, int j, uint4 (&dst)[kD8]) {
    const int steps_c = ((K - c * kKC8 < kKC8 ? K - c * kKC8 : kKC8) >> 6);
#pragma unroll
    for (int d = 0; d < kD8; ++d) {
      const int sl = wave + 4 * d;
      dst[d] = make_uint4(0u, 0u, 0u, 0u);
      if (sl < steps_c) {
        const uint8_t* p = wrow[0];
#pragma unroll
        for (int jj = 1; jj < TPW; ++jj) p = j == jj ? wrow[jj] : p;
        dst[d] = *(const uint4*)(p + ((int64_t)(c * 32 + sl) << 6));
      }
    }
  };
  uint4 cur[kD8], nxt[kD8];
  issue_seg(0, 0, nxt);

  f32x4 tot[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) tot[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nch; ++c) {
    const int k0 = c * kKC8;
    const int kc = K - k0 < kKC8 ? K - k0 : kKC8;   // multiple of 64
    const int steps_c = kc >> 6;
    __syncthreads();   // the previous chunk's readers are done with `planes`
    // ---- stage + split chunk c: thread t owns the 8-column group q = t of EVERY row
    {
      const int q = tid, k = q * 8;
      const bool on = k < kc;
      float4 xa[8], xb[8];
      float4 wa = make_float4(1.f, 1.f, 1.f, 1.f), wb = wa, ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
      if (on && a.norm && a.norm_weight) { wa = *(const float4*)(a.norm_weight + k0 + k); wb = *(const float4*)(a.norm_weight + k0 + k + 4); }
      if (on && a.norm && a.norm_bias) { ba = *(const float4*)(a.norm_bias + k0 + k); bb = *(const float4*)(a.norm_bias + k0 + k + 4); }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        xa[m] = xb[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on && m < M) {
          const float* p = a.x + (int64_t)m * a.ldx + k0 + k;
          xa[m] = *(const float4*)p;
          xb[m] = *(const float4*)(p + 4);
        }
      }
      auto block_rows = [&](float (&s)[8], bool is_max) {   // per-row reduction of s[m] over the workgroup -> s[m] (sum or max)
#pragma unroll
        for (int m = 0; m < 8; ++m) s[m] = is_max ? wave_max(s[m]) : wave_sum(s[m]);
        if (lane < 8) {
          float v = s[0];
#pragma unroll
          for (int m = 1; m < 8; ++m) v = lane == m ? s[m] : v;
          st_part[wave][lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m)
          s[m] = is_max ? fmaxf(fmaxf(st_part[0][m], st_part[1][m]), fmaxf(st_part[2][m], st_part[3][m]))
                        : (st_part[0][m] + st_part[1][m]) + (st_part[2][m] + st_part[3][m]);
        __syncthreads();
      };
      if (a.norm) {  // (eligibility: a fused norm implies K <= 2048, one chunk) two-pass statistics from the registers, then the affine part
        float s[8], mean[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) s[m] = ((xa[m].x + xa[m].y) + (xa[m].z + xa[m].w)) + ((xb[m].x + xb[m].y) + (xb[m].z + xb[m].w));
        block_rows(s, false);
#pragma unroll
        for (int m = 0; m < 8; ++m) mean[m] = a.norm == 1 ? s[m] / (float)K : 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          float qv = 0.f;
          if (on) {
            const float d0 = xa[m].x - mean[m], d1 = xa[m].y - mean[m], d2 = xa[m].z - mean[m], d3 = xa[m].w - mean[m];
            const float d4 = xb[m].x - mean[m], d5 = xb[m].y - mean[m], d6 = xb[m].z - mean[m], d7 = xb[m].w - mean[m];
            qv = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
          }
          s[m] = qv;
        }
        block_rows(s, false);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float var = s[m] / (float)K;
          const float rs = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
          const float mu = mean[m];
          xa[m] = make_float4((xa[m].x - mu) * rs * wa.x + ba.x, (xa[m].y - mu) * rs * wa.y + ba.y, (xa[m].z - mu) * rs * wa.z + ba.z,
                              (xa[m].w - mu) * rs * wa.w + ba.w);
          xb[m] = make_float4((xb[m].x - mu) * rs * wb.x + bb.x, (xb[m].y - mu) * rs * wb.y + bb.y, (xb[m].z - mu) * rs * wb.z + bb.z,
                              (xb[m].w - mu) * rs * wb.w + bb.w);
        }
      }
      // per-row power-of-two scale of this chunk into [-240, 240]: s = 2^ceil(log2(amax / 240))
      float amax[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        amax[m] = 0.f;
        if (on) amax[m] = fmaxf(fmaxf(fmaxf(fabsf(xa[m].x), fabsf(xa[m].y)), fmaxf(fabsf(xa[m].z), fabsf(xa[m].w))),
                                fmaxf(fmaxf(fabsf(xb[m].x), fabsf(xb[m].y)), fmaxf(fabsf(xb[m].z), fabsf(xb[m].w))));
      }
      block_rows(amax, true);
      float inv_s[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        int e = 0;
        const float mant = frexpf(amax[m] * (1.0f / 240.0f), &e);   // amax / 240 = mant * 2^e, mant in [0.5, 1)
        const int se = amax[m] > 0.f ? (mant == 0.5f ? e - 1 : e) : 0;
        inv_s[m] = ldexpf(1.0f, -se);
        if (tid == 0) st_val[m] = ldexpf(1.0f, se);
      }
      if (on) {
        const int step = q >> 3, r = q & 7, g = r >> 1, h = r & 1;
        const int base = ((step * 2 + h) * 4 + g) * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          float v[8] = {xa[m].x * inv_s[m], xa[m].y * inv_s[m], xa[m].z * inv_s[m], xa[m].w * inv_s[m],
                        xb[m].x * inv_s[m], xb[m].y * inv_s[m], xb[m].z * inv_s[m], xb[m].w * inv_s[m]};
          uint2 terms[NT];
          split_fp8(v, terms);
#pragma unroll
          for (int t = 0; t < NT; ++t) planes[t * img + base + m] = terms[t];
        }
      }
    }
    __syncthreads();
    const float xs = st_val[li & 7];   // this lane's D column is input row li (rows >= 8 alias li & 7: never stored)
    // ---- the TPW tiles of this workgroup against chunk c
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int d = 0; d < kD8; ++d) cur[d] = nxt[d];
      if (j + 1 < TPW) issue_seg(c, j + 1, nxt);
      else if (c + 1 < nch) issue_seg(c + 1, 0, nxt);
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < kD8; ++d) {
        const int sl = wave + 4 * d;
        if (sl < steps_c) {
          const uint2 w0 = make_uint2(cur[d].x, cur[d].y), w1 = make_uint2(cur[d].z, cur[d].w);
          const int p0 = ((sl * 2 + 0) * 4 + gi) * 8 + (li & 7);
          const int p1 = ((sl * 2 + 1) * 4 + gi) * 8 + (li & 7);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t] = mfma_fp8(w0, planes[t * img + p0], acc[t]);
            acc[t] = mfma_fp8(w1, planes[t * img + p1], acc[t]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // x / s = h0 + h1 / 16 + h2 / 256 + h3 / 4096
        float v = acc[NT - 1][r];
#pragma unroll
        for (int t = NT - 2; t >= 0; --t) v = v * (1.0f / 16.0f) + acc[t][r];
        tot[j][r] += v * xs;
      }
    }
  }
  // ---- per tile: split-K through LDS, epilogue.  D layout: lane holds column (lane & 15) = input row m, rows 4 (lane >> 4) + r = n
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = blockIdx.x + j * gridDim.x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * gi + r) * 16 + li] = tot[j][r];
    __syncthreads();
    const int i = tid & 15, m = tid >> 4;
    const int n = tile * 16 + i;
    if (tile >= ntiles || m >= M || n >= a.N) continue;
    const float v0 = ((red[0][i * 16 + m] + red[1][i * 16 + m]) + (red[2][i * 16 + m] + red[3][i * 16 + m])) * a.wscale[n];
    if (a.glu) {  // rows come in (gate, up) pairs: the even thread of a pair finishes both
      if (i & 1) continue;
      const float v1 = ((red[0][(i + 1) * 16 + m] + red[1][(i + 1) * 16 + m]) + (red[2][(i + 1) * 16 + m] + red[3][(i + 1) * 16 + m])) * a.wscale[n + 1];
      const float g = v0 + (a.bias ? a.bias[n] : 0.f), u = v1 + (a.bias ? a.bias[n + 1] : 0.f);
      a.y[(int64_t)m * a.ldy + (n >> 1)] = (g / (1.0f + expf(-g))) * u * a.out_scale;
      continue;
    }
    float v = fp8_act(v0 + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
    if (a.res) v += a.res[(int64_t)m * a.ldr + n];
    if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), v * a.out_scale, a.y2_dtype);
    else a.y[(int64_t)m * a.ldy + n] = v * a.out_scale;
  }
}

}  // namespace

int mi355_gemv_mfma_fp8_eligible(const mi355_gemv_args& a) {
  static const bool off = getenv("MI355_GEMV_MFMA_FP8") != nullptr && getenv("MI355_GEMV_MFMA_FP8")[0] == '0';
  if (off) return 0;
  if (a.wdtype != MI355_W_FP8 || !a.wscale) return 0;
  if (a.M < 5 || a.M > 8 || a.rope_cos || a.x_ids) return 0;
  if (a.K % 64 || a.K < 64 || a.ldw % 16 || ((uintptr_t)a.w) % 16 || a.ldx % 4 || ((uintptr_t)a.x) % 16) return 0;
  if (a.K > kKC8 && a.norm) return 0;   // the fused norm needs whole rows in one chunk
  if (a.glu && (a.N % 2)) return 0;
  return 1;
}

int mi355_gemv_mfma_fp8_launch(const mi355_gemv_args& a, hipStream_t st) {
  static bool attr_set = false;  // benign race: the attribute is idempotent
  const size_t lds = (size_t)kKC8 * 8 * NT;   // NT terms x 32 steps x 64 pieces x 8 bytes (one chunk)
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemv_mfma_fp8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemv_mfma_fp8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e4 = hipFuncSetAttribute((const void*)gemv_mfma_fp8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MI355_REQUIRE(e1 == hipSuccess && e2 == hipSuccess && e4 == hipSuccess, "gemv(mfma fp8): cannot reserve LDS");
    attr_set = true;
  }
  const int ntiles = (a.N + 15) / 16;
  const int tpw = ntiles >= 1024 ? 4 : (ntiles >= 512 ? 2 : 1);   // tiles per workgroup: amortise the split where there are more tiles than CUs x 2
  const int grid = (ntiles + tpw - 1) / tpw;
  MI355_CLEAR_ERROR();
  if (tpw == 4) hipLaunchKernelGGL(gemv_mfma_fp8_kernel<4>, dim3(grid), dim3(256), lds, st, a, ntiles);
  else if (tpw == 2) hipLaunchKernelGGL(gemv_mfma_fp8_kernel<2>, dim3(grid), dim3(256), lds, st, a, ntiles);
  else hipLaunchKernelGGL(gemv_mfma_fp8_kernel<1>, dim3(grid), dim3(256), lds, st, a, ntiles);
  MI355_LAUNCH_CHECK("gemv(mfma fp8)");
  return MI355_OK;
}
