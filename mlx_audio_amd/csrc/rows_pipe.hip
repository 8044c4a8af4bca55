// Decode steps for a BATCH of 9..64 sequences (gfx950): a pure matrix-pipe GEMM on pre-split input planes + a row epilogue kernel.
//
// What it replaces: nn.Linear at sequence length 1 inside the reference's batched generation (tts/models/qwen3_tts/qwen3_tts.py:1651-2060
// batch_generate over talker.py:229-336 / 385-500; BASELINE config[3] names 64 utterances).  At 64 rows a Linear is still a weight stream (the
// weights are read exactly once per step), but three things that are free at <= 8 rows are not any more (measured, profiles/r3_rows_ablation_call2.txt):
//   * every workgroup re-reading and re-splitting all 64 input rows (x is 512 KB at K = 2048: with N / 16 workgroups that is 100..400 MB of L2
//     traffic and conversion work per launch, against 8..50 MB of weights);
//   * 16-row weight tiles read from a row-major image (16 x 128-byte pieces 4 KB apart per load: every workgroup camps on the same HBM channels);
//   * few column tiles (N = 2048: 128) leaving half the CUs without a weight stream.
// So the step is cut differently here:
//   mi355_rows_gemm    grid = (column groups, K groups).  A workgroup owns T adjacent 16-column tiles and a range of 64-wide k steps; its four waves
//                      take interleaved steps.  A operand = a lane's 32 bytes of the TILE IMAGE (mi355_pack_tiles16_host: [tile][k step][lane][16]:
//                      2 KB contiguous per wave and step, consecutive steps contiguous -- a pure stream); B operand = the input rows as hi + lo
//                      images of the weights' 16-bit type ALREADY in MFMA fragment order in global memory ("planes": [k step][image][half][group][row] x 16
//                      bytes), loaded straight into registers -- no LDS staging, no conversion, no barrier in the loop.  The four partial tiles of
//                      the waves meet in LDS; the workgroup writes ONE fp32 partial tile to slab `kgroup` of the workspace.  No epilogue at all.
//   mi355_rows_finish  one workgroup per row: sums the K-group slabs in a fixed order (deterministic), then bias / activation / LayerScale /
//                      residual / SwiGLU / split destinations (q -> buffer, k | v -> KV-cache slot) exactly as mi355_gemv, and -- what makes the
//                      next GEMM's input free -- optionally normalises the finished row (LayerNorm / RMSNorm with REAL row statistics, two-pass) and
//                      writes it as planes.  With kgroups = 1 it is also the converter fp32 rows -> planes (attention output, step input).
// Input precision: hi + lo images = ~16 mantissa bits for bf16 weights (~22 for fp16), the split of the prefill GEMMs and of gemv_mfma.hip.
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __forceinline__ float pipe_act(float v, int act, float slope) {
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

template <bool F16>
__device__ __forceinline__ void pipe_split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
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
__device__ __forceinline__ f32x4 pipe_mfma(const uint4 a, const uint4 b, const f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------- the GEMM
// MR = row groups of 16 in the planes (R = 16 MR rows), T = adjacent column tiles per workgroup.
// eight OCP e4m3fn bytes -> eight bf16 values (exact: 4 significant bits, the exponent range of bf16 covers e4m3's)
__device__ __forceinline__ uint4 fp8x8_to_bf16x8(const uint2 v) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(b[0], b[1]), pack_bf16x2(c[0], c[1]), pack_bf16x2(d[0], d[1]));
}

// WT: element type of the tile image: MI355_W_BF16 / MI355_W_F16 (planes of the same type) / MI355_W_FP8 (e4m3 bytes decoded to bf16 in registers, bf16 planes)
template <int MR, int T, int WT>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(const mi355_rows_gemm_args a, const int ntiles, const int steps_total, const int spg) {
  constexpr bool F16 = WT == MI355_W_F16, FP8 = WT == MI355_W_FP8;
  constexpr int R = 16 * MR;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [wave][tile][row group][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gi = lane >> 4, li = lane & 15;
  const int t0 = blockIdx.x * T, kg = blockIdx.y;
  const int s_begin = kg * spg, s_end = s_begin + spg < steps_total ? s_begin + spg : steps_total;
  const uint4* const wt = (const uint4*)a.wt;
  const uint4* const pl = (const uint4*)a.planes;
  // A: uint4 index ((tile * steps_total + s) * 64 + lane) * 2 + half;   tiles past the end re-read the last tile (never stored)
  int64_t abase[T];
#pragma unroll
  for (int j = 0; j < T; ++j) abase[j] = ((int64_t)(t0 + j < ntiles ? t0 + j : ntiles - 1) * steps_total * 64 + lane) * 2;
  const int bbase = gi * R + li;   // B: uint4 index (((s * 2 + image) * 2 + half) * 4 + group) * R + 16 rg + row

  f32x4 acc[T][MR];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 A[T][2];              // the weights of a step: [tile][half]; a half is reloaded for the next step as soon as its MFMAs are issued
  uint2 A8[T][2];             // fp8 image: the raw bytes of a half (decoded right before use)
  uint4 B0[2][MR], B1[2][MR]; // the two halves of a step of the input rows: [image][row group]
  auto load_a = [&](const int s, const int h) {
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if constexpr (FP8) A8[j][h] = ((const uint2*)a.wt)[abase[j] + (int64_t)s * 128 + h];   // 16 bytes per lane and step: the same index arithmetic in 8-byte units
      else A[j][h] = wt[abase[j] + (int64_t)s * 128 + h];
    }
  };
  auto load_b = [&](const int s, const int h, uint4 (&dst)[2][MR]) {
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int r = 0; r < MR; ++r) dst[im][r] = pl[(((s * 2 + im) * 2 + h) * 4) * R + bbase + 16 * r];
  };
  auto mfma_half = [&](const int h, const uint4 (&B)[2][MR]) {
    if constexpr (FP8) {
#pragma unroll
      for (int j = 0; j < T; ++j) A[j][h] = fp8x8_to_bf16x8(A8[j][h]);
    }
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
      for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j][r] = pipe_mfma<F16>(A[j][h], B[im][r], acc[j][r]);
  };
  int s = s_begin + wave;
  if (s < s_end) {
    load_a(s, 0);
    load_b(s, 0, B0);
    load_a(s, 1);
  }
  for (; s < s_end; s += 4) {
    const int sn = s + 4;
    load_b(s, 1, B1);
    mfma_half(0, B0);
    if (sn < s_end) {
      load_a(sn, 0);
      load_b(sn, 0, B0);
    }
    mfma_half(1, B1);
    if (sn < s_end) load_a(sn, 1);
  }

  // ---- the four waves' partial tiles through LDS.  D layout: lane holds column li = input row (inside its group), rows 4 gi + e = n
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[((wave * T + j) * MR + r) * 256 + (4 * gi + e) * 16 + li] = acc[j][r][e];
  __syncthreads();
  if (a.glu_planes_out) {   // complete sums (one K group): SwiGLU here, the result leaves as planes -- thread (row m, tile j) builds the piece of outputs 8 (t0 + j) .. + 8
    for (int u = tid; u < a.M * T; u += 256) {
      const int m = u / T, j = u - m * T, tile = t0 + j;
      if (tile >= ntiles) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float gv = 0.f, uv = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int b = ((w * T + j) * MR + (m >> 4)) * 256 + (m & 15);
          gv += red[b + (2 * e) * 16];
          uv += red[b + (2 * e + 1) * 16];
        }
        if (a.wscale) { gv *= a.wscale[tile * 16 + 2 * e]; uv *= a.wscale[tile * 16 + 2 * e + 1]; }
        if (a.glu_bias) { gv += a.glu_bias[tile * 16 + 2 * e]; uv += a.glu_bias[tile * 16 + 2 * e + 1]; }
        o[e] = (gv / (1.0f + expf(-gv))) * uv;
      }
      uint4 hi, lo;
      pipe_split2<F16>(o[0], o[1], hi.x, lo.x);
      pipe_split2<F16>(o[2], o[3], hi.y, lo.y);
      pipe_split2<F16>(o[4], o[5], hi.z, lo.z);
      pipe_split2<F16>(o[6], o[7], hi.w, lo.w);
      const int p = tile, s = p >> 3, g = (p & 7) >> 1, h = p & 1;
      uint4* const po = (uint4*)a.glu_planes_out;
      po[(((s * 2 + 0) * 2 + h) * 4 + g) * R + m] = hi;
      po[(((s * 2 + 1) * 2 + h) * 4 + g) * R + m] = lo;
    }
    return;
  }
  constexpr int CW = 16 * T;            // columns of the workgroup
  constexpr int RSTEP = 256 / CW;       // rows written per pass
  const int c = tid % CW, r0 = tid / CW;
  const int n = t0 * 16 + c;
  if (n >= a.N) return;
  const int j = c >> 4, i = c & 15;
  float* const out = a.part + (int64_t)kg * a.kg_stride + n;
  for (int m = r0; m < a.M; m += RSTEP) {
    const int o = (j * MR + (m >> 4)) * 256 + i * 16 + (m & 15);
    out[(int64_t)m * a.ldp] = (red[o] + red[T * MR * 256 + o]) + (red[2 * T * MR * 256 + o] + red[3 * T * MR * 256 + o]);
  }
}

template <int MR, int T, int WT>
int launch_rows_gemm(const mi355_rows_gemm_args& a, hipStream_t st, const int ntiles, const int steps_total, const int spg) {
  static bool attr_set = false;  // benign race: the attribute is idempotent
  constexpr size_t lds = (size_t)4 * T * MR * 256 * sizeof(float);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)rows_gemm_kernel<MR, T, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MI355_REQUIRE(e == hipSuccess, "rows_gemm: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const dim3 grid((ntiles + T - 1) / T, a.kgroups);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((rows_gemm_kernel<MR, T, WT>), grid, dim3(256), lds, st, a, ntiles, steps_total, spg);
  MI355_LAUNCH_CHECK("rows_gemm");
  return MI355_OK;
}

template <int MR, int WT>
int launch_rows_gemm_t(const mi355_rows_gemm_args& a, hipStream_t st, const int T, const int ntiles, const int steps_total, const int spg) {
  if (T == 4) return launch_rows_gemm<MR, 4, WT>(a, st, ntiles, steps_total, spg);
  if (T == 2) return launch_rows_gemm<MR, 2, WT>(a, st, ntiles, steps_total, spg);
  return launch_rows_gemm<MR, 1, WT>(a, st, ntiles, steps_total, spg);
}

template <int MR>
int launch_rows_gemm_w(const mi355_rows_gemm_args& a, hipStream_t st, const int T, const int ntiles, const int steps_total, const int spg) {
  if (a.wdtype == MI355_W_FP8) return launch_rows_gemm_t<MR, MI355_W_FP8>(a, st, T, ntiles, steps_total, spg);
  if (a.wdtype == MI355_W_F16) return launch_rows_gemm_t<MR, MI355_W_F16>(a, st, T, ntiles, steps_total, spg);
  return launch_rows_gemm_t<MR, MI355_W_BF16>(a, st, T, ntiles, steps_total, spg);
}

int tiles_per_wg(const int N) {
  static const int t_env = getenv("MI355_ROWS_T") ? atoi(getenv("MI355_ROWS_T")) : 0;   // A/B knob
  if (t_env == 1 || t_env == 2 || t_env == 4) return t_env;
  // measured (profiles/r3_rows_sweep_call4.txt): two tiles per workgroup (132 registers: three workgroups per CU) beat four (200 registers) on every
  // decode shape of the Qwen3-TTS stacks although the input planes are then read by twice as many workgroups -- latency hiding, not L2 traffic, binds;
  // very wide outputs (vocabulary heads) keep four tiles
  const int ntiles = (N + 15) / 16;
  return ntiles >= 2048 ? 4 : (ntiles >= 32 ? 2 : 1);
}

// ---------------------------------------------------------------------------------------------- the row epilogue
constexpr int kMaxPieces = 4;   // pieces of 8 outputs a thread keeps in registers when the row is normalised: rows of up to 8192 outputs

template <bool F16>
__global__ __launch_bounds__(256) void rows_finish_kernel(const mi355_rows_finish_args a) {
  __shared__ float sred[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x;
  const int No = a.glu ? a.N / 2 : a.N;
  const int npieces = (No + 7) / 8;
  const float* const prow = a.part + (int64_t)m * a.ldp;
  // the finished outputs 8 p .. 8 p + 7 of this row (stored to y / y2 on the way); returns their sum over the valid columns
  auto piece = [&](const int p, float (&w)[8]) -> float {
    const int n0 = 8 * p;
    if (a.glu) {
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
      auto add4 = [](float4& s, const float4 t) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; };
      int g = 0;
      for (; g + 2 <= a.kgroups; g += 2) {   // two slabs' loads in flight, added in slab order
        const float* q = prow + (int64_t)g * a.kg_stride + 2 * n0;
        const float* q2 = q + a.kg_stride;
        const float4 t0 = *(const float4*)q, t1 = *(const float4*)(q + 4), t2 = *(const float4*)(q + 8), t3 = *(const float4*)(q + 12);
        const float4 u0 = *(const float4*)q2, u1 = *(const float4*)(q2 + 4), u2 = *(const float4*)(q2 + 8), u3 = *(const float4*)(q2 + 12);
        add4(s0, t0); add4(s1, t1); add4(s2, t2); add4(s3, t3);
        add4(s0, u0); add4(s1, u1); add4(s2, u2); add4(s3, u3);
      }
      for (; g < a.kgroups; ++g) {
        const float* q = prow + (int64_t)g * a.kg_stride + 2 * n0;
        add4(s0, *(const float4*)q); add4(s1, *(const float4*)(q + 4)); add4(s2, *(const float4*)(q + 8)); add4(s3, *(const float4*)(q + 12));
      }
      const float gu[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = a.wscale ? a.wscale[2 * (n0 + e)] : 1.f, su = a.wscale ? a.wscale[2 * (n0 + e) + 1] : 1.f;
        const float g = gu[2 * e] * sg + (a.bias ? a.bias[2 * (n0 + e)] : 0.f), u = gu[2 * e + 1] * su + (a.bias ? a.bias[2 * (n0 + e) + 1] : 0.f);
        w[e] = (g / (1.0f + expf(-g))) * u * a.out_scale;
      }
    } else {
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
      auto add4 = [](float4& s, const float4 t) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; };
      int g = 0;
      for (; g + 4 <= a.kgroups; g += 4) {   // four slabs' loads in flight, added in slab order
        const float* q = prow + (int64_t)g * a.kg_stride + n0;
        float4 t[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) { t[u][0] = *(const float4*)(q + u * a.kg_stride); t[u][1] = *(const float4*)(q + u * a.kg_stride + 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { add4(s0, t[u][0]); add4(s1, t[u][1]); }
      }
      for (; g < a.kgroups; ++g) {
        const float* q = prow + (int64_t)g * a.kg_stride + n0;
        add4(s0, *(const float4*)q); add4(s1, *(const float4*)(q + 4));
      }
      const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = n0 + e;
        float t = 0.f;
        if (n < No) {
          t = pipe_act(sv[e] * (a.wscale ? a.wscale[n] : 1.f) + (a.bias ? a.bias[n] : 0.f), a.post_act, a.post_slope) * (a.colscale ? a.colscale[n] : 1.f);
          if (a.res) t += a.res[(int64_t)m * a.ldr + n];
          t *= a.out_scale;
        }
        w[e] = t;
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = n0 + e;
      if (n < No) {
        if (a.y2 && n >= a.split) store_kv_elem(a.y2, (int64_t)m * a.ldy2 + (n - a.split), w[e], a.y2_dtype);
        else if (a.y) a.y[(int64_t)m * a.ldy + n] = w[e];
        sum += w[e];
      } else {
        w[e] = 0.f;
      }
    }
    return sum;
  };
  uint4* const pl = (uint4*)a.planes;
  auto emit = [&](const int p, const float (&t)[8]) {   // the (normalised) piece as fp32 (yn) and / or as planes
    if (a.yn) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (8 * p + e < No) a.yn[(int64_t)m * a.ldyn + 8 * p + e] = t[e];
    }
    if (pl) {
      uint4 hi, lo;
      pipe_split2<F16>(t[0], t[1], hi.x, lo.x);
      pipe_split2<F16>(t[2], t[3], hi.y, lo.y);
      pipe_split2<F16>(t[4], t[5], hi.z, lo.z);
      pipe_split2<F16>(t[6], t[7], hi.w, lo.w);
      const int s = p >> 3, g = (p & 7) >> 1, h = p & 1;
      pl[(((s * 2 + 0) * 2 + h) * 4 + g) * a.R + m] = hi;
      pl[(((s * 2 + 1) * 2 + h) * 4 + g) * a.R + m] = lo;
    }
  };
  if (!a.norm) {   // nothing needs the whole row: stream the pieces (rows of any length, gridDim.y column blocks)
    for (int p = blockIdx.y * 256 + tid; p < npieces; p += 256 * gridDim.y) {
      float w[8];
      piece(p, w);
      emit(p, w);
    }
    return;
  }
  float v[kMaxPieces][8], nwr[kMaxPieces][8], nbr[kMaxPieces][8];   // the row, and the norm weight / bias of its columns (loaded next to the slabs)
  float lsum = 0.f;
#pragma unroll
  for (int it = 0; it < kMaxPieces; ++it) {
    const int p = tid + 256 * it;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[it][e] = 0.f; nwr[it][e] = 1.f; nbr[it][e] = 0.f; }
    if (p < npieces) {
      if (No % 8 == 0) {
        if (a.norm_weight) { const float4 w0 = *(const float4*)(a.norm_weight + 8 * p), w1 = *(const float4*)(a.norm_weight + 8 * p + 4);
          nwr[it][0] = w0.x; nwr[it][1] = w0.y; nwr[it][2] = w0.z; nwr[it][3] = w0.w; nwr[it][4] = w1.x; nwr[it][5] = w1.y; nwr[it][6] = w1.z; nwr[it][7] = w1.w; }
        if (a.norm_bias) { const float4 b0 = *(const float4*)(a.norm_bias + 8 * p), b1 = *(const float4*)(a.norm_bias + 8 * p + 4);
          nbr[it][0] = b0.x; nbr[it][1] = b0.y; nbr[it][2] = b0.z; nbr[it][3] = b0.w; nbr[it][4] = b1.x; nbr[it][5] = b1.y; nbr[it][6] = b1.z; nbr[it][7] = b1.w; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * p + e < No) { if (a.norm_weight) nwr[it][e] = a.norm_weight[8 * p + e]; if (a.norm_bias) nbr[it][e] = a.norm_bias[8 * p + e]; }
      }
      lsum += piece(p, v[it]);
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (a.norm) {   // two-pass statistics of the finished row
    if (a.norm == 1) {
      const float s = wave_sum_fast(lsum);   // every lane is here (no early exit above): the DPP reduction
      if (lane == 0) sred[wave] = s;
      __syncthreads();
      mean = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)No;
    }
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxPieces; ++it) {
      const int p = tid + 256 * it;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (p < npieces && 8 * p + e < No) { const float d = v[it][e] - mean; q += d * d; }
    }
    q = wave_sum_fast(q);
    if (lane == 0) sred[4 + wave] = q;
    __syncthreads();
    const float var = ((sred[4] + sred[5]) + (sred[6] + sred[7])) / (float)No;
    rstd = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
  }
#pragma unroll
  for (int it = 0; it < kMaxPieces; ++it) {
    const int p = tid + 256 * it;
    if (p >= npieces) continue;
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = 8 * p + e;
      t[e] = n < No ? (v[it][e] - mean) * rstd * nwr[it][e] + nbr[it][e] : 0.f;
    }
    emit(p, t);
  }
}

// rows_finish_lean_kernel: the row epilogue of the decode stacks (o-proj / down-proj / step input: slab sums + bias + LayerScale + residual ->
// y, norm -> planes / yn) as STRAIGHT-LINE code.  rows_finish_kernel answers every flag of mi355_rows_finish_args at run time; hipcc compiles each
// `ptr ? ptr[n] : c` of it to a branch around a load with its own s_waitcnt vmcnt(0), so a launch was a chain of serial L2 round trips (6.9 us in
// situ for 2 MB of slabs, 370 launches per Qwen3-TTS frame: tools/scan_serial_waits.py counts them) in 24 000 instructions of generated code.
// Here: whole 8-column pieces, NP of them per thread; EVERY operand is loaded unconditionally before the first use (a null operand is replaced by
// the slab row -- valid memory -- and dropped by a select), threads past the row clamp to its last piece and only their stores are predicated.
// mi355_rows_finish dispatches here when the call has this shape (no GLU, no split destination, aligned operands, <= 4096 outputs).
template <bool F16, int NP>
__global__ __launch_bounds__(256) void rows_finish_lean_kernel(const mi355_rows_finish_args a) {
  __shared__ float sred[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x;
  const int No = a.N, npieces = No >> 3;
  const float* const prow = a.part + (int64_t)m * a.ldp;
  const bool has_ws = a.wscale != nullptr, has_b = a.bias != nullptr, has_cs = a.colscale != nullptr, has_res = a.res != nullptr;
  const bool has_nw = a.norm_weight != nullptr, has_nb = a.norm_bias != nullptr;
  auto ld8 = [](const float* q, float (&d)[8]) {
    const float4 t0 = *(const float4*)q, t1 = *(const float4*)(q + 4);
    d[0] = t0.x; d[1] = t0.y; d[2] = t0.z; d[3] = t0.w; d[4] = t1.x; d[5] = t1.y; d[6] = t1.z; d[7] = t1.w;
  };
  int pc[NP];
  bool live[NP];
  float wsv[NP][8], bv[NP][8], cv[NP][8], rv[NP][8], nwv[NP][8], nbv[NP][8];
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int p = tid + 256 * it;
    live[it] = p < npieces;
    pc[it] = live[it] ? p : npieces - 1;
    const int n0 = 8 * pc[it];
    const float* dummy = prow + n0;
    ld8(has_ws ? a.wscale + n0 : dummy, wsv[it]);
    ld8(has_b ? a.bias + n0 : dummy, bv[it]);
    ld8(has_cs ? a.colscale + n0 : dummy, cv[it]);
    ld8(has_res ? a.res + (int64_t)m * a.ldr + n0 : dummy, rv[it]);
    ld8(has_nw ? a.norm_weight + n0 : dummy, nwv[it]);
    ld8(has_nb ? a.norm_bias + n0 : dummy, nbv[it]);
  }
  float v[NP][8];
#pragma unroll
  for (int it = 0; it < NP; ++it)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
  int g = 0;
  for (; g + 4 <= a.kgroups; g += 4) {   // four slabs' loads in flight per piece, added in slab order (deterministic)
    float t[NP][4][8];
#pragma unroll
    for (int it = 0; it < NP; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8(prow + (int64_t)(g + u) * a.kg_stride + 8 * pc[it], t[it][u]);
#pragma unroll
    for (int it = 0; it < NP; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] += t[it][u][e];
  }
  for (; g < a.kgroups; ++g) {
    float t[NP][8];
#pragma unroll
    for (int it = 0; it < NP; ++it) ld8(prow + (int64_t)g * a.kg_stride + 8 * pc[it], t[it]);
#pragma unroll
    for (int it = 0; it < NP; ++it)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[it][e] += t[it][e];
  }
  float lsum = 0.f;
#pragma unroll
  for (int it = 0; it < NP; ++it)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] * (has_ws ? wsv[it][e] : 1.f) + (has_b ? bv[it][e] : 0.f);
  if (a.post_act != MI355_ACT_NONE) {   // one switch around the thread's values (the MLP-in epilogue of a non-gated stack: Whisper's GELU)
    auto each = [&](auto f) {
#pragma unroll
      for (int it = 0; it < NP; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = f(v[it][e]);
    };
    switch (a.post_act) {
      case MI355_ACT_LEAKY: each([&](float x) { return x > 0.f ? x : x * a.post_slope; }); break;
      case MI355_ACT_GELU: each([](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }); break;
      case MI355_ACT_SILU: each([](float x) { return x / (1.0f + expf(-x)); }); break;
      case MI355_ACT_GELU_TANH: each([](float x) { return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); }); break;
      case MI355_ACT_ELU: each([](float x) { return x > 0.f ? x : expm1f(x); }); break;
      case MI355_ACT_TANH: each([](float x) { return tanhf(x); }); break;
      default: break;
    }
  }
#pragma unroll
  for (int it = 0; it < NP; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[it][e] * (has_cs ? cv[it][e] : 1.f);
      t += has_res ? rv[it][e] : 0.f;
      t *= a.out_scale;
      v[it][e] = live[it] ? t : 0.f;
      lsum += v[it][e];
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (a.norm) {   // two-pass statistics of the finished row, the arithmetic of rows_finish_kernel
    if (a.norm == 1) {
      const float s = wave_sum_fast(lsum);
      if (lane == 0) sred[wave] = s;
      __syncthreads();
      mean = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)No;
    }
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NP; ++it)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = live[it] ? v[it][e] - mean : 0.f;
        q += d * d;
      }
    q = wave_sum_fast(q);
    if (lane == 0) sred[4 + wave] = q;
    __syncthreads();
    const float var = ((sred[4] + sred[5]) + (sred[6] + sred[7])) / (float)No;
    rstd = a.norm == 1 ? 1.0f / sqrtf(var + a.norm_eps) : rsqrtf(var + a.norm_eps);
  }
  uint4* const pl = (uint4*)a.planes;
  // every load has long landed; an opaque use makes the compiler say so HERE -- behind the conditional y stores its wait-count bookkeeping can no
  // longer tell loads from stores and would wait for the stores (vmcnt(0)) in front of the plane stores
#pragma unroll
  for (int it = 0; it < NP; ++it)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(nwv[it][e]), "+v"(nbv[it][e]));
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    if (!live[it]) continue;
    const int p = pc[it];
    if (a.y) {   // stored down here: a store in front of the statistics would sit in every later s_waitcnt vmcnt(0) of the generated code
      float* yp = a.y + (int64_t)m * a.ldy + 8 * p;
      *(float4*)yp = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
      *(float4*)(yp + 4) = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
    }
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      t[e] = a.norm ? (v[it][e] - mean) * rstd * (has_nw ? nwv[it][e] : 1.f) + (has_nb ? nbv[it][e] : 0.f) : v[it][e];
    if (a.yn) {
      float* yp = a.yn + (int64_t)m * a.ldyn + 8 * p;
      *(float4*)yp = make_float4(t[0], t[1], t[2], t[3]);
      *(float4*)(yp + 4) = make_float4(t[4], t[5], t[6], t[7]);
    }
    if (pl) {
      uint4 hi, lo;
      pipe_split2<F16>(t[0], t[1], hi.x, lo.x);
      pipe_split2<F16>(t[2], t[3], hi.y, lo.y);
      pipe_split2<F16>(t[4], t[5], hi.z, lo.z);
      pipe_split2<F16>(t[6], t[7], hi.w, lo.w);
      const int s = p >> 3, gg = (p & 7) >> 1, h = p & 1;
      pl[(((s * 2 + 0) * 2 + h) * 4 + gg) * a.R + m] = hi;
      pl[(((s * 2 + 1) * 2 + h) * 4 + gg) * a.R + m] = lo;
    }
  }
}

}  // namespace

// K groups mi355_rows_gemm splits a [N, K] projection into (what the caller must size the slab workspace for and hand to mi355_rows_finish)
extern "C" int32_t mi355_rows_kgroups(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || K % 64) return 0;
  static const int kg_env = getenv("MI355_ROWS_KG") ? atoi(getenv("MI355_ROWS_KG")) : 0;      // A/B knobs
  static const int wg_env = getenv("MI355_ROWS_WGS") ? atoi(getenv("MI355_ROWS_WGS")) : 0;
  const int steps = K / 64, T = tiles_per_wg(N);
  const int ng = ((N + 15) / 16 + T - 1) / T;
  int kg = ((wg_env > 0 ? wg_env : 512) + ng / 2) / ng;
  if (kg_env > 0) kg = kg_env;
  const int cap = steps / 4 > 0 ? steps / 4 : 1;   // at least one k step per wave
  if (kg > cap) kg = cap;
  if (kg < 1) kg = 1;
  const int spg = (steps + kg - 1) / kg;
  return (steps + spg - 1) / spg;                  // groups that actually hold steps
}

extern "C" int mi355_rows_gemm(const mi355_rows_gemm_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->wt && ap->planes && (ap->part || ap->glu_planes_out), "rows_gemm: null tensor");
  MI355_REQUIRE(!ap->glu_planes_out || (ap->kgroups == 1 && ap->N % 128 == 0 && ((uintptr_t)ap->glu_planes_out) % 16 == 0),
                "rows_gemm: the fused SwiGLU epilogue needs one K group and N %% 128 == 0");
  const mi355_rows_gemm_args a = *ap;
  MI355_REQUIRE(a.wdtype == MI355_W_BF16 || a.wdtype == MI355_W_F16 || a.wdtype == MI355_W_FP8, "rows_gemm: wdtype must be MI355_W_BF16, MI355_W_F16 or MI355_W_FP8");
  MI355_REQUIRE(a.N > 0 && a.K >= 64 && a.K % 64 == 0, "rows_gemm: K must be a positive multiple of 64 (got %d)", a.K);
  MI355_REQUIRE(a.R == 16 || a.R == 32 || a.R == 64, "rows_gemm: planes hold 16, 32 or 64 rows (got %d)", a.R);
  MI355_REQUIRE(a.M >= 1 && a.M <= a.R, "rows_gemm: M must be in [1, R] (got %d, R = %d)", a.M, a.R);
  MI355_REQUIRE(((uintptr_t)a.wt) % 16 == 0 && ((uintptr_t)a.planes) % 16 == 0, "rows_gemm: images must be 16-byte aligned");
  MI355_REQUIRE((a.glu_planes_out || a.ldp >= a.N) && a.kgroups >= 1 && a.kgroups <= a.K / 64, "rows_gemm: bad slab geometry (ldp %d, kgroups %d)", a.ldp, a.kgroups);
  MI355_REQUIRE(a.kgroups == 1 || a.kg_stride >= (int64_t)a.M * a.ldp, "rows_gemm: slabs overlap");
  const int steps = a.K / 64, ntiles = (a.N + 15) / 16, T = tiles_per_wg(a.N);
  const int spg = (steps + a.kgroups - 1) / a.kgroups;
  MI355_REQUIRE((steps + spg - 1) / spg == a.kgroups, "rows_gemm: %d K groups leave empty slabs for %d k steps (use mi355_rows_kgroups)", a.kgroups, steps);
  hipStream_t st = (hipStream_t)stream;
  if (a.R == 16) return launch_rows_gemm_w<1>(a, st, T, ntiles, steps, spg);
  if (a.R == 32) return launch_rows_gemm_w<2>(a, st, T, ntiles, steps, spg);
  return launch_rows_gemm_w<4>(a, st, T, ntiles, steps, spg);
}

extern "C" int mi355_rows_finish(const mi355_rows_finish_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->part, "rows_finish: null input");
  mi355_rows_finish_args a = *ap;
  MI355_REQUIRE(a.M >= 1 && a.N >= 1 && a.kgroups >= 1, "rows_finish: bad shape");
  MI355_REQUIRE(a.ldp % 4 == 0 && a.kg_stride % 4 == 0 && ((uintptr_t)a.part) % 16 == 0, "rows_finish: the slabs must be 16-byte aligned");
  MI355_REQUIRE(!a.glu || (a.N % 2 == 0 && !a.res && !a.colscale && a.post_act == MI355_ACT_NONE && !a.y2), "rows_finish: glu needs an even N and a plain epilogue");
  const int No = a.glu ? a.N / 2 : a.N;
  MI355_REQUIRE(a.glu ? a.N % 16 == 0 : a.N % 8 == 0 || a.ldp >= ((a.N + 7) / 8) * 8, "rows_finish: rows must be padded to whole 8-column pieces");
  MI355_REQUIRE(No <= 8 * 256 * kMaxPieces || !a.norm, "rows_finish: a normalised row has at most %d outputs (got %d)", 8 * 256 * kMaxPieces, No);
  MI355_REQUIRE(!a.norm || ((!a.norm_weight || ((uintptr_t)a.norm_weight) % 16 == 0) && (!a.norm_bias || ((uintptr_t)a.norm_bias) % 16 == 0)),
                "rows_finish: norm weight / bias must be 16-byte aligned");
  MI355_REQUIRE(a.norm >= 0 && a.norm <= 2, "rows_finish: norm must be 0 (none), 1 (LayerNorm) or 2 (RMSNorm)");
  MI355_REQUIRE(!a.y2 || (a.split > 0 && a.split < a.N && !a.planes && !a.yn && !a.norm), "rows_finish: a split destination excludes planes / normalised outputs");
  MI355_REQUIRE(!a.yn || a.norm || !a.y, "rows_finish: yn without a norm would repeat y");
  MI355_REQUIRE(a.y2_dtype >= MI355_KV_F32 && a.y2_dtype <= MI355_KV_F16, "rows_finish: y2_dtype must be MI355_KV_F32, MI355_KV_BF16 or MI355_KV_F16");
  MI355_REQUIRE(!a.planes || ((a.R == 16 || a.R == 32 || a.R == 64) && a.M <= a.R && No % 64 == 0 && ((uintptr_t)a.planes) % 16 == 0 &&
                              (a.planes_dtype == MI355_W_BF16 || a.planes_dtype == MI355_W_F16)),
                "rows_finish: planes need R in {16, 32, 64} >= M, a row length that is a multiple of 64 and a 16-bit element type");
  MI355_REQUIRE(a.y || a.y2 || a.planes || a.yn, "rows_finish: no destination");
  if (a.out_scale == 0.f) a.out_scale = 1.f;
  MI355_CLEAR_ERROR();
  {   // the straight-line kernel for the shape the decode stacks use (see rows_finish_lean_kernel)
    static const bool lean_off = getenv("MI355_ROWS_FINISH_OLD") != nullptr && getenv("MI355_ROWS_FINISH_OLD")[0] == '1';   // A/B knob
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool lean = !lean_off && !a.glu && !a.y2 && No % 8 == 0 && No <= 4096 && al16(a.wscale) && al16(a.bias) &&
                      al16(a.colscale) && al16(a.res) && al16(a.norm_weight) && al16(a.norm_bias) && (!a.res || a.ldr % 4 == 0) &&
                      (!a.y || (al16(a.y) && a.ldy % 4 == 0)) && (!a.yn || (al16(a.yn) && a.ldyn % 4 == 0));
    if (lean) {
      const bool f16 = a.planes && a.planes_dtype == MI355_W_F16, two = No > 2048;
      if (f16 && two) hipLaunchKernelGGL((rows_finish_lean_kernel<true, 2>), dim3(a.M), dim3(256), 0, (hipStream_t)stream, a);
      else if (f16) hipLaunchKernelGGL((rows_finish_lean_kernel<true, 1>), dim3(a.M), dim3(256), 0, (hipStream_t)stream, a);
      else if (two) hipLaunchKernelGGL((rows_finish_lean_kernel<false, 2>), dim3(a.M), dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((rows_finish_lean_kernel<false, 1>), dim3(a.M), dim3(256), 0, (hipStream_t)stream, a);
      MI355_LAUNCH_CHECK("rows_finish (lean)");
      return MI355_OK;
    }
  }
  int cb = 1;   // column blocks of a row: only a normalised row needs one workgroup to see all of it
  if (!a.norm) { cb = ((No + 7) / 8 + 255) / 256; if (cb > 16) cb = 16; }
  if (a.planes && a.planes_dtype == MI355_W_F16) hipLaunchKernelGGL(rows_finish_kernel<true>, dim3(a.M, cb), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(rows_finish_kernel<false>, dim3(a.M, cb), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("rows_finish");
  return MI355_OK;
}

// uint8 [N, K] (host: the e4m3 codes of mi355_pack_rowmajor_fp8_host) -> the fp8 tile image: the same permutation as mi355_pack_tiles16_host, one byte per element
extern "C" int mi355_pack_tiles8_host(const uint8_t* codes, int64_t N, int64_t K, uint8_t* out) {
  MI355_REQUIRE(codes && out && N > 0 && K > 0 && K % 64 == 0, "pack_tiles8: bad arguments (K must be a multiple of 64)");
  const int64_t ntiles = (N + 15) / 16, steps = K / 64;
  for (int64_t t = 0; t < ntiles; ++t)
    for (int64_t s = 0; s < steps; ++s)
      for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 16; ++i) {
          uint8_t* o = out + (((t * steps + s) * 4 + g) * 16 + i) * 16;
          const int64_t n = 16 * t + i;
          for (int e = 0; e < 16; ++e) o[e] = n < N ? codes[n * K + 64 * s + 16 * g + e] : 0;
        }
  return MI355_OK;
}

// fp32 [N, K] (host) -> the tile image mi355_rows_gemm streams: [ceil(N / 16)][K / 64][lane = 16 group + row][16 elements] with
// element e of (tile t, step s, group g, row i) = W[16 t + i][64 s + 16 g + e]; rows past N are zero.  out: ceil(N / 16) * 16 * K elements.
extern "C" int mi355_pack_tiles16_host(const float* w, int64_t N, int64_t K, int32_t dtype, uint16_t* out) {
  MI355_REQUIRE(w && out && N > 0 && K > 0 && K % 64 == 0, "pack_tiles16: bad arguments (K must be a multiple of 64)");
  MI355_REQUIRE(dtype == MI355_W_BF16 || dtype == MI355_W_F16, "pack_tiles16: dtype must be MI355_W_BF16 or MI355_W_F16");
  const int64_t ntiles = (N + 15) / 16, steps = K / 64;
  for (int64_t t = 0; t < ntiles; ++t)
    for (int64_t s = 0; s < steps; ++s)
      for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 16; ++i) {
          uint16_t* o = out + (((t * steps + s) * 4 + g) * 16 + i) * 16;
          const int64_t n = 16 * t + i;
          for (int e = 0; e < 16; ++e) {
            const float v = n < N ? w[n * K + 64 * s + 16 * g + e] : 0.f;
            o[e] = dtype == MI355_W_F16 ? host_f32_to_f16(v) : host_f32_to_bf16(v);
          }
        }
  return MI355_OK;
}
