// Instance-norm statistics -> AdaIN coefficients, and row LayerNorm (gfx950).
// Reference call sites: InstanceNorm1d / AdaIN1d (tts/models/kokoro/istftnet.py:173-338),
// nn.LayerNorm + AdaLayerNorm (tts/models/kokoro/modules.py:35,71-90,445,481,523,537,551).
#include "common.h"

namespace {

// Partial per-channel sum / sum-of-squares in float64 (robust against cancellation without a second
// pass over HBM): grid (ceil(C/32), nsplit, B), 256 threads = 32 row groups x 8 float4 lanes, so a
// wave reads 8 full 128-B row segments per instruction.
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const mi355_adain_coef_args a, int rows_per_split) {
  __shared__ double red[2][32][33];
  const int tid = threadIdx.x, cg = tid & 7, rg = tid >> 3;
  const int b = blockIdx.z, c = blockIdx.x * 32 + cg * 4;
  const int len = a.lens ? a.lens[b] : a.L;
  const int r0 = blockIdx.y * rows_per_split;
  const int r1 = min(len, r0 + rows_per_split);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  if (c < a.C) {
    for (int l = r0 + rg; l < r1; l += 32) {
      const float4 v = *(const float4*)(xb + (int64_t)l * a.ldx + c);
      const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double d = (double)t[i]; s[i] += d; q[i] += d * d; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][rg][cg * 4 + i] = s[i]; red[1][rg][cg * 4 + i] = q[i]; }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, ch = tid & 31;
    double t = 0;
    for (int r = 0; r < 32; ++r) t += red[which][r][ch];
    const int cc = blockIdx.x * 32 + ch;
    if (cc < a.C && r0 < len) atomicAdd(a.sums + ((int64_t)b * a.C + cc) * 2 + which, t);
  }
}

__global__ void adain_finalize_kernel(const mi355_adain_coef_args a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= a.out_ld) return;
  float sc = 0.f, sh = 0.f;
  if (c < a.C) {
    const int len = a.lens ? a.lens[b] : a.L;
    const double n = (double)(len > 0 ? len : 1);
    const double mean = a.sums[((int64_t)b * a.C + c) * 2] / n;
    double var = a.sums[((int64_t)b * a.C + c) * 2 + 1] / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = 1.0f / sqrtf((float)var + a.eps);
    float g = 0.f, be = 0.f;
    if (a.gb) { g = a.gb[(int64_t)b * a.gb_ld + c]; be = a.gb[(int64_t)b * a.gb_ld + a.C + c]; }
    sc = (1.0f + g) * rstd;
    sh = be - (float)mean * sc;
  }
  a.scale[(int64_t)b * a.out_ld + c] = sc;
  a.shift[(int64_t)b * a.out_ld + c] = sh;
}

// AdaIN coefficients from the (sum, M2) block partials a conv_gemm epilogue wrote.  256 threads = CPW channels x (256 / CPW) block lanes, float64
// throughout, two sweeps over the partials (they are L2-resident: 8 bytes per block and channel): first the total sum -> the mean, then
// M2 = sum_e [M2_e + cnt_e (mean_e - mean)^2] -- the exact decomposition of the sum of squared deviations over row blocks; no division inside
// the sweeps.  The sweeps are latency-bound (one 8-byte read per block, blocks C * 8 bytes apart), so a workgroup takes only CPW = 4 channels and
// spreads their blocks over 64 lanes each: 8 dependent rounds for the 495 blocks of a 31 681-row utterance instead of 31 with 16 lanes per
// channel (17.7 -> 7 us at one utterance).  grid (ceil(out_ld / CPW), B).
// Batches (B >= 8) are bandwidth problems instead (64 utterances: 32 MB of partials per call): CPW = 16 -- the 16 channel lanes of a block row read 128
// contiguous bytes, a full line per request instead of a quarter -- and ONE sweep: a lane keeps its <= kAdainKeep partials in registers between the mean
// and the M2 passes (longer utterances fall back to the second read).  Round 4: 28.8 -> us per call in the 64-utterance step (profiles/r4_*).
constexpr int kAdainKeep = 32;
template <int CPW>
__global__ __launch_bounds__(256) void adain_from_partials_kernel(const mi355_adain_partials_args a) {
  constexpr int LPC = 256 / CPW;   // block lanes per channel
  __shared__ double red[CPW][LPC + 1];
  const int cl = threadIdx.x % CPW, eg = threadIdx.x / CPW;
  const int c = blockIdx.x * CPW + cl, b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const int nblk = (len + MI355_STATS_ROWS - 1) / MI355_STATS_ROWS;
  const bool cok = c < a.C;
  const float* pb = a.partials + (int64_t)b * a.bstride + (int64_t)(cok ? c : 0) * 2;
  const int64_t estride = (int64_t)a.C * 2;
  const bool keep = CPW > 4 && nblk <= kAdainKeep * LPC;   // wave-uniform (per utterance)
  float2 kept[CPW > 4 ? kAdainKeep : 1];
  double s = 0.0;
  if (cok) {
    if (keep) {
#pragma unroll
      for (int j = 0; j < (CPW > 4 ? kAdainKeep : 1); ++j) {
        // unconditional (clamped) loads, zeroed by a select: as `e < nblk ? load : 0` each of the 32 compiled to a branch with its own
        // s_waitcnt vmcnt(0) -- 32 SERIAL round trips per lane (tools/scan_serial_waits.py)
        const int e = eg + j * LPC;
        const int ec = e < nblk ? e : (nblk > 0 ? nblk - 1 : 0);
        kept[j] = *(const float2*)(pb + (int64_t)ec * estride);
      }
#pragma unroll
      for (int j = 0; j < (CPW > 4 ? kAdainKeep : 1); ++j) {
        const bool in = eg + j * LPC < nblk;
        kept[j].x = in ? kept[j].x : 0.f;
        kept[j].y = in ? kept[j].y : 0.f;
      }
#pragma unroll
      for (int j = 0; j < (CPW > 4 ? kAdainKeep : 1); ++j) s += (double)kept[j].x;
    } else {   // two independent loads in flight per lane
      int e = eg;
      for (; e + LPC < nblk; e += 2 * LPC) s += (double)pb[(int64_t)e * estride] + (double)pb[(int64_t)(e + LPC) * estride];
      for (; e < nblk; e += LPC) s += (double)pb[(int64_t)e * estride];
    }
  }
  red[cl][eg] = s;
  __syncthreads();
  double tot = 0.0;
  for (int j = 0; j < LPC; ++j) tot += red[cl][j];   // every lane of a channel adds the shares in the same order: one value per channel
  const double mean = len > 0 ? tot / (double)len : 0.0;
  __syncthreads();
  double m2 = 0.0;
  if (cok) {
    const double inv_full = 1.0 / (double)MI355_STATS_ROWS;
    auto term = [&](const float2 sv, const int e) {
      const int cnt = min(MI355_STATS_ROWS, len - e * MI355_STATS_ROWS);
      const double me = cnt == MI355_STATS_ROWS ? (double)sv.x * inv_full : (double)sv.x / (double)cnt;
      const double d = me - mean;
      return (double)sv.y + d * d * (double)cnt;
    };
    if (keep) {
#pragma unroll
      for (int j = 0; j < (CPW > 4 ? kAdainKeep : 1); ++j) {
        const int e = eg + j * LPC;
        if (e < nblk) m2 += term(kept[j], e);
      }
    } else {
      int e = eg;
      for (; e + LPC < nblk; e += 2 * LPC) {
        const float2 v0 = *(const float2*)(pb + (int64_t)e * estride), v1 = *(const float2*)(pb + (int64_t)(e + LPC) * estride);
        m2 += term(v0, e) + term(v1, e + LPC);
      }
      for (; e < nblk; e += LPC) m2 += term(*(const float2*)(pb + (int64_t)e * estride), e);
    }
  }
  red[cl][eg] = m2;
  __syncthreads();
  if (eg != 0 || c >= a.out_ld) return;
  float sc = 0.f, sh = 0.f;
  if (cok) {
    double q = 0.0;
    for (int j = 0; j < LPC; ++j) q += red[cl][j];
    double var = len > 0 ? q / (double)len : 0.0;
    if (var < 0) var = 0;
    const float rstd = 1.0f / sqrtf((float)var + a.eps);
    float g = 0.f, be = 0.f;
    if (a.gb) { g = a.gb[(int64_t)b * a.gb_ld + c]; be = a.gb[(int64_t)b * a.gb_ld + a.C + c]; }
    sc = (1.0f + g) * rstd;
    sh = be - (float)mean * sc;
  }
  a.scale[(int64_t)b * a.out_ld + c] = sc;
  a.shift[(int64_t)b * a.out_ld + c] = sh;
}

// one wave per row, up to 1024 channels held in registers (two-pass mean / variance like mx.var)
// VEC: weight / bias / ada_gb rows are 16-byte aligned (the host checks): float4 operand loads; otherwise four scalar loads each
template <bool VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const mi355_layernorm_args a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * a.L) return;
  const int b = (int)(row / a.L), l = (int)(row - (int64_t)b * a.L);
  const int len = a.lens ? a.lens[b] : a.L;
  if (l >= len) return;
  const float* xr = a.x + (int64_t)b * a.x_bstride + (int64_t)l * a.ldx;
  const float* rr = a.res ? a.res + (int64_t)b * a.res_bstride + (int64_t)l * a.ldr : nullptr;
  float v[4][4];
  float s = 0.f;
  // the per-channel operands are requested WITH the row (16-byte loads), not after the two reductions: behind them they were a second, dependent round trip in
  // every wave's life -- 202 -> ~110 us for Whisper's 96 000 x 768 rows, the time of a plain copy (tools/ln_bench.py)
  float4 w4[4], b4[4], g4[4], e4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + lane * 4;
    w4[i] = make_float4(1.f, 1.f, 1.f, 1.f); b4[i] = make_float4(0.f, 0.f, 0.f, 0.f); g4[i] = b4[i]; e4[i] = b4[i];
    if (c < a.C) {
      auto ld4 = [](const float* p) {
        if constexpr (VEC) return *(const float4*)p;
        else return make_float4(p[0], p[1], p[2], p[3]);
      };
      if (a.weight) w4[i] = ld4(a.weight + c);
      if (a.weight && a.bias) b4[i] = ld4(a.bias + c);
      if (a.ada_gb) {
        g4[i] = ld4(a.ada_gb + (int64_t)b * a.ada_ld + c);
        e4[i] = ld4(a.ada_gb + (int64_t)b * a.ada_ld + a.C + c);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < a.C) {
      float4 t = *(const float4*)(xr + c);
      if (rr) { const float4 r4 = *(const float4*)(rr + c); t.x += r4.x; t.y += r4.y; t.z += r4.z; t.w += r4.w; }
      v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      s += (t.x + t.y) + (t.z + t.w);
    } else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
  }
  const float mean = wave_sum_fast(s) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < a.C) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float var = wave_sum_fast(q) / (float)a.C;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)l * a.ldy;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < a.C) {
      float o[4];
      const float wv[4] = {w4[i].x, w4[i].y, w4[i].z, w4[i].w}, bv[4] = {b4[i].x, b4[i].y, b4[i].z, b4[i].w};
      const float gv[4] = {g4[i].x, g4[i].y, g4[i].z, g4[i].w}, ev[4] = {e4[i].x, e4[i].y, e4[i].z, e4[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = (v[i][j] - mean) * rstd;
        if (a.weight) t = t * wv[j] + bv[j];
        if (a.ada_gb) t = (1.0f + gv[j]) * t + ev[j];
        if (a.post_act == MI355_ACT_LEAKY) t = t > 0.f ? t : t * a.post_slope;
        o[j] = t;
      }
      if (a.y_split) *(uint4*)(yr + c) = make_uint4(split16_word(o[0], a.y_split), split16_word(o[1], a.y_split), split16_word(o[2], a.y_split), split16_word(o[3], a.y_split));
      else *(float4*)(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, uint32_t* __restrict__ y, const int64_t n4, const int fmt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 t = ((const float4*)x)[i];
    ((uint4*)y)[i] = make_uint4(split16_word(t.x, fmt), split16_word(t.y, fmt), split16_word(t.z, fmt), split16_word(t.w, fmt));
  }
}

}  // namespace

extern "C" int mi355_split16(const float* x, void* y, int64_t n, int32_t fmt, void* stream) {
  MI355_REQUIRE(x && y && n > 0, "split16: null tensor / empty");
  MI355_REQUIRE(fmt == 2 || fmt == 4, "split16: fmt must be 2 (bfloat16) or 4 (IEEE half)");
  MI355_REQUIRE(n % 4 == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0, "split16: n must be a multiple of 4 and the pointers 16-byte aligned");
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)(n4 + 255) / 256 > 8192u ? 8192u : (unsigned)((n4 + 255) / 256);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(split16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint32_t*)y, n4, (int)fmt);
  MI355_LAUNCH_CHECK("split16");
  return MI355_OK;
}

extern "C" int mi355_adain_coef(const mi355_adain_coef_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->sums && ap->scale && ap->shift, "adain_coef: null tensor");
  const mi355_adain_coef_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.C > 0 && a.L > 0, "adain_coef: bad shape");
  MI355_REQUIRE(a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.ldx >= ((a.C + 3) & ~3), "adain_coef: ldx must be a multiple of 4 and cover C rounded up to 4");
  MI355_REQUIRE(a.out_ld >= a.C, "adain_coef: out_ld < C");
  hipStream_t st = (hipStream_t)stream;
  if (!a.reuse_sums) {
    hipError_t e = hipMemsetAsync(a.sums, 0, sizeof(double) * 2 * (size_t)a.B * a.C, st);
    MI355_REQUIRE(e == hipSuccess, "adain_coef: memset failed: %s", hipGetErrorString(e));
    const int cblocks = (a.C + 31) / 32;
    int nsplit = (1024 + cblocks * a.B - 1) / (cblocks * a.B);
    const int max_split = (a.L + 63) / 64;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    const int rows = (a.L + nsplit - 1) / nsplit;
    MI355_CLEAR_ERROR();
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(cblocks, nsplit, a.B), dim3(256), 0, st, a, rows);
    MI355_LAUNCH_CHECK("instnorm_partial");
  }
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(adain_finalize_kernel, dim3((a.out_ld + 127) / 128, a.B), dim3(128), 0, st, a);
  MI355_LAUNCH_CHECK("adain_finalize");
  return MI355_OK;
}

extern "C" int mi355_adain_from_partials(const mi355_adain_partials_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->partials && ap->scale && ap->shift, "adain_from_partials: null tensor");
  const mi355_adain_partials_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.C > 0 && a.L > 0 && a.out_ld >= a.C, "adain_from_partials: bad shape");
  MI355_CLEAR_ERROR();
  MI355_REQUIRE(a.bstride % 2 == 0 && ((uintptr_t)a.partials) % 8 == 0, "adain_from_partials: partials must be 8-byte aligned");
  // (MI355_ADAIN_CPW=4 keeps the latency-shaped kernel at every batch size: A/B aid)
  static const int cpw_env = getenv("MI355_ADAIN_CPW") ? atoi(getenv("MI355_ADAIN_CPW")) : 0;
  if (a.B >= 8 && cpw_env != 4) hipLaunchKernelGGL(adain_from_partials_kernel<16>, dim3((a.out_ld + 15) / 16, a.B), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(adain_from_partials_kernel<4>, dim3((a.out_ld + 3) / 4, a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("adain_from_partials");
  return MI355_OK;
}

extern "C" int mi355_layernorm(const mi355_layernorm_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y, "layernorm: null tensor");
  const mi355_layernorm_args a = *ap;
  MI355_REQUIRE(a.C > 0 && a.C <= 1024 && a.C % 4 == 0, "layernorm: C must be a multiple of 4 and <= 1024 (got %d)", a.C);
  MI355_REQUIRE(a.ldx % 4 == 0 && a.ldy % 4 == 0 && a.x_bstride % 4 == 0 && a.y_bstride % 4 == 0, "layernorm: strides must be multiples of 4");
  MI355_REQUIRE(!a.res || (a.ldr % 4 == 0 && a.res_bstride % 4 == 0), "layernorm: residual strides must be multiples of 4");
  MI355_REQUIRE(a.y_split == 0 || a.y_split == 2 || a.y_split == 4, "layernorm: y_split must be 0, 2 or 4");
  const int64_t rows = (int64_t)a.B * a.L;
  MI355_CLEAR_ERROR();
  const bool vec = ((uintptr_t)a.weight % 16 == 0) && ((uintptr_t)a.bias % 16 == 0) && ((uintptr_t)a.ada_gb % 16 == 0) && (a.ada_ld % 4 == 0);
  if (vec) hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("layernorm");
  return MI355_OK;
}
