// Extrema pass of the fake-quantised conv inputs (KittenTTS: tts/models/kitten_tts/quant.py:4-24): {-min, max} of the prologue's output per
// utterance, joined with 0.  Read-only over x: the quantised tensor itself is produced inside the consuming conv's prologue
// (mi355_conv_gemm_args.pre_fq), with the prologue value of conv_common.h's fq_pre_value in both places.
#include <algorithm>
#include "conv_common.h"

using namespace mi355conv;

namespace {

// 256 threads = rps rows x C4 channel quads (rps = 256 / C4 when a row is narrower than the workgroup); the channel coefficients of a thread's quad
// are loaded once
__global__ __launch_bounds__(256) void fq_extrema_kernel(const mi355_fake_quant_args a) {
  const int b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  const bool affine = a.pre_scale != nullptr;
  const int64_t poff = (int64_t)b * a.pre_ld;
  float nmn = 0.f, mx = 0.f;
  const int C4 = (a.C + 3) >> 2, rps = C4 >= 256 ? 1 : 256 / C4, span = C4 >= 256 ? 256 : C4;
  const int ty = C4 >= 256 ? 0 : (int)threadIdx.x / C4, tx = (int)threadIdx.x - ty * span;
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (a.x_bstride % 4 == 0) && (((uintptr_t)a.x) & 15) == 0;
  if (ty < rps) {
    for (int c4 = tx; c4 < C4; c4 += span) {
      fq_coef k[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) k[j] = fq_load_coef(a.pre_scale, a.pre_shift, poff, a.pre_act, a.pre_alpha, 4 * c4 + j < a.C ? 4 * c4 + j : a.C - 1);
      int l = blockIdx.x * rps + ty;
      const int lstep = gridDim.x * rps;
      if (vec) {
        // four rows' loads in flight per thread (unconditional: a row past the end repeats the last one -- a maximum does not care); as one
        // load per trip of a rolled loop the sweep ran at a fraction of the HBM rate
        for (; l < len; l += 4 * lstep) {
          float4 t[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int lu = l + u * lstep < len ? l + u * lstep : len - 1;
            t[u] = *(const float4*)(xb + (int64_t)lu * a.ldx + 4 * c4);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float v[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (4 * c4 + j < a.C) {
                const float tv = fq_pre_value(v[j], k[j], affine, a.pre_act, a.pre_slope);
                nmn = fmaxf(nmn, -tv);
                mx = fmaxf(mx, tv);
              }
            }
          }
        }
      }
      for (; l < len; l += lstep) {
        float v[4];
        if (vec) {
          const float4 t = *(const float4*)(xb + (int64_t)l * a.ldx + 4 * c4);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = 4 * c4 + j < a.C ? xb[(int64_t)l * a.ldx + 4 * c4 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (4 * c4 + j < a.C) {
            const float t = fq_pre_value(v[j], k[j], affine, a.pre_act, a.pre_slope);
            nmn = fmaxf(nmn, -t);
            mx = fmaxf(mx, t);
          }
        }
      }
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
    // both are >= 0: the integer order of their bit patterns is their float order, and max is order-independent (deterministic atomics)
    atomicMax((int*)a.minmax + 2 * b, __float_as_int(nmn));
    atomicMax((int*)a.minmax + 2 * b + 1, __float_as_int(mx));
  }
}

// {-min, max} of the prologue's output from per-block, per-channel (min, max) of its INPUT (the producing conv's ext_partial).  Grid (slices, B): a
// workgroup takes a slice of the row blocks; thread (tx = channel, ty = block row of the slice pass) keeps the extrema of ITS records, evaluates the
// prologue at both ends and joins: the maximum over partial ranges of f at their ends is f at the ends of the whole range when f rounds monotonically
// (and a bound of the same quality where Snake does not); slices meet in the same order-independent atomics as the sweep's.
__global__ __launch_bounds__(256) void fq_extrema_partials_kernel(const mi355_fake_quant_args a, const int blocks_per_slice) {
  const int b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const int nblk = (len + MI355_STATS_ROWS - 1) / MI355_STATS_ROWS;
  const int k0 = blockIdx.x * blocks_per_slice, k1 = min(nblk, k0 + blocks_per_slice);
  const float* pb = a.x + (int64_t)b * a.x_bstride;
  const bool affine = a.pre_scale != nullptr;
  const int64_t poff = (int64_t)b * a.pre_ld;
  const int CT = a.C >= 256 ? 256 : a.C, rows = 256 / CT;
  const int tx = (int)threadIdx.x % CT, ty = (int)threadIdx.x / CT;
  float nmn = 0.f, mx = 0.f;
  if (ty < rows && k0 < k1) {
    for (int c = tx; c < a.C; c += CT) {
      float lo = INFINITY, hi = -INFINITY;
      int k = k0 + ty;
      for (; k + 3 * rows < k1; k += 4 * rows) {   // four records in flight (unconditional inside the range)
        float2 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *(const float2*)(pb + ((int64_t)(k + u * rows) * a.C + c) * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) { lo = fminf(lo, r[u].x); hi = fmaxf(hi, r[u].y); }
      }
      for (; k < k1; k += rows) {
        const float2 r = *(const float2*)(pb + ((int64_t)k * a.C + c) * 2);
        lo = fminf(lo, r.x);
        hi = fmaxf(hi, r.y);
      }
      if (lo <= hi) {   // this thread saw at least one record
        const fq_coef kc = fq_load_coef(a.pre_scale, a.pre_shift, poff, a.pre_act, a.pre_alpha, c);
        const float t0 = fq_pre_value(lo, kc, affine, a.pre_act, a.pre_slope), t1 = fq_pre_value(hi, kc, affine, a.pre_act, a.pre_slope);
        nmn = fmaxf(nmn, fmaxf(-t0, -t1));
        mx = fmaxf(mx, fmaxf(t0, t1));
      }
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
    atomicMax((int*)a.minmax + 2 * b, __float_as_int(nmn));      // both >= 0: integer order of the bit patterns = float order
    atomicMax((int*)a.minmax + 2 * b + 1, __float_as_int(mx));
  }
}

}  // namespace

extern "C" int mi355_fake_quant_extrema_from_partials(const mi355_fake_quant_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->minmax, "fake_quant_extrema_from_partials: null tensor");
  const mi355_fake_quant_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0 && ((uintptr_t)a.x) % 8 == 0 && a.x_bstride % 2 == 0 &&
                a.x_bstride >= (int64_t)((a.L + MI355_STATS_ROWS - 1) / MI355_STATS_ROWS) * a.C * 2, "fake_quant_extrema_from_partials: bad shape");
  MI355_REQUIRE(!a.pre_scale == !a.pre_shift, "fake_quant_extrema_from_partials: pre_scale and pre_shift go together");
  MI355_REQUIRE(a.pre_act == MI355_ACT_NONE || a.pre_act == MI355_ACT_LEAKY || a.pre_act == MI355_ACT_SNAKE, "fake_quant_extrema_from_partials: unsupported prologue activation %d", a.pre_act);
  MI355_REQUIRE(a.pre_act != MI355_ACT_SNAKE || a.pre_alpha, "fake_quant_extrema_from_partials: snake needs pre_alpha");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a.minmax, 0, sizeof(float) * 2 * a.B, st);
  MI355_REQUIRE(e == hipSuccess, "fake_quant_extrema_from_partials: memset failed: %s", hipGetErrorString(e));
  const int nblk = (a.L + MI355_STATS_ROWS - 1) / MI355_STATS_ROWS;
  const int rows = a.C >= 256 ? 1 : 256 / a.C;
  // ~2048 workgroups over the batch, each at least four passes of its thread rows deep
  const int want = std::max(1, 2048 / std::max(1, a.B));
  const int per = std::max(4 * rows, (nblk + want - 1) / want);
  const int slices = (nblk + per - 1) / per;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(fq_extrema_partials_kernel, dim3(slices, a.B), dim3(256), 0, st, a, per);
  MI355_LAUNCH_CHECK("fake_quant_extrema_from_partials");
  return MI355_OK;
}

extern "C" int mi355_fake_quant_extrema(const mi355_fake_quant_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->minmax, "fake_quant_extrema: null tensor");
  const mi355_fake_quant_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0 && a.ldx >= a.C, "fake_quant_extrema: bad shape");
  MI355_REQUIRE(!a.pre_scale == !a.pre_shift, "fake_quant_extrema: pre_scale and pre_shift go together");
  MI355_REQUIRE(a.pre_act == MI355_ACT_NONE || a.pre_act == MI355_ACT_LEAKY || a.pre_act == MI355_ACT_SNAKE, "fake_quant_extrema: unsupported prologue activation %d", a.pre_act);
  MI355_REQUIRE(a.pre_act != MI355_ACT_SNAKE || a.pre_alpha, "fake_quant_extrema: snake needs pre_alpha");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(a.minmax, 0, sizeof(float) * 2 * a.B, st);
  MI355_REQUIRE(e == hipSuccess, "fake_quant_extrema: memset failed: %s", hipGetErrorString(e));
  const int c4 = (a.C + 3) / 4, rps = c4 >= 256 ? 1 : 256 / c4;
  // ~4096 workgroups over the batch, each sweeping several row groups when the tensor is large
  const unsigned nblk = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)a.L + 8 * rps - 1) / (8 * rps), 4096 / std::max(1, a.B) + 1));
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(fq_extrema_kernel, dim3(nblk, a.B), dim3(256), 0, st, a);
  MI355_LAUNCH_CHECK("fake_quant_extrema");
  return MI355_OK;
}
