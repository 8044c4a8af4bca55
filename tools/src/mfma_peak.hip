// What the matrix pipe of THIS box sustains, measured the way the conv kernel uses it (tools, not product; VERDICT r2 item 2 "or prove the ceiling").
// One kernel per MFMA shape: every wave holds NACC independent accumulators and issues back-to-back MFMAs on register-resident operands for
// `iters` rounds -- no loads, no LDS, no barriers in the loop, 2 waves per SIMD.  A wave's matrix pipe is then busy every cycle, so
//   issued cycles per SIMD  = waves_per_simd * iters * NACC * cycles_per_mfma        (32x32x16 16-bit: 8 passes x 4 = 32 cycles)
//   sustained shader clock  = issued cycles / elapsed time
// Operands: random 16-bit patterns (the activity factor of real data), or all zero (MI355_MFMA_ZERO=1): a power-limited part clocks the
// second one higher, which is the evidence that the 2.5 PFLOP/s dense peak (2.4 GHz) is a power/clock figure, not an issue-rate one.
// Usage: mfma_peak [ms_target ...]   -> one JSON line per (shape, duration).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  uint4 ra[2], rb[2];
  ra[0] = src[(t * 4 + 0) & 65535];
  ra[1] = src[(t * 4 + 1) & 65535];
  rb[0] = src[(t * 4 + 2) & 65535];
  rb[1] = src[(t * 4 + 3) & 65535];
  float total = 0.f;
  if constexpr (SHAPE == 0 || SHAPE == 2) {  // 32x32x16 bf16 / f16
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (SHAPE == 0)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i & 1]), __builtin_bit_cast(bf16x8, rb[(i >> 1) & 1]), acc[i], 0, 0, 0);
        else
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[i & 1]), __builtin_bit_cast(f16x8, rb[(i >> 1) & 1]), acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) total += acc[i][j];
  } else if constexpr (SHAPE == 3) {  // 32x32x64 fp8 (e4m3) on the block-scaled pipe with unit scales (E8M0 127): the only 2x-rate fp8 form of gfx950
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    i32x8 a8[2], b8[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a8[0][j] = ((const int*)&ra[0])[j]; a8[0][j + 4] = ((const int*)&ra[1])[j];
      a8[1][j] = ((const int*)&ra[1])[j]; a8[1][j + 4] = ((const int*)&rb[0])[j];
      b8[0][j] = ((const int*)&rb[0])[j]; b8[0][j + 4] = ((const int*)&rb[1])[j];
      b8[1][j] = ((const int*)&rb[1])[j]; b8[1][j + 4] = ((const int*)&ra[0])[j];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) total += acc[i][j];
  } else if constexpr (SHAPE == 5) {  // 32x32x64 fp4 (e2m1) on the block-scaled pipe with unit scales: 16 bytes per operand and lane
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    i32x8 a8[2], b8[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a8[0][j] = ((const int*)&ra[0])[j]; a8[0][j + 4] = 0;
      a8[1][j] = ((const int*)&ra[1])[j]; a8[1][j + 4] = 0;
      b8[0][j] = ((const int*)&rb[0])[j]; b8[0][j + 4] = 0;
      b8[1][j] = ((const int*)&rb[1])[j]; b8[1][j + 4] = 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) total += acc[i][j];
  } else if constexpr (SHAPE == 4) {  // 32x32x32 int8
    i32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, ra[i & 1]), __builtin_bit_cast(i32x4, rb[(i >> 1) & 1]), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) total += (float)acc[i][j];
  } else {  // 16x16x32 bf16
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[i & 1]), __builtin_bit_cast(bf16x8, rb[(i >> 1) & 1]), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) total += acc[i][j];
  }
  if (total == 12345.678f) sink[t] = total;  // keeps the loop; practically never taken
}

struct Shape { const char* name; int id; double flop_per_mfma; int cycles; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const bool zero = getenv("MI355_MFMA_ZERO") && atoi(getenv("MI355_MFMA_ZERO"));
  uint4* src;
  float* sink;
  const size_t nsrc = 65536;
  hipMalloc(&src, nsrc * sizeof(uint4));
  hipMalloc(&sink, (size_t)cus * 4 * 256 * sizeof(float) * 4);
  uint16_t* h = (uint16_t*)malloc(nsrc * 16);
  uint32_t s = 12345u;
  for (size_t i = 0; i < nsrc * 8; ++i) {
    s = s * 1664525u + 1013904223u;
    // a random finite 16-bit float of modest magnitude in either format: sign random, exponent field near the bias, mantissa random
    const uint16_t sign = (s >> 31) << 15;
    const uint16_t bf = sign | (uint16_t)((120 + ((s >> 8) & 7)) << 7) | (uint16_t)((s >> 12) & 0x7f);
    h[i] = zero ? 0 : bf;
  }
  hipMemcpy(src, h, nsrc * 16, hipMemcpyHostToDevice);
  // a second operand pool of random BYTES for the 8-bit shapes: finite e4m3 values of mid-range exponent (never the NaN pattern 0x7f), which as int8 are
  // ordinary signed values
  uint4* src8;
  hipMalloc(&src8, nsrc * sizeof(uint4));
  uint8_t* h8 = (uint8_t*)malloc(nsrc * 16);
  for (size_t i = 0; i < nsrc * 16; ++i) {
    s = s * 1664525u + 1013904223u;
    h8[i] = zero ? 0 : (uint8_t)(((s >> 31) << 7) | ((5 + ((s >> 9) % 6)) << 3) | ((s >> 13) & 7));
  }
  hipMemcpy(src8, h8, nsrc * 16, hipMemcpyHostToDevice);
  // cycles of the two 8-bit shapes = what a 2x-rate pipe would need (K = 64 fp8: 64; K = 32 int8: 32); "implied_clock_ghz" for them is that assumption's clock
  const Shape shapes[6] = {{"v_mfma_f32_32x32x16_bf16", 0, 2.0 * 32 * 32 * 16, 32},
                           {"v_mfma_f32_16x16x32_bf16", 1, 2.0 * 16 * 16 * 32, 16},
                           {"v_mfma_f32_32x32x16_f16", 2, 2.0 * 32 * 32 * 16, 32},
                           {"v_mfma_scale_f32_32x32x64_f8f6f4(e4m3, unit scales)", 3, 2.0 * 32 * 32 * 64, 64},
                           {"v_mfma_i32_32x32x32_i8", 4, 2.0 * 32 * 32 * 32, 32},
                           {"v_mfma_scale_f32_32x32x64_f8f6f4(fp4 e2m1, unit scales)", 5, 2.0 * 32 * 32 * 64, 32}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int wg_per_cu = 2;  // 2 workgroups x 4 waves = 8 waves per CU = 2 per SIMD with 4 independent accumulators each (the conv kernel's occupancy)
  const dim3 grid(cus * wg_per_cu), block(256);
  constexpr int NACC = 4;
  int ntargets = argc > 1 ? argc - 1 : 3;
  double targets_default[3] = {2.0, 50.0, 1000.0};
  const int only = getenv("MI355_MFMA_ONLY") ? atoi(getenv("MI355_MFMA_ONLY")) : -1;   // one shape id only
  for (int si = 0; si < 6; ++si) {
    if (only >= 0 && si != only && !(only == 35 && (si == 3 || si == 5 || si == 2))) continue;
    for (int ti = 0; ti < ntargets; ++ti) {
      const double target_ms = argc > 1 ? atof(argv[1 + ti]) : targets_default[ti];
      // iterations for the target duration at a nominal 2.0 GHz: per SIMD, waves x iters x NACC x cycles
      int iters = (int)(target_ms * 1e-3 * 2.0e9 / ((double)wg_per_cu * NACC * shapes[si].cycles));
      if (iters < 1) iters = 1;
      float best = 1e30f, last = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        if (si == 0) mfma_loop<0, NACC><<<grid, block>>>(src, sink, iters);
        if (si == 1) mfma_loop<1, NACC><<<grid, block>>>(src, sink, iters);
        if (si == 2) mfma_loop<2, NACC><<<grid, block>>>(src, sink, iters);
        if (si == 3) mfma_loop<3, NACC><<<grid, block>>>(src8, sink, iters);
        if (si == 4) mfma_loop<4, NACC><<<grid, block>>>(src8, sink, iters);
        if (si == 5) mfma_loop<5, NACC><<<grid, block>>>(src8, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
      }
      const double mfmas = (double)cus * wg_per_cu * 4.0 * iters * NACC;
      const double tflops = mfmas * shapes[si].flop_per_mfma / (last * 1e-3) / 1e12;
      const double cycles_per_simd = (double)wg_per_cu * iters * NACC * shapes[si].cycles;
      printf("{\"shape\": \"%s\", \"operands\": \"%s\", \"cus\": %d, \"waves_per_simd\": 2, \"target_ms\": %.1f, \"ms_last_of_3\": %.3f, \"ms_best\": %.3f, "
             "\"tflops_last\": %.1f, \"frac_of_2500\": %.3f, \"implied_clock_ghz\": %.3f}\n",
             shapes[si].name, zero ? "zero" : "random", cus, target_ms, last, best, tflops, tflops / 2500.0, cycles_per_simd / (last * 1e-3) / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
