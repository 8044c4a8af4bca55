// Shared helpers for the gfx950 kernels of libmi355audio.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mi355audio.h"

void mi355_set_error(const char* fmt, ...);

#define MI355_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      mi355_set_error(__VA_ARGS__);         \
      return MI355_ERR_ARG;                 \
    }                                       \
  } while (0)

// hipGetLastError() is sticky per thread: clear whatever an unrelated earlier runtime call left behind
// before launching, so MI355_LAUNCH_CHECK reports only this launch.
#define MI355_CLEAR_ERROR() (void)hipGetLastError()

#define MI355_LAUNCH_CHECK(name)                                          \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      mi355_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return MI355_ERR_LAUNCH;                                            \
    }                                                                     \
  } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// round-to-nearest-even fp32 -> bf16 bits (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
typedef __bf16 mi355_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mi355_f32x2 __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32 (round-to-nearest-even, a in the low half)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const mi355_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mi355_bf16x2));
}

// round-to-nearest-even fp32 -> fp16 bits, saturating at +-65504 (the conv prologue never produces inf on purpose)
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  a = __builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f);
  b = __builtin_fminf(__builtin_fmaxf(b, -65504.f), 65504.f);
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}

// host: fp32 -> IEEE binary16 bits, round-to-nearest-even, saturating to the largest finite value
static inline uint16_t host_f32_to_f16(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);           // NaN
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);          // >= 65520 rounds past the largest finite: saturate
  if (u < 0x33000001u) return (uint16_t)sign;                       // < 2^-25 (or exactly): rounds to zero
  const int e = (int)(u >> 23) - 127;
  uint32_t m = (u & 0x7fffffu) | 0x800000u;                         // 24-bit significand
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;                    // bits dropped (subnormal: more)
  uint32_t half = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1u))) ++half;
  if (e < -14) return (uint16_t)(sign | half);                      // subnormal (a carry lands in the exponent correctly)
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));
}

static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
