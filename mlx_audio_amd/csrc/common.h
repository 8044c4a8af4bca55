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
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  return (uint32_t)f32_to_bf16_bits(a) | ((uint32_t)f32_to_bf16_bits(b) << 16);
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
