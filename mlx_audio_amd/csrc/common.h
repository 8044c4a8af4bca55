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

// fp32 -> SPLIT word (include/mi355audio.h: x_split / y_split): the 16-bit hi part in bits 0-15 and the 16-bit lo residual in bits 16-31, in IEEE half
// (fmt 4; the value clamped to +-65504 first) or bfloat16 (fmt 2) -- the two numbers the conv prologue of that precision makes of the value
__device__ __forceinline__ uint32_t split16_word(float v, int fmt) {
  if (fmt == 4) {
    const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    const _Float16 h = (_Float16)c;
    const _Float16 l = (_Float16)(c - (float)h);
    return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
  }
  const uint16_t h = f32_to_bf16_bits(v);
  const uint16_t l = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
  return (uint32_t)h | ((uint32_t)l << 16);
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

// ---------------------------------------------------------------------------------------------- weight element types of the GEMV images
// A lane's 16-byte piece of a weight row holds 8 sixteen-bit elements (bf16 / fp16) or 16 fp8 elements (OCP e4m3fn, no inf, max 448).
template <int WT> struct mi355_wt { static constexpr int EPL = 8; };
template <> struct mi355_wt<MI355_W_FP8> { static constexpr int EPL = 16; };
constexpr float kFp8Unbias = 256.0f;  // see cvt_w16<MI355_W_FP8>: the decoded value is the e4m3 value / 2^8

// 16 bytes -> EPL floats.  fp8: a byte s eeee mmm is moved into binary16 position (s 0eeee mmm0000000): the 4-bit exponent lands in the low
// bits of the 5-bit field and subnormals stay subnormals, so v_cvt_f32_f16 decodes every finite e4m3 code exactly, scaled by
// 2^(7 - 15) = 2^-8; the caller folds 2^8 into the per-row scale (a power-of-two multiply: exact).
template <int WT>
__device__ __forceinline__ void cvt_w16(const uint4 w, float (&f)[mi355_wt<WT>::EPL]) {
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (WT == MI355_W_FP8) {
      const uint32_t x01 = ((u[i] << 8) & 0x0000ff00u) | ((u[i] << 16) & 0xff000000u);
      const uint32_t x23 = ((u[i] >> 8) & 0x0000ff00u) | (u[i] & 0xff000000u);
      const uint32_t h01 = (x01 & 0x80008000u) | ((x01 & 0x7f007f00u) >> 1);
      const uint32_t h23 = (x23 & 0x80008000u) | ((x23 & 0x7f007f00u) >> 1);
      f[4 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h01 & 0xffffu));
      f[4 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h01 >> 16));
      f[4 * i + 2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h23 & 0xffffu));
      f[4 * i + 3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(h23 >> 16));
    } else if constexpr (WT == MI355_W_F16) {
      f[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u[i] & 0xffffu));
      f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u[i] >> 16));
    } else {
      f[2 * i] = __builtin_bit_cast(float, u[i] << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
    }
  }
}

// host: fp32 -> OCP e4m3fn bits (bias 7, 3 mantissa bits, subnormal quantum 2^-9, max finite 448, no inf), round-to-nearest-even,
// saturating at +-448; NaN -> 0x7f.  (What torch.float8_e4m3fn's conversion produces for every |x| <= 448.)
static inline uint8_t host_f32_to_e4m3(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return (uint8_t)(sign | 0x7fu);
  float ax;
  __builtin_memcpy(&ax, &u, 4);
  if (ax >= 448.0f) return (uint8_t)(sign | 0x7eu);
  int e = (int)(u >> 23) - 127;
  if (e < -6) e = -6;                                   // subnormal range: fixed quantum 2^-9
  const float q = __builtin_ldexpf(1.0f, e - 3);        // spacing of the e4m3 grid around ax
  const float n = __builtin_rintf(ax / q);              // exact scaling, round-to-nearest-even (default rounding mode)
  int ni = (int)n;                                      // 0..16: 8..15 = normal mantissas, 16 = carry into the next binade
  if (ni == 0) return sign;
  if (ni >= 16) { ni = 8; e += 1; }
  if (ni < 8) return (uint8_t)(sign | (uint8_t)ni);     // subnormal (only reachable with e == -6)
  if (e > 8) return (uint8_t)(sign | 0x7eu);
  return (uint8_t)(sign | (uint8_t)(((e + 7) << 3) | (ni - 8)));
}

// host: fp32 -> OCP e2m1 (FP4) code: sign, 2 exponent bits (bias 1), 1 mantissa bit: magnitudes {0, 0.5, 1, 1.5, 2, 3, 4, 6}; round to nearest
// (ties to the even CODE, as v_cvt_scalef32_pk_fp4_f32 does: probed, profiles/r6_mfma_fp4_probe_call2.jsonl), saturating at +-6; NaN -> +6.
static inline uint8_t host_f32_to_e2m1(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  const uint8_t sign = (uint8_t)((u >> 28) & 0x8u);
  u &= 0x7fffffffu;
  float ax;
  __builtin_memcpy(&ax, &u, 4);
  if (!(ax < 6.0f)) return (uint8_t)(sign | 7u);
  static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  int c = 0;
  while (c < 7 && ax > mag[c + 1]) ++c;                 // mag[c] < ax <= mag[c + 1]  (c == 0: 0 <= ax <= 0.5)
  if (ax <= mag[c]) return (uint8_t)(sign | (uint8_t)c);
  const float mid = 0.5f * (mag[c] + mag[c + 1]);
  const int pick = ax < mid ? c : (ax > mid ? c + 1 : ((c & 1) ? c + 1 : c));   // tie: the even code
  return (uint8_t)(sign | (uint8_t)pick);
}

// one value into a buffer of element type MI355_KV_F32 / MI355_KV_BF16 / MI355_KV_F16 (index in elements); round to nearest even
__device__ __forceinline__ void store_kv_elem(void* base, int64_t idx, float v, int dtype) {
  if (dtype == MI355_KV_F32) ((float*)base)[idx] = v;
  else if (dtype == MI355_KV_BF16) ((uint16_t*)base)[idx] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
  else ((uint16_t*)base)[idx] = (uint16_t)(pack_f16x2(v, 0.f) & 0xffffu);
}

// Dynamic per-tensor uint8 fake quantisation (tts/models/kitten_tts/quant.py:4-20), float32 op by op: mn <= 0 <= mx are the tensor's
// extrema joined with 0.  scale == 0 (constant-zero tensor) -> every output is 0.
struct FakeQuant {
  float scale, zp;
  __device__ __forceinline__ FakeQuant(float mn, float mx) {
    scale = (mx - mn) / 255.0f;
    const float safe = scale == 0.f ? 1.0f : scale;
    zp = fminf(fmaxf(rintf(-mn / safe), 0.0f), 255.0f);
  }
  __device__ __forceinline__ float operator()(float x) const {
    if (scale == 0.f) return 0.f;
    const float q = fminf(fmaxf(rintf(x / scale + zp), 0.0f), 255.0f);
    return (q - zp) * scale;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Wave sum / max on the DPP path (4 row-local steps + 4 readlanes, ~50 cycles) instead of six ds_bpermute round trips (~600 cycles): for code
// that runs with ALL 64 lanes active (readlane of an inactive lane is undefined).  The sum's association differs from wave_sum's.
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2, 3, 0, 1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror: every lane = its 16-lane row's sum
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
// The rotary pair and the per-head sum of squares with their roundings PINNED (no fp contraction): three kernels apply them (head_norm_rope_kernel,
// the fused decode-attention prologue, the q|k|v GEMV epilogue) and the tests hold their cache rows bit-equal -- with contraction left to the compiler
// the choice of which product joins the fma followed the code around the expression (it changed when the operands' loads were moved, round 4).
// (x * cos) + (rotate(x) * sin): each term rounded like the reference's separate multiplies and add.
__device__ __forceinline__ void rope_pair(const float x0, const float x1, const float c, const float s, float& y0, float& y1) {
#pragma clang fp contract(off)
  const float p0 = x0 * c, p1 = x1 * s, p2 = x1 * c, p3 = x0 * s;
  y0 = p0 - p1;
  y1 = p2 + p3;
}
__device__ __forceinline__ float sumsq2(const float x0, const float x1) {
#pragma clang fp contract(off)
  const float p0 = x0 * x0, p1 = x1 * x1;
  return p0 + p1;
}
// wave_max on the same DPP path (all 64 lanes active)
__device__ __forceinline__ float wave_max_fast(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
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
