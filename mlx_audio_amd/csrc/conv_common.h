// Device helpers shared by the conv_gemm translation units (conv_gemm.hip: 4-wave kernels + dispatcher,
// conv_ws4.hip: the wave-specialised producer / consumer kernel).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace mi355conv {

__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// nn.GELU(approx="tanh") / nn.gelu_approx: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh(float x) { return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); }

// one 32x32x16 MFMA on 16-byte A / B fragments: bf16 (PREC 1, 2) or fp16 (PREC 3, 4) inputs, fp32 accumulate
// PREC 2 / 4 split the fp32 activation into hi + lo images of the weight's 16-bit type (two MFMAs per fragment)
template <int PREC>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (PREC >= 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// number of LDS images of the activation window: hi + lo for the split, one otherwise
template <int PREC>
constexpr int a_images() { return (PREC == 2 || PREC == 4) ? 2 : 1; }
// the part of t the first (hi) image carries, as an fp32 value
template <int PREC>
__device__ __forceinline__ float split_hi(float t) {
  if constexpr (PREC == 3) return t;
  else if constexpr (PREC == 4) return (float)(_Float16)__builtin_fminf(__builtin_fmaxf(t, -65504.f), 65504.f);
  else return bf16_bits_to_f32(f32_to_bf16_bits(t));
}
template <int PREC>
__device__ __forceinline__ uint32_t pack_lo(float a, float b) {
  if constexpr (PREC == 4) return pack_f16x2(a, b);
  else return pack_bf16x2(a, b);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads / writes are done
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

}  // namespace mi355conv

// conv_ws4.hip: wave-specialised kernel (tile code 6128128 [+ 10000000 * feature bits for A/B runs]).  `feat` bit 0: consumers run at
// s_setprio 1; bit 1: the producers keep two activation windows in flight.  Returns MI355_ERR_UNSUPPORTED (and sets the error text) for
// argument combinations it has no instantiation for; `min_tiles_ok` tells the auto dispatcher whether the launch would fill the chip.
int mi355_conv_ws4_launch(const mi355_conv_gemm_args& a, hipStream_t st, int feat);
bool mi355_conv_ws4_eligible(const mi355_conv_gemm_args& a, bool vec);
