// conv_ws4, precision 6 instantiations: fp16 hi pass + block-scaled FP4 (e2m1) lo pass (v_mfma_scale_f32_32x32x64_f8f6f4 at 4x the 16-bit rate) on the
// MX4 weight image (mi355_pack_conv_weight_mx4_host).  Conv mode, 128-column tiles, column-wave consumers: the resblock / upsampler convs of the vocoders.
#include "conv_ws4.h"

using namespace mi355conv;

int mi355_conv_ws4_p6(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  const bool gemm = false;   // K == 1 layers run this kernel in conv mode too (one hi item + one half-empty lo item per chunk)
  WS4_CASE(6, P_NONE, 0);
  WS4_CASE(6, P_LEAKY, 0);
  WS4_CASE(6, P_SNAKE, 0);
  return MI355_ERR_UNSUPPORTED;
}
