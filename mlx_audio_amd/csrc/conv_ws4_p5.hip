// conv_ws4, precision 5 instantiations: fp16 hi pass + block-scaled e4m3 lo pass (v_mfma_scale_f32_32x32x64_f8f6f4) on the MX weight image
// (mi355_pack_conv_weight_mx_host).  Conv mode, 128-column tiles: the resblock / upsampler convs of the vocoders.
#include "conv_ws4.h"

using namespace mi355conv;

int mi355_conv_ws4_p5(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  const bool gemm = false;   // K == 1 layers run this kernel in conv mode too (one hi item + one half-empty lo item per chunk)
  if ((feat & 2) && pre == P_SNAKE && epi == 0) return launch_ws4<5, P_SNAKE, 0, false, false, 0, 128, false, false>(a, st, feat & 9);   // A / B aid: the 2 x 2 consumer layout
  WS4_CASE(5, P_NONE, 0);
  WS4_CASE(5, P_LEAKY, 0);
  WS4_CASE(5, P_SNAKE, 0);
  return MI355_ERR_UNSUPPORTED;
}
