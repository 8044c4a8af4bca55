// conv_ws4, quantising prologues (mi355_conv_gemm_args.pre_fq: KittenTTS activation quantisation), bf16 hi+lo, 128- and 64-column tiles.
#include "conv_ws4.h"

using namespace mi355conv;

#define WS4_FQ(PRE, BNV) \
  if (bn == BNV && pre == PRE) return launch_ws4<2, PRE, 0, false, false, 0, BNV, true>(a, st, feat)

int mi355_conv_ws4_fq(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  if (epi != 0) return MI355_ERR_UNSUPPORTED;
  WS4_FQ(P_NONE, 128);
  WS4_FQ(P_LEAKY, 128);
  WS4_FQ(P_SNAKE, 128);
  WS4_FQ(P_NONE, 64);
  WS4_FQ(P_LEAKY, 64);
  WS4_FQ(P_SNAKE, 64);
  return MI355_ERR_UNSUPPORTED;
}
