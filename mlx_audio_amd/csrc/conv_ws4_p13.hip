// conv_ws4, single-pass instantiations (precision 1: bf16, precision 3: fp16): the opt-in fast modes of the Kokoro / Whisper engines.
#include "conv_ws4.h"

using namespace mi355conv;

int mi355_conv_ws4_p13(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  const bool gemm = gemm_mode(a);
  if (a.precision == 1) {
    WS4_CASE(1, P_NONE, 0);
    WS4_CASE(1, P_LEAKY, 0);
    WS4_CASE(1, P_SNAKE, 0);
    WS4_GEMM(1, 0);
    WS4_GEMM(1, 1);
    WS4_CASE_N64(1, P_NONE, 0);
    WS4_CASE_N64(1, P_LEAKY, 0);
    WS4_CASE_N64(1, P_SNAKE, 0);
    WS4_GEMM_N64(1, 0);
  } else {
    WS4_CASE(3, P_NONE, 0);
    WS4_CASE(3, P_LEAKY, 0);
    WS4_CASE(3, P_SNAKE, 0);
    WS4_CASE(3, P_NONE, 1);
    WS4_GEMM(3, 0);
    WS4_GEMM(3, 1);
    WS4_CASE_N64(3, P_NONE, 0);
    WS4_CASE_N64(3, P_LEAKY, 0);
    WS4_CASE_N64(3, P_SNAKE, 0);
    WS4_GEMM_N64(3, 0);
  }
  return MI355_ERR_UNSUPPORTED;
}
