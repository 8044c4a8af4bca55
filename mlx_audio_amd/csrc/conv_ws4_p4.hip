// conv_ws4, fp16 hi+lo (precision 4) instantiations: fp16 checkpoints (Whisper), float32 codec checkpoints held as fp16 images (Vocos, DAC, SNAC).
#include "conv_ws4.h"

using namespace mi355conv;

int mi355_conv_ws4_p4(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  const bool gemm = gemm_mode(a);
  WS4_CASE(4, P_NONE, 0);
  WS4_CASE(4, P_LEAKY, 0);
  WS4_CASE(4, P_SNAKE, 0);
  WS4_CASE(4, P_SNAKEBETA, 0);
  WS4_CASE(4, P_ELU, 0);
  WS4_CASE(4, P_NONE, 1);
  WS4_GEMM(4, 0);
  WS4_GEMM(4, 1);
  WS4_GEMM(4, 2);
  WS4_GEMM(4, 3);
  // 64-column tiles: the late, thin stages of the codec decoders (DAC / SNAC: Snake, Vocos / EnCodec: ELU / none)
  WS4_CASE_N64(4, P_NONE, 0);
  WS4_CASE_N64(4, P_LEAKY, 0);
  WS4_CASE_N64(4, P_SNAKE, 0);
  WS4_CASE_N64(4, P_ELU, 0);
  WS4_GEMM_N64(4, 0);
  return MI355_ERR_UNSUPPORTED;
}
