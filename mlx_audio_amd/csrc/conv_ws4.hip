// conv_ws4: dispatcher + the bf16 hi+lo (precision 2) instantiations of the wave-specialised kernel (conv_ws4.h), its timing-ablation and
// timeline-probe builds.
#include "conv_ws4.h"

using namespace mi355conv;

namespace {
unsigned long long* g_dbg_buffer = nullptr;  // host copy of the probe buffer pointer (handed to the DBG instantiation through its geometry record)
}

namespace mi355conv {
int g_ws4_resident = 0;
}

// Process-wide, host-side setting read at launch time: persistent workgroups of the wave-specialised kernel's next launches (0 = the default, two
// per CU).  For running independent convs side by side on several streams, each on a share of the chip (tools/bench_conv_concurrent.py).
extern "C" int mi355_conv_ws4_resident(int wgs) {
  MI355_REQUIRE(wgs >= 0, "conv_ws4_resident: negative count");
  mi355conv::g_ws4_resident = wgs;
  return MI355_OK;
}

bool mi355_conv_ws4_eligible(const mi355_conv_gemm_args& a, bool vec) {
  if (!vec || a.Lin <= 0 || a.flat_valid != 0) return false;   // (flattened strided convs: the producers mask by row, not by flat element index)
  if (a.precision == 5 || a.precision == 6)   // conv mode only (K == 1 layers included), K odd with an even number of tap pairs, 128-column tiles
    return a.K % 4 == 3 && 128 + (a.K - 1) * a.dil <= 192 && a.Cout > 64 && !a.pre_fq && (pre_kind(a) == P_NONE || pre_kind(a) == P_LEAKY || pre_kind(a) == P_SNAKE) &&
           epi_family(a) == 0;
  if (!gemm_mode(a) && 128 + (a.K - 1) * a.dil > 192) return false;
  return a.precision >= 1 && a.precision <= 4 && pre_kind(a) >= 0;
}

// the timeline probe's buffer: [ceil(grid / 16)][34] uint64 (device memory owned by the caller; nullptr switches the probe off)
extern "C" int mi355_conv_ws4_debug_buffer(void* p) {
  g_dbg_buffer = (unsigned long long*)p;
  return MI355_OK;
}

int mi355_conv_ws4_p2(const mi355_conv_gemm_args& a, hipStream_t st, int feat, unsigned long long* dbg, int bn) {
  const int pre = pre_kind(a), epi = epi_family(a);
  const bool gemm = gemm_mode(a);
  if (bn == 128 && (feat & 4) && pre == P_SNAKE && epi == 0 && !gemm) return launch_ws4<2, P_SNAKE, 0, false, true>(a, st, feat, dbg);
  if (const int abl = bn == 128 ? feat >> 4 : 0) {  // timing ablations (wrong results by design)
    if (gemm && epi == 0) {   // GEMM mode (round 6: what bounds the wide linears)
      switch (abl) {
        case 1: return launch_ws4<2, P_NONE, 0, true, false, 1>(a, st, feat & 15);
        case 4: return launch_ws4<2, P_NONE, 0, true, false, 4>(a, st, feat & 15);
        case 5: return launch_ws4<2, P_NONE, 0, true, false, 5>(a, st, feat & 15);
      }
    }
    MI355_REQUIRE(pre == P_SNAKE && epi == 0 && !gemm, "conv_gemm(ws4): ablation tiles exist for the precision-2 Snake kernel and the plain GEMM mode only");
    switch (abl) {
      case 1: return launch_ws4<2, P_SNAKE, 0, false, false, 1>(a, st, feat & 15);
      case 2: return launch_ws4<2, P_SNAKE, 0, false, false, 2>(a, st, feat & 15);
      case 3: return launch_ws4<2, P_SNAKE, 0, false, false, 3>(a, st, feat & 15);
      case 4: return launch_ws4<2, P_SNAKE, 0, false, false, 4>(a, st, feat & 15);
      case 7: return launch_ws4<2, P_SNAKE, 0, false, false, 7>(a, st, feat & 15);
    }
    mi355_set_error("conv_gemm(ws4): unknown ablation %d", abl);
    return MI355_ERR_UNSUPPORTED;
  }
  // Kokoro: AdaIN resblocks (Snake), AdainResBlk1d / upsamplers (LeakyReLU), linears (+ GELU: PL-BERT FFN); bf16 codec checkpoints (Mimi: ELU,
  // Qwen3 codec: SnakeBeta, LayerScale / SiLU / GELU-tanh linears)
  WS4_CASE(2, P_NONE, 0);
  WS4_CASE(2, P_LEAKY, 0);
  WS4_CASE(2, P_SNAKE, 0);
  WS4_CASE(2, P_SNAKEBETA, 0);
  WS4_CASE(2, P_ELU, 0);
  WS4_CASE(2, P_NONE, 1);
  WS4_GEMM(2, 0);
  WS4_GEMM(2, 1);
  WS4_GEMM(2, 2);
  WS4_GEMM(2, 3);
  // 64-column tiles: thin tensors (KittenTTS: 64-channel generator stage; codec decoders' late stages)
  WS4_CASE_N64(2, P_NONE, 0);
  WS4_CASE_N64(2, P_LEAKY, 0);
  WS4_CASE_N64(2, P_SNAKE, 0);
  WS4_CASE_N64(2, P_ELU, 0);
  WS4_GEMM_N64(2, 0);
  return MI355_ERR_UNSUPPORTED;
}

int mi355_conv_ws4_launch(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn) {
  int rc = MI355_ERR_UNSUPPORTED;
  if (a.pre_fq) rc = a.precision == 2 ? mi355_conv_ws4_fq(a, st, feat & 3, bn) : MI355_ERR_UNSUPPORTED;
  else if (a.precision == 2) rc = mi355_conv_ws4_p2(a, st, feat, g_dbg_buffer, bn);
  else if (a.precision == 4) rc = mi355_conv_ws4_p4(a, st, feat & 3, bn);
  else if (a.precision == 6) rc = (feat & ~9) ? mi355_conv_ws4_p5_probe(a, st, feat, g_dbg_buffer) : mi355_conv_ws4_p6(a, st, feat & 9, bn);
  else if (a.precision == 5) rc = (feat & ~11) ? mi355_conv_ws4_p5_probe(a, st, feat, g_dbg_buffer) : mi355_conv_ws4_p5(a, st, feat & 11, bn);
  else if (a.precision == 1 || a.precision == 3) rc = mi355_conv_ws4_p13(a, st, feat & 3, bn);
  if (rc == MI355_ERR_UNSUPPORTED)
    mi355_set_error("conv_gemm(ws4): no instantiation for precision %d, prologue %d, epilogue family %d, %d-column tiles", a.precision, pre_kind(a), epi_family(a), bn);
  return rc;
}
