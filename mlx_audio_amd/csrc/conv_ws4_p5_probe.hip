// conv_ws4, precision 5: the timeline-probe (DBG) and timing-ablation (ABL) builds of the Snake instantiation -- measurement aids only (an
// ablation's results are wrong by design); reached through the explicit tile codes of tools/bench_conv.py --ablate --precision 5 and
// tools/conv_timeline.py --precision 5.  Their own translation unit: the production objects stay byte-identical.
#include "conv_ws4.h"

using namespace mi355conv;

int mi355_conv_ws4_p5_probe(const mi355_conv_gemm_args& a, hipStream_t st, int feat, unsigned long long* dbg) {
  const int pre = pre_kind(a), epi = epi_family(a);
  MI355_REQUIRE(pre == P_SNAKE && epi == 0, "conv_gemm(ws4): the precision-5 / 6 probe / ablation tiles exist for the Snake kernel only");
  if (a.precision == 6) {   // the FP4 lo pass (round 6)
    if (feat & 4) return launch_ws4<6, P_SNAKE, 0, false, true>(a, st, feat & 9, dbg);
    switch (feat >> 4) {
      case 1: return launch_ws4<6, P_SNAKE, 0, false, false, 1>(a, st, feat & 9);
      case 4: return launch_ws4<6, P_SNAKE, 0, false, false, 4>(a, st, feat & 9);
      case 5: return launch_ws4<6, P_SNAKE, 0, false, false, 5>(a, st, feat & 9);
    }
    mi355_set_error("conv_gemm(ws4): unknown precision-6 ablation %d", feat >> 4);
    return MI355_ERR_UNSUPPORTED;
  }
  if (feat & 4) return launch_ws4<5, P_SNAKE, 0, false, true>(a, st, feat & 11, dbg);
  switch (feat >> 4) {   // (the no-LDS-read builds spill under this register plan: not instantiated)
    case 1: return launch_ws4<5, P_SNAKE, 0, false, false, 1>(a, st, feat & 11);
    case 4: return launch_ws4<5, P_SNAKE, 0, false, false, 4>(a, st, feat & 11);
    case 5: return launch_ws4<5, P_SNAKE, 0, false, false, 5>(a, st, feat & 11);
  }
  mi355_set_error("conv_gemm(ws4): unknown precision-5 ablation %d", feat >> 4);
  return MI355_ERR_UNSUPPORTED;
}
