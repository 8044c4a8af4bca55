// Bidirectional LSTM recurrence as ONE persistent workgroup per (direction, utterance) (gfx950).
// Replaces the per-time-step Python loops of the reference's hand-rolled LSTM
// (tts/models/kokoro/modules.py:150-240).  4H threads, thread r owns gate row r of Wh; the bf16
// recurrent weights stay resident on the CU for the whole sequence: KREG columns in VGPRs, KLDS
// columns in LDS ([k/8][row][8] so that consecutive lanes read consecutive 16 B), the rest streamed
// from L2 with coalesced 16-B loads.  h lives in LDS (fp32) and is broadcast-read; c lives in the
// registers of the first H threads.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

// 16-bit weight pair -> two floats: bf16 is the high half of a float; F16 = IEEE half (float32 checkpoints, precision 4)
template <bool F16>
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  if constexpr (F16) {
    lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
    hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
  } else {
    lo = __builtin_bit_cast(float, w << 16);
    hi = __builtin_bit_cast(float, w & 0xffff0000u);
  }
}

template <int H, int KREG, int KLDS, bool F16>
__global__ __launch_bounds__(4 * H) void lstm_kernel(const mi355_lstm_args a) {
  constexpr int G = 4 * H;
  constexpr int NREG = KREG / 8, NLDS = KLDS / 8, NGLB = (H - KREG - KLDS) / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wl = (uint4*)smem;                              // [NLDS][G] 16-B weight groups
  float* hbuf = (float*)(smem + (size_t)NLDS * G * 16);  // [H]
  float* gates = hbuf + H;                               // [G]
  __shared__ float wext[2 * ((H + 63) / 64)];            // quant_h: per-wave {-min, max} of the step's hidden vector
  const int r = threadIdx.x, dir = blockIdx.x, b = blockIdx.y;
  const int len = a.lens ? a.lens[b] : a.L;
  const uint4* wg = (const uint4*)a.wh + (size_t)dir * (H / 8) * G;  // [H/8][G]
  uint4 wreg[NREG > 0 ? NREG : 1];
#pragma unroll
  for (int i = 0; i < NREG; ++i) wreg[i] = wg[(size_t)i * G + r];
#pragma unroll
  for (int i = 0; i < NLDS; ++i) wl[i * G + r] = wg[(size_t)(NREG + i) * G + r];
  if (r < H) hbuf[r] = 0.f;
  float c = 0.f;
  __syncthreads();
  const float* xpb = a.xp + (int64_t)b * a.xp_bstride + (size_t)dir * G + r;
  float* ob = a.out + (int64_t)b * a.out_bstride + dir * H;
  const int gate = r / H;
  const float wsc = (F16 && a.wh_scale != 0.f) ? a.wh_scale : 1.0f;   // power of two: exact

  auto dot8 = [&](const uint4 w, const float* h8, float acc) {
    const float4 h0 = *(const float4*)(h8);
    const float4 h1 = *(const float4*)(h8 + 4);
    float w0, w1;
    unpack2<F16>(w.x, w0, w1); acc = fmaf(w0, h0.x, acc); acc = fmaf(w1, h0.y, acc);
    unpack2<F16>(w.y, w0, w1); acc = fmaf(w0, h0.z, acc); acc = fmaf(w1, h0.w, acc);
    unpack2<F16>(w.z, w0, w1); acc = fmaf(w0, h1.x, acc); acc = fmaf(w1, h1.y, acc);
    unpack2<F16>(w.w, w0, w1); acc = fmaf(w0, h1.z, acc); acc = fmaf(w1, h1.w, acc);
    return acc;
  };

  for (int s = 0; s < len; ++s) {
    const int t = dir ? (len - 1 - s) : s;
    const float xpv = xpb[(int64_t)t * a.ldxp];
    uint4 wglb[NGLB > 0 ? NGLB : 1];
#pragma unroll
    for (int i = 0; i < NGLB; ++i) wglb[i] = wg[(size_t)(NREG + NLDS + i) * G + r];
    float acc0 = 0.f, acc1 = 0.f;
    // keep the packed bf16 weights opaque per step: otherwise LICM hoists the UNPACKED fp32 copies
    // out of the time loop and doubles the register footprint (spills)
#pragma unroll
    for (int i = 0; i < NREG; ++i)
      asm volatile("" : "+v"(wreg[i].x), "+v"(wreg[i].y), "+v"(wreg[i].z), "+v"(wreg[i].w));
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      if (i & 1) acc1 = dot8(wreg[i], hbuf + i * 8, acc1);
      else acc0 = dot8(wreg[i], hbuf + i * 8, acc0);
    }
#pragma unroll
    for (int i = 0; i < NLDS; ++i) {
      const uint4 w = wl[i * G + r];
      if (i & 1) acc1 = dot8(w, hbuf + (NREG + i) * 8, acc1);
      else acc0 = dot8(w, hbuf + (NREG + i) * 8, acc0);
    }
#pragma unroll
    for (int i = 0; i < NGLB; ++i) {
      if (i & 1) acc1 = dot8(wglb[i], hbuf + (NREG + NLDS + i) * 8, acc1);
      else acc0 = dot8(wglb[i], hbuf + (NREG + NLDS + i) * 8, acc0);
    }
    const float pre = xpv + (acc0 + acc1) * wsc;
    float gv;
    if (gate == 2) gv = tanhf(pre);
    else gv = 1.0f / (1.0f + expf(-pre));
    gates[r] = gv;
    __syncthreads();
    float h = 0.f;
    if (r < H) {
      const float ig = gates[r], fg = gates[H + r], gg = gates[2 * H + r], og = gates[3 * H + r];
      c = fg * c + ig * gg;
      h = og * tanhf(c);
      ob[(int64_t)t * a.ldo + r] = h;
      if (!a.quant_h) hbuf[r] = h;
    }
    if (a.quant_h) {  // the next step's recurrent product sees fq(h): extrema over the H values of this utterance and direction
      constexpr int NWV = (H + 63) / 64;
      if (r < NWV * 64) {
        const float nmn = wave_max(fmaxf(-h, 0.f)), mx = wave_max(fmaxf(h, 0.f));  // lanes >= H carry h = 0
        if ((r & 63) == 0) { wext[2 * (r >> 6)] = nmn; wext[2 * (r >> 6) + 1] = mx; }
      }
      __syncthreads();
      if (r < H) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { a0 = fmaxf(a0, wext[2 * w]); a1 = fmaxf(a1, wext[2 * w + 1]); }
        const FakeQuant fq(-a0, a1);
        hbuf[r] = fq(h);
      }
    }
    __syncthreads();
  }
}

template <int H, int KREG, int KLDS, bool F16>
int launch_lstm_t(const mi355_lstm_args& a, hipStream_t st) {
  constexpr int G = 4 * H;
  const size_t lds = (size_t)(KLDS / 8) * G * 16 + (size_t)H * 4 + (size_t)G * 4;
  hipError_t e = hipFuncSetAttribute((const void*)lstm_kernel<H, KREG, KLDS, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  MI355_REQUIRE(e == hipSuccess, "lstm: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((lstm_kernel<H, KREG, KLDS, F16>), dim3(2, a.B), dim3(G), lds, st, a);
  MI355_LAUNCH_CHECK("lstm_bidir");
  return MI355_OK;
}

template <int H, int KREG, int KLDS>
int launch_lstm(const mi355_lstm_args& a, hipStream_t st) {
  return a.wh_f16 ? launch_lstm_t<H, KREG, KLDS, true>(a, st) : launch_lstm_t<H, KREG, KLDS, false>(a, st);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// lstm_oct_kernel: the same recurrence with the work of a step laid out so that the hidden vector is read from LDS 8x less.
//
// lstm_kernel above gives thread r ONE gate row: every one of the 4H threads reads all H values of h per step -- 4H x H x 4 B = 1 MB through
// the CU's LDS pipe (128 B / clk) = 8192 clk = 3.4 us per step at H = 256, which is exactly what it measures (profiles/
// r4_kernel_stats_b64_call21.txt: 398 us per launch); the 262 144 FMAs of a step are 2048 clk.  Here a thread owns EIGHT gate rows (i, f, g, o of
// two adjacent hidden units) over ONE EIGHTH of k (slice s = lane & 7): it reads H / 8 values of h (each of the 8 lanes of an octet another
// slice: 8 addresses per ds_read_b128, slices at a pitch of H / 8 + 4 floats so that they fall on disjoint banks), does the same 256 FMAs,
// and the octet all-reduces its 8 partial sums with three DPP steps (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror: every lane ends with
// every sum, identical on all eight).  Lane (quad q, gate g) then activates gate g of unit q, the quad broadcasts its four gates
// (quad_perm [n,n,n,n]) and every lane of the quad advances the same c / h: no gate exchange through LDS and ONE barrier per step (h is
// double-buffered).  Weights: the same packed image ([H/8][4H] 16-byte groups); 24 groups per thread in registers, the rest in LDS.
// Which of a thread's NW 16-byte weight groups (index = k-group i * 8 + row8) live in LDS instead of registers: none up to H = 128 (NW <= 16);
// at H = 256 (NW = 32) rows 6 and 7 of every k-group plus row 5 of the last = 9 groups x 16 KB, 23 in registers.  Spread over the k-groups so
// that only two or three LDS weight reads are in flight at a time (their landing registers are what the 128-register budget lacks).
constexpr bool oct_in_lds(int idx, int NW) { return NW > 16 && ((idx & 7) >= 6 || ((idx >> 3) == 3 && (idx & 7) == 5)); }
constexpr int oct_slot(int idx, int NW) {   // position among the groups of the same class
  int n = 0;
  for (int j = 0; j < idx; ++j) n += oct_in_lds(j, NW) == oct_in_lds(idx, NW);
  return n;
}
constexpr int oct_lds_groups(int NW) {
  int n = 0;
  for (int j = 0; j < NW; ++j) n += oct_in_lds(j, NW);
  return n;
}
template <int H, bool F16>
__global__ __launch_bounds__(4 * H) void lstm_oct_kernel(const mi355_lstm_args a) {
  constexpr int G = 4 * H, SL = H / 8, NG = SL / 8;   // k per slice, 16-byte weight groups per row and slice
  static_assert(NG >= 1, "H >= 64");
  constexpr int NW = 8 * NG;                          // weight groups per thread: index = i * 8 + row8 (k-group i of the slice, row8 = unit * 4 + gate)
  constexpr int NLDS = oct_lds_groups(NW), NREG = NW - NLDS;
  constexpr int HP = SL + 4;                          // slice pitch of h in floats: 8 slices on disjoint 4-bank groups
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wl = (uint4*)smem;                                   // [NLDS][G]
  float* hbuf = (float*)(smem + (size_t)NLDS * G * 16);       // [2][8 * HP]
  __shared__ float wext[2 * (G / 64)];                        // quant_h: per-wave {-min, max}
  const int t = threadIdx.x, dir = blockIdx.x, b = blockIdx.y;
  const int s = t & 7, u = t >> 3, q = s >> 2, g = s & 3;
  const int jq = 2 * u + q;                                   // the hidden unit this lane finishes
  const int len = a.lens ? a.lens[b] : a.L;
  const uint4* wg = (const uint4*)a.wh + (size_t)dir * (H / 8) * G;  // [H/8][G]
  auto wsrc = [&](const int idx) {   // weight group idx of this thread: k-group s * NG + i of row (gate gg of unit 2 u + uu)
    const int i = idx >> 3, r8 = idx & 7;
    return wg + (size_t)(s * NG + i) * G + ((r8 & 3) * H + 2 * u + (r8 >> 2));
  };
  uint4 wreg[NREG];
#pragma unroll
  for (int idx = 0; idx < NW; ++idx) {
    if (oct_in_lds(idx, NW)) wl[oct_slot(idx, NW) * G + t] = *wsrc(idx);
    else wreg[oct_slot(idx, NW)] = *wsrc(idx);
  }
  for (int i = t; i < 2 * 8 * HP; i += G) hbuf[i] = 0.f;
  float c = 0.f;
  __syncthreads();
  // wave-uniform row bases + one 32-bit lane offset each (64-bit per-lane pointers cost two registers apiece and were spilled)
  const float* const xrow = a.xp + (int64_t)b * a.xp_bstride + (size_t)dir * G;
  float* const orow = a.out + (int64_t)b * a.out_bstride + dir * H;
  const int xoff = g * H + jq;
  const float wsc = (F16 && a.wh_scale != 0.f) ? a.wh_scale : 1.0f;   // power of two: exact
  const int hdst = (jq / SL) * HP + (jq % SL);

  auto dot8 = [&](const uint4 w, const float4 h0, const float4 h1, float acc) {
    float w0, w1;
    unpack2<F16>(w.x, w0, w1); acc = fmaf(w0, h0.x, acc); acc = fmaf(w1, h0.y, acc);
    unpack2<F16>(w.y, w0, w1); acc = fmaf(w0, h0.z, acc); acc = fmaf(w1, h0.w, acc);
    unpack2<F16>(w.z, w0, w1); acc = fmaf(w0, h1.x, acc); acc = fmaf(w1, h1.y, acc);
    unpack2<F16>(w.w, w0, w1); acc = fmaf(w0, h1.z, acc); acc = fmaf(w1, h1.w, acc);
    return acc;
  };
  auto dpp = [](const float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };

  for (int st = 0; st < len; ++st) {
    const int tt = dir ? (len - 1 - st) : st;
    const float xpv = (xrow + (int64_t)tt * a.ldxp)[xoff];
    const float* hb = hbuf + (st & 1) * 8 * HP + s * HP;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    // keep the packed weights opaque per step (LICM would otherwise hoist the UNPACKED fp32 copies out of the time loop: spills)
#pragma unroll
    for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(wreg[i].x), "+v"(wreg[i].y), "+v"(wreg[i].z), "+v"(wreg[i].w));
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float4 h0 = *(const float4*)(hb + i * 8), h1 = *(const float4*)(hb + i * 8 + 4);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int idx = i * 8 + r;
        const uint4 w = oct_in_lds(idx, NW) ? wl[oct_slot(idx, NW) * G + t] : wreg[oct_in_lds(idx, NW) ? 0 : oct_slot(idx, NW)];
        acc[r] = dot8(w, h0, h1, acc[r]);
      }
      asm volatile("" ::: "memory");   // one k-group's LDS reads in flight at a time
    }
    // octet all-reduce: lane ^ 1, lane ^ 2, then the other quad (7 - lane); every lane ends with the same eight sums
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      acc[r] += dpp(acc[r], std::integral_constant<int, 0xB1>{});
      acc[r] += dpp(acc[r], std::integral_constant<int, 0x4E>{});
      acc[r] += dpp(acc[r], std::integral_constant<int, 0x141>{});
    }
    float mine = acc[0];
#pragma unroll
    for (int r = 1; r < 8; ++r) mine = s == r ? acc[r] : mine;
    const float pre = xpv + mine * wsc;
    // gate g of unit q: sigmoid, or tanh for g == 2 as 2 sigmoid(2 x) - 1 (one exponential for every lane of the wave, no divergence)
    const float m = g == 2 ? 2.0f : 1.0f;
    const float sg = 1.0f / (1.0f + expf(-m * pre));
    const float gv = g == 2 ? 2.0f * sg - 1.0f : sg;
    const float ig = dpp(gv, std::integral_constant<int, 0x00>{}), fg = dpp(gv, std::integral_constant<int, 0x55>{});
    const float gg = dpp(gv, std::integral_constant<int, 0xAA>{}), og = dpp(gv, std::integral_constant<int, 0xFF>{});
    c = fg * c + ig * gg;
    const float h = og * tanhf(c);
    float* hn = hbuf + ((st + 1) & 1) * 8 * HP;
    if (g == 0) (orow + (int64_t)tt * a.ldo)[jq] = h;
    if (!a.quant_h) {
      if (g == 0) hn[hdst] = h;
    } else {  // the next step's recurrent product sees fq(h): extrema over the H values of this utterance and direction
      const float nmn = wave_max(fmaxf(-h, 0.f)), mx = wave_max(fmaxf(h, 0.f));   // every unit appears in 4 lanes: harmless for an extremum
      if ((t & 63) == 0) { wext[2 * (t >> 6)] = nmn; wext[2 * (t >> 6) + 1] = mx; }
      __syncthreads();
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int w = 0; w < G / 64; ++w) { a0 = fmaxf(a0, wext[2 * w]); a1 = fmaxf(a1, wext[2 * w + 1]); }
      const FakeQuant fq(-a0, a1);
      if (g == 0) hn[hdst] = fq(h);
    }
    __syncthreads();
  }
}

template <int H, bool F16>
int launch_lstm_oct_t(const mi355_lstm_args& a, hipStream_t st) {
  constexpr int G = 4 * H, NW = 8 * (H / 64), NLDS = oct_lds_groups(NW);
  const size_t lds = (size_t)NLDS * G * 16 + (size_t)2 * 8 * (H / 8 + 4) * 4;
  hipError_t e = hipFuncSetAttribute((const void*)lstm_oct_kernel<H, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  MI355_REQUIRE(e == hipSuccess, "lstm: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((lstm_oct_kernel<H, F16>), dim3(2, a.B), dim3(G), lds, st, a);
  MI355_LAUNCH_CHECK("lstm_bidir");
  return MI355_OK;
}
template <int H>
int launch_lstm_oct(const mi355_lstm_args& a, hipStream_t st) {
  return a.wh_f16 ? launch_lstm_oct_t<H, true>(a, st) : launch_lstm_oct_t<H, false>(a, st);
}

}  // namespace

extern "C" int mi355_lstm_bidir(const mi355_lstm_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->xp && ap->wh && ap->out, "lstm: null tensor");
  const mi355_lstm_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0, "lstm: bad shape");
  MI355_REQUIRE(a.ldxp >= 8 * a.H && a.ldo >= 2 * a.H, "lstm: ldxp/ldo too small");
  hipStream_t st = (hipStream_t)stream;
  // MI355_LSTM_OCT=0: the one-gate-row-per-thread kernel for every size (A/B aid; read per call so that a test can flip it)
  const char* oct_env = getenv("MI355_LSTM_OCT");
  if (!(oct_env && oct_env[0] == '0')) {
    switch (a.H) {
      case 256: return launch_lstm_oct<256>(a, st);
      case 128: return launch_lstm_oct<128>(a, st);
      case 64: return launch_lstm_oct<64>(a, st);
    }
  }
  switch (a.H) {
    case 256: return launch_lstm<256, 192, 64>(a, st);
    case 128: return launch_lstm<128, 128, 0>(a, st);
    case 64: return launch_lstm<64, 64, 0>(a, st);
    case 32: return launch_lstm<32, 32, 0>(a, st);
  }
  mi355_set_error("lstm: unsupported hidden size %d (supported: 32, 64, 128, 256)", a.H);
  return MI355_ERR_UNSUPPORTED;
}
