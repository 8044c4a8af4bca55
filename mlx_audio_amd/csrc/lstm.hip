// Bidirectional LSTM recurrence as ONE persistent workgroup per (direction, utterance) (gfx950).
// Replaces the per-time-step Python loops of the reference's hand-rolled LSTM
// (tts/models/kokoro/modules.py:150-240).  4H threads, thread r owns gate row r of Wh; the bf16
// recurrent weights stay resident on the CU for the whole sequence: KREG columns in VGPRs, KLDS
// columns in LDS ([k/8][row][8] so that consecutive lanes read consecutive 16 B), the rest streamed
// from L2 with coalesced 16-B loads.  h lives in LDS (fp32) and is broadcast-read; c lives in the
// registers of the first H threads.
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

}  // namespace

extern "C" int mi355_lstm_bidir(const mi355_lstm_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->xp && ap->wh && ap->out, "lstm: null tensor");
  const mi355_lstm_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0, "lstm: bad shape");
  MI355_REQUIRE(a.ldxp >= 8 * a.H && a.ldo >= 2 * a.H, "lstm: ldxp/ldo too small");
  hipStream_t st = (hipStream_t)stream;
  switch (a.H) {
    case 256: return launch_lstm<256, 192, 64>(a, st);
    case 128: return launch_lstm<128, 128, 0>(a, st);
    case 64: return launch_lstm<64, 64, 0>(a, st);
    case 32: return launch_lstm<32, 32, 0>(a, st);
  }
  mi355_set_error("lstm: unsupported hidden size %d (supported: 32, 64, 128, 256)", a.H);
  return MI355_ERR_UNSUPPORTED;
}
