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
  if constexpr (PREC >= 3)   // (3, 4: fp16 images; 5, 6: the fp16 hi pass of the MX images)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// number of 16-bit LDS images of the activation window: hi + lo for the 16-bit splits, one otherwise (PREC 5 keeps its lo residual as an
// 8-bit block-scaled plane behind the hi image: window_bytes)
template <int PREC>
constexpr int a_images() { return (PREC == 2 || PREC == 4) ? 2 : 1; }
// PREC 5 = fp16 hi pass + MX lo pass: the lo residual t - fp16(t) as OCP e4m3 bytes with one E8M0 scale byte per window row and 32-channel chunk,
// multiplied by a second, e4m3 image of the weights (one E8M0 scale per output column) on v_mfma_scale_f32_32x32x64_f8f6f4, which pairs two taps
// per instruction (K block b of the instruction = the 32 channels of tap 2 p + b) and runs at twice the 16-bit rate.
// LDS bytes of one staged window of R rows: [hi image R x 80][lo image R x 48][R scale bytes, padded to 16] (padded pitches instead of a swizzle)
template <int PREC>
__host__ __device__ constexpr int window_bytes(const int R) {
  return PREC == 5 ? R * 128 + ((R + 15) & ~15)    // PREC 5: 80-byte hi rows + 48-byte lo rows (unswizzled, conv_ws4.h)
         : PREC == 6 ? R * 96 + ((R + 15) & ~15)   // PREC 6: 80-byte hi rows + 16-byte FP4 lo rows
                     : a_images<PREC>() * R * 64;
}
// the part of t the first (hi) image carries, as an fp32 value
template <int PREC>
__device__ __forceinline__ float split_hi(float t) {
  if constexpr (PREC == 3) return t;
  else if constexpr (PREC == 4 || PREC == 5 || PREC == 6) return (float)(_Float16)__builtin_fminf(__builtin_fmaxf(t, -65504.f), 65504.f);
  else return bf16_bits_to_f32(f32_to_bf16_bits(t));
}
template <int PREC>
__device__ __forceinline__ uint32_t pack_lo(float a, float b) {
  if constexpr (PREC == 4) return pack_f16x2(a, b);
  else return pack_bf16x2(a, b);
}

// ---- fake-quantised conv inputs (KittenTTS: tts/models/kitten_tts/quant.py:4-24, istftnet.py:131) -------------------------------------------
// The module input that is quantised is the PROLOGUE'S OUTPUT t = act(scale x + shift).  The extrema pass (conv_quant.hip) and every conv
// prologue that applies the quantiser evaluate t with this one function, operation by operation (explicit fmaf, v_sin on revolutions, v_rcp +
// one Newton step: TU-independent, nothing left for the compiler to contract), so the extrema are the extrema of exactly the values the conv
// quantises.  al_rev = alpha / 2 pi, ial = 1 / alpha.
struct fq_coef { float sc, sh, al_rev, ial; };
__device__ __forceinline__ fq_coef fq_load_coef(const float* pre_scale, const float* pre_shift, const int64_t pre_off, const int pre_act,
                                                const float* pre_alpha, const int c) {
  fq_coef k{1.f, 0.f, 1.f, 1.f};
  if (pre_scale) { k.sc = pre_scale[pre_off + c]; k.sh = pre_shift[pre_off + c]; }
  if (pre_act == MI355_ACT_SNAKE) {
    const float al = pre_alpha[c];
    const float r0 = __builtin_amdgcn_rcpf(al);
    k.ial = fmaf(fmaf(-al, r0, 1.0f), r0, r0);
    k.al_rev = al * 0.15915494309189535f;
  }
  return k;
}
__device__ __forceinline__ float fq_pre_value(const float x, const fq_coef& k, const bool affine, const int pre_act, const float slope) {
  float u = affine ? fmaf(x, k.sc, k.sh) : x;
  if (pre_act == MI355_ACT_LEAKY) {
    const float m = u * slope;
    u = u > 0.f ? u : m;
  } else if (pre_act == MI355_ACT_SNAKE) {
    const float sn = __builtin_amdgcn_sinf(k.al_rev * u);
    u = fmaf(k.ial, sn * sn, u);
  }
  return u;
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads / writes are done
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Shared epilogue: y = ((acc + bias -> act) + res + y_old) * out_scale, plain or polyphase (conv_transpose) store.
// All loads of one 32x32 fragment are issued back to back on clamped addresses and only the stores are predicated:
// a per-element "if (valid) v += res[...]" makes hipcc branch around every load and wait vmcnt(0) each time
// (64 dependent round trips per wave).  `folded`: res / y_old were already added into the accumulators.
// EPI: which epilogue activation this instantiation carries: -1 = all of them behind runtime branches (the 4-wave kernels), 0 = none /
// LeakyReLU, 1 = GELU (erf), 2 = SiLU, 3 = GELU-tanh.  The wave-specialised kernel is instantiated per activation: every activation body is
// inlined once per accumulator element (~14 KB of code each), and the instantiation that carried all of them was 2x the code of the plain one
// and did not fit the instruction cache (profiles/r1_static_code_size_conv_gemm.txt; now profiles/r2_static_code_size_conv.txt).
// EXT (the quantising-prologue instantiations of the wave-specialised kernel only): per 64-row block and channel the (min, max) of the STORED output
// go to a.ext_partial beside the statistics -- the producer of a fake-quantised conv's input hands over what the consumer's extrema pass would
// otherwise re-read the whole tensor for (every prologue in use is monotone per channel: mi355_fake_quant_extrema_from_partials).
// MT / MO: the caller's accumulator array holds MT row blocks and this call works on blocks MO .. MO + MF - 1 as its blocks 0 .. MF - 1 (the column
// waves of the precision-5 kernel run the epilogue once per 64-row statistics block).
template <int MF, int NF, int WM, int WN, int EPI, bool EXT = false, int MT = MF, int MO = 0>
__device__ __forceinline__ void conv_epilogue(const mi355_conv_gemm_args& a, f32x16 (&acc)[MT][NF], const int b, const int l0,
                                              const int n0, const int wm, const int wn, const int lane, const int len_out,
                                              const bool folded) {
  const int len_up = a.lens_up ? a.lens_up[b] : a.up_Lout;
  const int len_up_c = len_up > 0 ? len_up : 1;
  float* yb = a.y + (int64_t)b * a.y_bstride;
  const float* rb = (a.res && !folded) ? a.res + (int64_t)b * a.res_bstride : nullptr;
  const bool accum = a.accumulate && !folded;
  const bool want_stats = a.stats_partial != nullptr;
  // fused statistics (single pass, shifted by the lane's first stored value K so that s2 - s1^2/n does not cancel)
  float sK[NF], s1[NF], s2[NF];
  int scnt[NF];
  const bool want_ext = EXT && a.ext_partial != nullptr;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) { sK[nf] = 0.f; s1[nf] = 0.f; s2[nf] = 0.f; scnt[nf] = 0; }
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int n = n0 + wn * WN + nf * 32 + (lane & 31);
      const bool nok = n < a.Cout;
      const int ncl = nok ? n : a.Cout - 1;
      int ocol = ncl, rph = 0;
      if (a.up_s) { rph = ncl / a.up_cout; ocol = ncl - rph * a.up_cout; }
      const float bias = a.bias ? a.bias[ocol] : 0.f;
      const float cscale = a.post_colscale ? a.post_colscale[ocol] : 1.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // 8 accumulator rows at a time keeps the live set small
        int orows[8];
        bool ok[8];
        float rv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = h * 8 + q;
          const int u = l0 + wm * WM + mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          bool valid = nok && u < len_out;
          int orow = u < len_out ? u : len_out - 1;
          if (a.up_s) {
            int nc = orow * a.up_s + rph - a.up_p;
            valid = valid && nc >= 0 && nc < len_up;
            nc = nc < 0 ? 0 : (nc >= len_up_c ? len_up_c - 1 : nc);
            orow = nc + a.up_row_off;
          }
          ok[q] = valid;
          orows[q] = orow;
          rv[q] = 0.f;
        }
        if (rb) {
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] = rb[(int64_t)(orows[q] >> a.res_shift) * a.ldr + ocol];
        }
        if (accum) {
          float yv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) yv[q] = yb[(int64_t)orows[q] * a.ldy + ocol];
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] += yv[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = acc[MO + mf][nf][h * 8 + q] + bias;
          if constexpr (EPI == -1 || EPI == 0) {
            if (a.post_act == MI355_ACT_LEAKY) v = v > 0.f ? v : v * a.post_slope;
          }
          if constexpr (EPI == -1) {
            if (a.post_act == MI355_ACT_GELU) v = gelu_erf(v);
          } else if constexpr (EPI == 1) {
            v = gelu_erf(v);
          }
          if constexpr (EPI == -1) {
            if (a.post_act == MI355_ACT_SILU) v = v / (1.0f + expf(-v));
            else if (a.post_act == MI355_ACT_GELU_TANH) v = gelu_tanh(v);
            else if (a.post_act == MI355_ACT_ELU) v = v > 0.f ? v : expm1f(v);
            else if (a.post_act == MI355_ACT_TANH) v = tanhf(v);
          } else if constexpr (EPI == 2) {
            v = v / (1.0f + expf(-v));
          } else if constexpr (EPI == 3) {
            v = gelu_tanh(v);
          }
          v = (v * cscale + rv[q]) * a.out_scale;
          if (ok[q]) {
            if (a.y_split) ((uint32_t*)yb)[(int64_t)orows[q] * a.ldy + ocol] = split16_word(v, a.y_split);   // for the launch that consumes y (x_split)
            else yb[(int64_t)orows[q] * a.ldy + ocol] = v;
          }
          if (want_stats && ok[q]) {
            sK[nf] = scnt[nf] == 0 ? v : sK[nf];
            const float d = v - sK[nf];
            s1[nf] += d;
            s2[nf] += d * d;
            ++scnt[nf];
          }

        }
      }
    }
  // Fused instance-norm statistics of what was just stored: this wave's WM = 64 rows x 32 columns per nf.  Lanes l and
  // l + 32 hold the same column (rows interleaved): their (count, mean, M2) triples are merged with Chan's formula after
  // one xor-32 exchange.  (sum, M2 about the block mean) per column goes to stats_partial[b][row block][n]; the float64
  // merge over row blocks happens in adain_from_partials.
  if constexpr (WM == MI355_STATS_ROWS) {
    if (want_stats && a.up_s == 0) {
      const int row0 = l0 + wm * WM;
      if (row0 < len_out) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int n = n0 + wn * WN + nf * 32 + (lane & 31);
          const float cl = (float)scnt[nf];
          const float ml = scnt[nf] ? sK[nf] + s1[nf] / cl : 0.f;
          const float vl = scnt[nf] ? s2[nf] - s1[nf] * s1[nf] / cl : 0.f;
          const float cp = __shfl_xor(cl, 32, 64), mp = __shfl_xor(ml, 32, 64), vp = __shfl_xor(vl, 32, 64);
          const float ct = cl + cp;
          const float dm = mp - ml;
          const float sum = ml * cl + mp * cp;
          const float m2 = vl + vp + (ct > 0.f ? dm * dm * cl * cp / ct : 0.f);
          if (lane < 32 && n < a.Cout)
            *(float2*)(a.stats_partial + (int64_t)b * a.stats_bstride + ((int64_t)(row0 / MI355_STATS_ROWS) * a.Cout + n) * 2) = make_float2(sum, m2);
        }
      }
    }
    if constexpr (EXT) {
      // edge tiles only (interior tiles take conv_epilogue_interior): the extrema from the values this lane just stored, read back (a thread sees its
      // own stores) -- whatever epilogue variant produced them, and nothing of the store loop stays live
      if (want_ext && a.up_s == 0) {
        const int row0 = l0 + wm * WM;
        if (row0 < len_out) {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int n = n0 + wn * WN + nf * 32 + (lane & 31);
            const int ncl = n < a.Cout ? n : a.Cout - 1;
            float mn = INFINITY, mx = -INFINITY;
#pragma unroll 1   // (rolled: edge tiles only -- eight loads in flight per trip are plenty, and the kernel's register budget stays the main loop's)
            for (int g8 = 0; g8 < MF * 2; ++g8) {
              float v[8];
              int us[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int r = (g8 & 1) * 8 + q;
                us[q] = row0 + (g8 >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                v[q] = yb[(int64_t)(us[q] < len_out ? us[q] : len_out - 1) * a.ldy + ncl];
              }
#pragma unroll
              for (int q = 0; q < 8; ++q)
                if (us[q] < len_out) { mn = fminf(mn, v[q]); mx = fmaxf(mx, v[q]); }
            }
            mn = fminf(mn, __shfl_xor(mn, 32, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (lane < 32 && n < a.Cout)
              *(float2*)(a.ext_partial + (int64_t)b * a.ext_bstride + ((int64_t)(row0 / MI355_STATS_ROWS) * a.Cout + n) * 2) = make_float2(mn, mx);
          }
        }
      }
    }
  }
}

// Interior-tile epilogue of the wave-specialised kernel: every row and column of the wave's 64 x 64 block is in range, plain store, no
// epilogue activation, residual / running sum already folded into the accumulators (or absent).  Same arithmetic, in the same order, as
// conv_epilogue (bias, out_scale, shifted single-pass statistics), so the two paths are bit-identical; what goes away is the per-element
// clamping, predication and 64-bit address arithmetic: one uniform base pointer + a 32-bit lane offset per store.
template <int MF, int NF, int WM, int WN, bool EXT = false, int MT = MF, int MO = 0>
__device__ __forceinline__ void conv_epilogue_interior(const mi355_conv_gemm_args& a, f32x16 (&acc)[MT][NF], const int b, const int l0,
                                                       const int n0, const int wm, const int wn, const int lane) {
  char* yw = (char*)(a.y + (int64_t)b * a.y_bstride + (int64_t)(l0 + wm * WM) * a.ldy + (n0 + wn * WN));  // wave-uniform base
  const uint32_t ldb = (uint32_t)a.ldy * 4u;                                                             // row pitch in bytes
  const uint32_t lane_off = (uint32_t)(4 * (lane >> 5)) * ldb + (uint32_t)(lane & 31) * 4u;            // < 4 GB inside the wave block
  const bool want_stats = a.stats_partial != nullptr;
  const float oscale = a.out_scale;
  float sK[NF], s1[NF], s2[NF];
  const bool want_ext = EXT && a.ext_partial != nullptr;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) { sK[nf] = 0.f; s1[nf] = 0.f; s2[nf] = 0.f; }
  auto body = [&](auto stats_tag) {
    constexpr bool STATS = decltype(stats_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float bias = a.bias ? a.bias[n0 + wn * WN + nf * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = mf * 32 + (r & 3) + 8 * (r >> 2);
          const float v = (acc[MO + mf][nf][r] + bias) * oscale;
          *(float*)(yw + (lane_off + row * ldb + (uint32_t)(nf * 128))) = v;
          if constexpr (STATS) {
            if (mf == 0 && r == 0) sK[nf] = v;
            const float d = v - sK[nf];
            s1[nf] += d;
            s2[nf] += d * d;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one 32 x 32 fragment at a time: keeps the stored values from piling up in registers
      }
  };
  if (want_stats) body(std::true_type{});
  else body(std::false_type{});
  if constexpr (WM == MI355_STATS_ROWS) {
    if (want_stats) {
      const int row0 = l0 + wm * WM;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int n = n0 + wn * WN + nf * 32 + (lane & 31);
        const float cl = (float)(MF * 16);
        const float ml = sK[nf] + s1[nf] / cl;
        const float vl = s2[nf] - s1[nf] * s1[nf] / cl;
        const float cp = cl, mp = __shfl_xor(ml, 32, 64), vp = __shfl_xor(vl, 32, 64);
        const float ct = cl + cp;
        const float dm = mp - ml;
        const float sum = ml * cl + mp * cp;
        const float m2 = vl + vp + dm * dm * cl * cp / ct;
        if (lane < 32)
          *(float2*)(a.stats_partial + (int64_t)b * a.stats_bstride + ((int64_t)(row0 / MI355_STATS_ROWS) * a.Cout + n) * 2) = make_float2(sum, m2);
      }
    }
    if constexpr (EXT) {
      // a second pass over the accumulators (the stored values recomputed: the same two operations), behind the stores and the statistics: nothing of
      // the store loop stays live here, so the kernel's register budget is the one it had
      if (want_ext) {
        const int row0 = l0 + wm * WM;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int n = n0 + wn * WN + nf * 32 + (lane & 31);
          const float bias = a.bias ? a.bias[n] : 0.f;
          float amn = INFINITY, amx = -INFINITY;   // extrema of the raw accumulators: v = (acc + bias) * oscale is monotone in acc (float32 add / multiply
#pragma unroll                                      // round monotonically), so the stored values' extrema are that expression at the two ends
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              amn = fminf(amn, acc[MO + mf][nf][r]);
              amx = fmaxf(amx, acc[MO + mf][nf][r]);
            }
          const float e0 = (amn + bias) * oscale, e1 = (amx + bias) * oscale;
          float mn = fminf(e0, e1), mx = fmaxf(e0, e1);
          mn = fminf(mn, __shfl_xor(mn, 32, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          if (lane < 32)
            *(float2*)(a.ext_partial + (int64_t)b * a.ext_bstride + ((int64_t)(row0 / MI355_STATS_ROWS) * a.Cout + n) * 2) = make_float2(mn, mx);
        }
      }
    }
  }
}

// Interior-tile epilogue of the POLYPHASE store (conv_transpose as a stride-1 conv with up_s * up_cout columns: column n = phase * up_cout + channel goes to
// output row u * up_s + phase - up_p): every GEMM row of the tile and every output row it maps to is in range, no epilogue activation, no column scale,
// no statistics.  Same arithmetic as conv_epilogue (bias, + residual at the OUTPUT row, out_scale), so the two paths are bit-identical; what goes away is
// the per-element division-free but 64-bit, predicated address arithmetic: one base pointer per 32-column fragment and a 32-bit row step
// (Kokoro's two upsamplers at 64 utterances: 1.32 + 0.79 ms of the step on the general path, profiles/r6_shape_table_b64_call18.txt).
template <int MF, int NF, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_up_interior(const mi355_conv_gemm_args& a, f32x16 (&acc)[MF][NF], const int b, const int l0, const int n0,
                                                          const int wm, const int wn, const int lane) {
  const float oscale = a.out_scale;
  // wave-uniform bases (the output row of the wave block's first GEMM row at phase 0) + one 32-bit lane offset per 32-column fragment
  const int64_t urow = (int64_t)(l0 + wm * WM) * a.up_s - a.up_p + a.up_row_off;
  char* yw = (char*)(a.y + (int64_t)b * a.y_bstride + urow * a.ldy);
  const char* rw = a.res ? (const char*)(a.res + (int64_t)b * a.res_bstride + urow * a.ldr) : nullptr;
  const uint32_t ystep = (uint32_t)a.up_s * (uint32_t)a.ldy * 4u;   // bytes between the output rows of consecutive GEMM rows (one phase)
  const uint32_t rstep = (uint32_t)a.up_s * (uint32_t)a.ldr * 4u;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n0 + wn * WN + nf * 32 + (lane & 31);
    const int rph = n / a.up_cout, ocol = n - rph * a.up_cout;
    const float bias = a.bias ? a.bias[ocol] : 0.f;
    const uint32_t lrow = (uint32_t)(4 * (lane >> 5));
    const uint32_t yoff = lrow * ystep + ((uint32_t)rph * (uint32_t)a.ldy + (uint32_t)ocol) * 4u;
    const uint32_t roff = lrow * rstep + ((uint32_t)rph * (uint32_t)a.ldr + (uint32_t)ocol) * 4u;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // 8 accumulator rows at a time keeps the live set small
        float rv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rv[q] = 0.f;
        if (rw) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            rv[q] = *(const float*)(rw + (roff + (uint32_t)(mf * 32 + (r & 3) + 8 * (r >> 2)) * rstep));
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = h * 8 + q;
          const float v = ((acc[mf][nf][r] + bias) + rv[q]) * oscale;
          *(float*)(yw + (yoff + (uint32_t)(mf * 32 + (r & 3) + 8 * (r >> 2)) * ystep)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
}

}  // namespace mi355conv

// conv_ws4.hip: wave-specialised kernel (tile code 6128128 [+ 10000000 * feature bits for A/B runs]).  `feat` bit 0: consumers at default
// priority (A/B aid; they run at s_setprio 1 otherwise), bit 2: timeline probe build, bit 3: one workgroup per tile instead of persistent
// workgroups, bits 4..: timing ablations.  Returns MI355_ERR_UNSUPPORTED (and sets the error text) for argument combinations it has no
// instantiation for.
int mi355_conv_ws4_launch(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn = 128);
bool mi355_conv_ws4_eligible(const mi355_conv_gemm_args& a, bool vec);
