// conv_gemm: conv1d / linear / polyphase conv_transpose1d as an implicit GEMM on the CDNA4 matrix
// cores (v_mfma_f32_32x32x16_bf16), written for gfx950 only.
//
// Data flow per 256-thread workgroup (4 waves, 2x2 over a BM x BN output tile):
//   for every 32-channel chunk of the input:
//     - the (BM + (K-1)*dil)-row activation window is read ONCE from HBM (fp32, channels-last, 128 B
//       per row per chunk, float4 per lane), the fused prologue (AdaIN affine, Snake / LeakyReLU) is
//       applied in registers, the value is split into bf16 hi + bf16 lo and written to LDS
//       (64-B rows, 16-B chunks XOR-swizzled by (row>>2)&3 so that the 32x32x16 A-fragment
//       ds_read_b128 is bank-conflict free for every tap shift);
//     - per tap, the pre-packed weight fragments (lane-linear MFMA B-fragment order) are streamed
//       L2 -> LDS with global_load_lds_dwordx4 one step ahead of the MFMAs (double buffer);
//     - A fragments for tap k are simply the window rows shifted by k*dil: no im2col is materialised,
//       and the transcendental prologue is evaluated once per input element, not once per tap.
//   epilogue: bias, activation, residual (optionally read at row>>1: nearest x2 shortcut), scale,
//   accumulate, and either a plain channels-last store or the polyphase conv_transpose scatter.
//
// Reference call sites replaced: see include/mi355audio.h (mi355_conv_gemm).
#include "conv_common.h"

using namespace mi355conv;

namespace {

constexpr int kThreads = 256;

template <int BM, int BN, int PREC, bool VEC>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(const mi355_conv_gemm_args a) {
  constexpr int WM = BM / 2, WN = BN / 2, MF = WM / 32, NF = WN / 32;
  constexpr int BBYTES = (BN / 32) * 2048;  // one (chunk, tap) weight slice of this block
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.z, l0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int len_out = a.lens_out ? a.lens_out[b] : a.Lout;
  if (l0 >= len_out) return;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Lin;
  const int K = a.K, dil = a.dil;
  const int R = BM + (K - 1) * dil;
  char* A_hi = smem;
  char* A_lo = smem + R * 64;
  char* Bs = smem + R * 64 * a_images<PREC>();
  const int nchunks = (a.Cin + 31) >> 5;
  const int NTp = ((a.Cout + 127) >> 7) << 2;
  const int nsteps = nchunks * K;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  const int64_t flat_hi = (int64_t)len_in * a.flat_valid;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto issue_B = [&](int step, int buf) {
    const char* src = (const char*)a.w + ((int64_t)step * NTp + (n0 >> 5)) * 2048;
#pragma unroll
    for (int i = 0; i < BN / 64; ++i) {
      const int off = (i * 4 + wave) * 1024;
      glds16(src + off + lane * 16, Bs + buf * BBYTES + off);
    }
  };

  auto stage_A = [&](int chunk) {
    const int c4 = (tid & 7) * 4;
    const int c = chunk * 32 + c4;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {1.f, 1.f, 1.f, 1.f},
          ial[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.pre_scale) {
      const float4 s4 = *(const float4*)(a.pre_scale + (int64_t)b * a.pre_ld + c);
      const float4 h4 = *(const float4*)(a.pre_shift + (int64_t)b * a.pre_ld + c);
      sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
      sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
    }
    if (a.pre_act == MI355_ACT_SNAKE) {
      const float4 a4 = *(const float4*)(a.pre_alpha + c);
      al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
#pragma unroll
      for (int i = 0; i < 4; ++i) ial[i] = 1.0f / al[i];
      if (a.pre_inv_beta) {  // SnakeBeta: x + sin^2(alpha x) * inv_beta[c]
        const float4 b4 = *(const float4*)(a.pre_inv_beta + c);
        ial[0] = b4.x; ial[1] = b4.y; ial[2] = b4.z; ial[3] = b4.w;
      }
    }
    for (int r = tid >> 3; r < R; r += kThreads / 8) {
      const int gl = l0 - a.pad + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      bool ok[4] = {false, false, false, false};
      const int64_t base = (int64_t)a.x_off + (int64_t)gl * a.ldx + c;
      if (VEC) {
        if (gl >= 0 && gl < len_in && c < a.Cin) {
          const float4 t = *(const float4*)(xb + base);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#pragma unroll
          for (int i = 0; i < 4; ++i) ok[i] = (c + i) < a.Cin;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool o = (c + i) < a.Cin;
          if (a.flat_valid > 0) o = o && (base + i) >= 0 && (base + i) < flat_hi;
          else o = o && gl >= 0 && gl < len_in;
          ok[i] = o;
          if (o) v[i] = xb[base + i];
        }
      }
      float hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = v[i] * sc[i] + sh[i];
        if (a.pre_act == MI355_ACT_LEAKY) t = t > 0.f ? t : t * a.pre_slope;
        else if (a.pre_act == MI355_ACT_SNAKE) {
          const float s = __sinf(al[i] * t);
          t = t + ial[i] * (s * s);
        } else if (a.pre_act == MI355_ACT_ELU) t = t > 0.f ? t : expm1f(t);
        t = ok[i] ? t : 0.f;
        const float h = split_hi<PREC>(t);
        hi[i] = h;
        lo[i] = t - h;
      }
      const int addr = r * 64 + ((((c4 >> 3) ^ ((r >> 2) & 3))) << 4) + ((c4 & 4) << 1);
      uint2 ph;
      if constexpr (PREC >= 3) {
        ph.x = pack_f16x2(hi[0], hi[1]);
        ph.y = pack_f16x2(hi[2], hi[3]);
      } else {
        ph.x = pack_bf16x2(hi[0], hi[1]);
        ph.y = pack_bf16x2(hi[2], hi[3]);
      }
      *(uint2*)(A_hi + addr) = ph;
      if (PREC == 2 || PREC == 4) {
        uint2 pl;
        pl.x = pack_lo<PREC>(lo[0], lo[1]);
        pl.y = pack_lo<PREC>(lo[2], lo[3]);
        *(uint2*)(A_lo + addr) = pl;
      }
    }
  };

  auto compute = [&](int tap, int buf) {
    const char* Bb = Bs + buf * BBYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 bfr[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        bfr[nf] = *(const bf16x8*)(Bb + ((((wn * NF + nf) * 2 + kk) * 64 + lane) << 4));
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int row = wm * WM + mf * 32 + (lane & 31) + tap * dil;
        const int cidx = kk * 2 + (lane >> 5);
        const int addr = row * 64 + ((cidx ^ ((row >> 2) & 3)) << 4);
        const bf16x8 ah = *(const bf16x8*)(A_hi + addr);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = mfma16<PREC>(ah, bfr[nf], acc[mf][nf]);
        if (PREC == 2 || PREC == 4) {
          const bf16x8 alo = *(const bf16x8*)(A_lo + addr);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[mf][nf] = mfma16<PREC>(alo, bfr[nf], acc[mf][nf]);
        }
      }
    }
  };

  issue_B(0, 0);
  int chunk = 0, tap = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (tap == 0) {
      if (s) __syncthreads();  // every wave is done reading the previous chunk's window
      stage_A(chunk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's weight slice has landed in LDS
    __syncthreads();
    if (s + 1 < nsteps) issue_B(s + 1, (s + 1) & 1);
    compute(tap, s & 1);
    if (++tap == K) { tap = 0; ++chunk; }
  }

  // ---------------------------------------------------------------- epilogue
  conv_epilogue<MF, NF, WM, WN, true>(a, acc, b, l0, n0, wm, wn, lane, len_out, false);
}

constexpr int kWsThreads = 512;
constexpr int kWsNld = 6;  // A-window passes of 32 rows per chunk: R <= 192

// -----------------------------------------------------------------------------------------------------
// Third wave-specialised variant (tile code 7128128): ONE barrier per 32-channel chunk instead of one per tap.
// The per-step barrier existed only because the weight slice went through a shared LDS ring.  Here every consumer
// wave loads its own B fragments straight from L2 into registers -- a fragment is 1 KB contiguous in the packed
// image, i.e. one coalesced global_load_dwordx4 per (n-tile, kk) -- one step ahead (plain loads: hipcc counts the
// vmcnt itself, there is no LDS DMA in this kernel).  LDS holds only the two activation-window buffers; the
// producers hand a converted window over once per chunk and already have the loads of the window after that in
// flight.  Between two barriers a consumer wave free-runs K x 16 MFMAs.
// -----------------------------------------------------------------------------------------------------
// ABL (ablation bits, timing experiments only -- results are WRONG when non-zero; reachable only through the explicit
// tile codes ABL*10000000 + 7128128 used by tools/bench_conv.py): 1 = no weight-fragment loads after the first,
// 2 = no activation-fragment LDS reads, 4 = producers only take part in the barriers, 8 = no epilogue.
template <int PREC, int ABL = 0, bool EXT = false>
__global__ __launch_bounds__(kWsThreads, 4) void conv_gemm_ws3_kernel(const mi355_conv_gemm_args a, const int tiles_per_item,
                                                                     const int P, const int NT, const int fold_glog) {
  constexpr int BM = 128, BN = 128, WM = 64, WN = 64, MF = 2, NF = 2;
  const int fold = fold_glog & 1;
  constexpr int NA = a_images<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup ids go round-robin over the 8 XCDs (id & 7 = XCD, each with its own L2).  An XCD owns runs of 8 CONSECUTIVE row tiles
  // (all NT column tiles of a row tile back to back on it): neighbouring tiles share their (K-1)*dil halo rows through that L2 and
  // the column tiles re-read the same activation window from it, while runs still interleave over the XCDs (ragged batches balance).
  // (`fold_glog` = fold | glog << 8.)
  const int id = blockIdx.x;
  const int kq = id >> 3;
  const int ny = kq % NT;
  const int pg = kq / NT;
  const int glog = fold_glog >> 8;  // log2 of the run length (launch_ws3: 8 when there are enough row tiles for several rounds per XCD)
  const int p = (((((pg >> glog) << 3) + (id & 7)) << glog)) | (pg & ((1 << glog) - 1));
  if (p >= P) return;
  const int b = p / tiles_per_item;
  const int l0 = (p - b * tiles_per_item) * BM, n0 = ny * BN;
  const int len_out = a.lens_out ? a.lens_out[b] : a.Lout;
  if (l0 >= len_out) return;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Lin;
  const int K = a.K, dil = a.dil;
  const int R = BM + (K - 1) * dil;
  const int ABYTES = R * 64;
  char* Abase = smem;  // [2 buffers][NA (hi, lo)][R * 64]
  const int nchunks = (a.Cin + 31) >> 5;
  const int NTp = ((a.Cout + 127) >> 7) << 2;
  const int nsteps = nchunks * K;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------ producers
    const int ptid = tid - 256;
    const int c4 = (ptid & 7) * 4;
    const int prow = ptid >> 3;
    const float* xb = a.x + (int64_t)b * a.x_bstride + a.x_off;
    float4 areg[kWsNld];
    // Row group i of this wave covers rows [wrow0 + 32 i, wrow0 + 32 i + 8): groups entirely past the window (R rows) are neither loaded
    // nor converted (wave-uniform branches; K = 3, dil = 1 needs 130 of the 192 rows the six groups could hold).
    const int wrow0 = (wave - 4) * 8;
    auto loadA = [&](int chunk) {
      int c = chunk * 32 + c4;
      if (c >= a.Cin) c = 0;
#pragma unroll
      for (int i = 0; i < kWsNld; ++i) {
        if (wrow0 + i * 32 < R) {
          int gl = l0 - a.pad + prow + i * 32;
          gl = gl < 0 ? 0 : (gl >= a.Lin ? a.Lin - 1 : gl);
          areg[i] = *(const float4*)(xb + (int64_t)gl * a.ldx + c);
        }
      }
    };
    // The prologue activation is selected ONCE per window (ACT is a compile-time tag inside): per-element uniform branches on
    // a.pre_act cost more issue slots on the SIMDs the consumers' MFMAs share than the arithmetic itself.
    auto convertA = [&](int chunk, char* A_hi) {
      char* A_lo = A_hi + ABYTES;
      const int c = chunk * 32 + c4;
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {1.f, 1.f, 1.f, 1.f},
            ial[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.pre_scale) {
        const float4 s4 = *(const float4*)(a.pre_scale + (int64_t)b * a.pre_ld + c);
        const float4 h4 = *(const float4*)(a.pre_shift + (int64_t)b * a.pre_ld + c);
        sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
        sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
      }
      if (a.pre_act == MI355_ACT_SNAKE) {
        const float4 a4 = *(const float4*)(a.pre_alpha + c);
        al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 1 / alpha: v_rcp_f32 + one Newton step (<= 1 ulp; the IEEE divide is ~12 VALU ops per channel and window)
          const float r0 = __builtin_amdgcn_rcpf(al[i]);
          ial[i] = fmaf(fmaf(-al[i], r0, 1.0f), r0, r0);
          al[i] *= 0.15915494309189535f;  // v_sin_f32 takes revolutions
        }
        if constexpr (EXT) {
          if (a.pre_inv_beta) {  // SnakeBeta: x + sin^2(alpha x) * inv_beta[c]
            const float4 b4 = *(const float4*)(a.pre_inv_beta + c);
            ial[0] = b4.x; ial[1] = b4.y; ial[2] = b4.z; ial[3] = b4.w;
          }
        }
      }
      bool cok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cok[j] = (c + j) < a.Cin;
      auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        const float slope = ACT == MI355_ACT_LEAKY ? a.pre_slope : 1.f;
#pragma unroll
        for (int i = 0; i < kWsNld; ++i) {
          const int r = prow + i * 32;
          if (r < R) {
            const int gl = l0 - a.pad + r;
            const bool rowok = gl >= 0 && gl < len_in;
            const float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float u = v[j] * sc[j] + sh[j];
              if constexpr (ACT == MI355_ACT_LEAKY) {
                const float m = u * slope;
                u = u > 0.f ? u : m;
              } else if constexpr (ACT == MI355_ACT_SNAKE) {
                const float sn = __builtin_amdgcn_sinf(al[j] * u);
                u = u + ial[j] * (sn * sn);
              } else if constexpr (ACT == MI355_ACT_ELU) {
                u = u > 0.f ? u : expm1f(u);
              }
              t[j] = (rowok && cok[j]) ? u : 0.f;
            }
            const int addr = r * 64 + ((((c4 >> 3) ^ ((r >> 2) & 3))) << 4) + ((c4 & 4) << 1);
            uint2 ph;
            float hi[4];
            if constexpr (PREC >= 3) {
#pragma unroll
              for (int j = 0; j < 4; ++j) hi[j] = split_hi<PREC>(t[j]);
              ph.x = pack_f16x2(hi[0], hi[1]);
              ph.y = pack_f16x2(hi[2], hi[3]);
            } else {  // one v_cvt_pk_bf16_f32 per pair; the fp32 value of each half is a shift / mask of the packed word
              ph.x = pack_bf16x2(t[0], t[1]);
              ph.y = pack_bf16x2(t[2], t[3]);
              hi[0] = __builtin_bit_cast(float, ph.x << 16);
              hi[1] = __builtin_bit_cast(float, ph.x & 0xffff0000u);
              hi[2] = __builtin_bit_cast(float, ph.y << 16);
              hi[3] = __builtin_bit_cast(float, ph.y & 0xffff0000u);
            }
            *(uint2*)(A_hi + addr) = ph;
            if (PREC == 2 || PREC == 4) {
              uint2 pl;
              pl.x = pack_lo<PREC>(t[0] - hi[0], t[1] - hi[1]);
              pl.y = pack_lo<PREC>(t[2] - hi[2], t[3] - hi[3]);
              *(uint2*)(A_lo + addr) = pl;
            }
          }
        }
      };
      if (a.pre_act == MI355_ACT_SNAKE) body(std::integral_constant<int, MI355_ACT_SNAKE>{});
      else if (a.pre_act == MI355_ACT_LEAKY) body(std::integral_constant<int, MI355_ACT_LEAKY>{});
      else if (EXT && a.pre_act == MI355_ACT_ELU) body(std::integral_constant<int, MI355_ACT_ELU>{});
      else body(std::integral_constant<int, MI355_ACT_NONE>{});
    };
    if constexpr ((ABL & 4) != 0) {
      for (int ci = 0; ci < nchunks; ++ci) lds_barrier();
      return;
    }
    loadA(0);
    convertA(0, Abase);
    if (nchunks > 1) loadA(1);
    lds_barrier();  // barrier #0: window 0 staged, window 1 in flight
    for (int ci = 0; ci + 1 < nchunks; ++ci) {
      convertA(ci + 1, Abase + ((ci + 1) & 1) * NA * ABYTES);  // buffer last read during chunk ci-1
      if (ci + 2 < nchunks) loadA(ci + 2);
      lds_barrier();  // end of chunk ci
    }
    return;
  }

  // -------------------------------------------------------------------------------- consumers
  const int wm = wave >> 1, wn = wave & 1;
  float* yb = a.y + (int64_t)b * a.y_bstride;
  const float* rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
  // fragment (nf, kk) of weight slice s: 1 KB at wfrag + s * NTp * 2048 + (nf * 2 + kk) * 1024
  const char* wfrag = (const char*)a.w + ((int64_t)((n0 >> 5) + wn * NF)) * 2048 + lane * 16;
  const int64_t wstep = (int64_t)NTp * 2048;
  // two static register sets for the weight fragments: even steps compute from b0 while b1 is being loaded, odd steps
  // the other way round (no register-to-register hand-over, and hipcc's counted vmcnt stays exact)
  bf16x8 b0[NF * 2], b1[NF * 2];
#pragma unroll
  for (int f = 0; f < NF * 2; ++f) b0[f] = *(const bf16x8*)(wfrag + f * 1024);
  f32x16 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
  if (fold && l0 + BM <= len_out && n0 + BN <= a.Cout) {  // interior tile: no clamping, 32-bit lane offsets from wave-uniform bases
    const char* rw = rb ? (const char*)(rb + (int64_t)(l0 + wm * WM) * a.ldr + (n0 + wn * WN)) : nullptr;
    const char* yr = (const char*)(yb + (int64_t)(l0 + wm * WM) * a.ldy + (n0 + wn * WN));
    const uint32_t rpb = (uint32_t)a.ldr * 4u, ypb = (uint32_t)a.ldy * 4u;
    const uint32_t roff = (uint32_t)(4 * (lane >> 5)) * rpb + (uint32_t)(lane & 31) * 4u;
    const uint32_t yoff = (uint32_t)(4 * (lane >> 5)) * ypb + (uint32_t)(lane & 31) * 4u;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        if (rw) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = *(const float*)(rw + (roff + (uint32_t)(mf * 32 + (r & 3) + 8 * (r >> 2)) * rpb + (uint32_t)(nf * 128)));
        }
        if (a.accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] += *(const float*)(yr + (yoff + (uint32_t)(mf * 32 + (r & 3) + 8 * (r >> 2)) * ypb + (uint32_t)(nf * 128)));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = rv[r];
      }
  } else if (fold) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int n = n0 + wn * WN + nf * 32 + (lane & 31);
        const bool nok = n < a.Cout;
        const int ncl = nok ? n : a.Cout - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float rv[8];
          int us[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            us[q] = l0 + wm * WM + mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            rv[q] = 0.f;
          }
          if (rb) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[q] = rb[(int64_t)(us[q] < len_out ? us[q] : len_out - 1) * a.ldr + ncl];
          }
          if (a.accumulate) {
            float yv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) yv[q] = yb[(int64_t)(us[q] < len_out ? us[q] : len_out - 1) * a.ldy + ncl];
#pragma unroll
            for (int q = 0; q < 8; ++q) rv[q] += yv[q];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[mf][nf][h * 8 + q] = (nok && us[q] < len_out) ? rv[q] : 0.f;
        }
      }
  }

  lds_barrier();  // barrier #0
  {
    int ci = 0, tap = 0;
    auto compute = [&](const bf16x8 (&bf)[NF * 2]) {
      const char* A_hi = Abase + (ci & 1) * NA * ABYTES;
      const char* A_lo = A_hi + ABYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int row = wm * WM + mf * 32 + (lane & 31) + tap * dil;
          const int cidx = kk * 2 + (lane >> 5);
          const int addr = row * 64 + ((cidx ^ ((row >> 2) & 3)) << 4);
          bf16x8 ah;
          if constexpr ((ABL & 2) != 0) ah = bf[(mf + kk) & 3];
          else ah = *(const bf16x8*)(A_hi + addr);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16<PREC>(ah, bf[nf * 2 + kk], acc[mf][nf]);
          if (PREC == 2 || PREC == 4) {
            bf16x8 alo;
            if constexpr ((ABL & 2) != 0) alo = bf[(mf + kk + 1) & 3];
            else alo = *(const bf16x8*)(A_lo + addr);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16<PREC>(alo, bf[nf * 2 + kk], acc[mf][nf]);
          }
        }
      }
      if (++tap == K) {  // end of chunk: the next window is staged, this one may be overwritten
        tap = 0;
        ++ci;
        if (ci < nchunks) lds_barrier();
      }
    };
    // The prefetch is unconditional (past the last slice it re-reads the last one): a branch around the loads would make
    // hipcc assume they may not have been issued and wait for them with vmcnt(3..0) right away.
    const char* wlast = wfrag + (int64_t)(nsteps - 1) * wstep;
    if constexpr ((ABL & 1) != 0) {
      for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int f = 0; f < NF * 2; ++f) asm volatile("" : "+v"(b0[f]));  // opaque: not hoistable, no loads
        compute(b0);
      }
    } else
    for (int s = 0; s < nsteps; s += 2) {
      const char* w1 = s + 1 < nsteps ? wfrag + (int64_t)(s + 1) * wstep : wlast;
#pragma unroll
      for (int f = 0; f < NF * 2; ++f) b1[f] = *(const bf16x8*)(w1 + f * 1024);
      asm volatile("" ::: "memory");  // keep the prefetch AHEAD of the MFMAs (hipcc otherwise sinks it behind them to save registers)
      __builtin_amdgcn_sched_barrier(0);
      compute(b0);
      if (s + 1 >= nsteps) break;
      const char* w0 = s + 2 < nsteps ? wfrag + (int64_t)(s + 2) * wstep : wlast;
#pragma unroll
      for (int f = 0; f < NF * 2; ++f) b0[f] = *(const bf16x8*)(w0 + f * 1024);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      compute(b1);
    }
  }
  if constexpr ((ABL & 8) != 0) {
    float t = 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[mf][nf][r];
    if (t == 1.2345e-30f) yb[0] = t;  // keeps the MFMAs alive without an epilogue
    return;
  }
  const bool plain = a.up_s == 0 && a.post_act == MI355_ACT_NONE && (fold || (!a.res && !a.accumulate)) && !(EXT && a.post_colscale);
  if (plain && l0 + BM <= len_out && n0 + BN <= a.Cout) conv_epilogue_interior<MF, NF, WM, WN>(a, acc, b, l0, n0, wm, wn, lane);
  else conv_epilogue<MF, NF, WM, WN, EXT>(a, acc, b, l0, n0, wm, wn, lane, len_out, fold != 0);
}

template <int PREC, int ABL = 0, bool EXT = false>
int launch_ws3(const mi355_conv_gemm_args& a, hipStream_t st) {
  const int R = 128 + (a.K - 1) * a.dil;
  MI355_REQUIRE(R <= 32 * kWsNld, "conv_gemm(ws3): window of %d rows exceeds %d (K=%d dil=%d)", R, 32 * kWsNld, a.K, a.dil);
  const size_t lds = (size_t)2 * a_images<PREC>() * R * 64;
  const int tiles_per_item = (a.Lout + 127) / 128;
  const int P = a.B * tiles_per_item;
  const int NT = (a.Cout + 127) / 128;
  const int fold = (a.res || a.accumulate) && a.post_act == MI355_ACT_NONE && a.up_s == 0 && a.res_shift == 0 && !a.post_colscale;
  // runs of 2^glog consecutive row tiles per XCD: long runs share halos in L2, but every XCD must still get several rounds of runs
  // (P = 96 with runs of 8 would leave four XCDs with half the work of the others)
  const int glog = P >= 512 ? 3 : (P >= 256 ? 2 : (P >= 128 ? 1 : 0));
  const int per = 8 << glog;
  const unsigned grid = (unsigned)(((P + per - 1) / per) * per * NT);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_gemm_ws3_kernel<PREC, ABL, EXT>), dim3(grid), dim3(kWsThreads), lds, st, a, tiles_per_item, P, NT, fold | (glog << 8));
  MI355_LAUNCH_CHECK("conv_gemm(ws3)");
  return MI355_OK;
}

template <int BM, int BN, int PREC, bool VEC>
int launch(const mi355_conv_gemm_args& a, hipStream_t st) {
  const int R = BM + (a.K - 1) * a.dil;
  const size_t lds = (size_t)R * 64 * a_images<PREC>() + 2 * (BN / 32) * 2048;
  MI355_REQUIRE(lds <= 64 * 1024, "conv_gemm: window too large for LDS (K=%d dil=%d)", a.K, a.dil);
  dim3 grid((a.Lout + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, PREC, VEC>), grid, dim3(kThreads), lds, st, a);
  MI355_LAUNCH_CHECK("conv_gemm");
  return MI355_OK;
}

}  // namespace

extern "C" int mi355_conv_gemm(const mi355_conv_gemm_args* ap, void* stream) {
  MI355_REQUIRE(ap, "conv_gemm: null args");
  mi355_conv_gemm_args a = *ap;
  MI355_REQUIRE(a.x && a.w && a.y, "conv_gemm: null tensor");
  MI355_REQUIRE(a.B > 0 && a.Lout > 0 && a.Cout > 0 && a.Cin > 0 && a.K > 0, "conv_gemm: bad shape");
  MI355_REQUIRE(a.dil >= 1, "conv_gemm: dilation must be >= 1");
  MI355_REQUIRE((a.pre_scale == nullptr) == (a.pre_shift == nullptr), "conv_gemm: pre_scale/pre_shift must come together");
  MI355_REQUIRE(a.pre_act != MI355_ACT_SNAKE || a.pre_alpha, "conv_gemm: snake needs pre_alpha");
  MI355_REQUIRE(a.pre_act != MI355_ACT_GELU, "conv_gemm: gelu is an epilogue activation");
  MI355_REQUIRE(!a.pre_scale || (a.pre_ld % 4 == 0 && a.pre_ld >= ((a.Cin + 31) & ~31)),
                "conv_gemm: pre_ld must be >= Cin padded to 32 and a multiple of 4");
  MI355_REQUIRE(!a.up_s || (a.up_cout > 0 && a.Cout % a.up_cout == 0 && a.Cout / a.up_cout == a.up_s),
                "conv_gemm: polyphase store needs Cout == up_s*up_cout");
  if (a.precision == 0) a.precision = 2;
  MI355_REQUIRE(a.precision >= 1 && a.precision <= 4, "conv_gemm: precision must be 1, 2, 3 or 4");
  if (a.out_scale == 0.f) a.out_scale = 1.f;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (a.flat_valid == 0) && (a.ldx % 4 == 0) && (a.x_off % 4 == 0) && (a.x_bstride % 4 == 0) &&
                   (((uintptr_t)a.x) % 16 == 0);
  if (a.stats_partial) {
    MI355_REQUIRE(a.up_s == 0, "conv_gemm: fused statistics need a plain (non-polyphase) store");
    MI355_REQUIRE(a.stats_bstride % 2 == 0 && ((uintptr_t)a.stats_partial) % 8 == 0, "conv_gemm: stats_partial must be 8-byte aligned");
  }
  const bool ext = a.pre_inv_beta || a.post_colscale || a.pre_act == MI355_ACT_ELU || a.post_act > MI355_ACT_GELU;
  int tile = a.tile;
  const bool ws_ok = vec && (128 + (a.K - 1) * a.dil) <= 32 * kWsNld && a.Lin > 0;
  if (tile == 0) {
    const int bn = a.Cout <= 64 ? 64 : 128;
    const long wgs128 = (long)a.B * ((a.Lout + 127) / 128) * ((a.Cout + bn - 1) / bn);
    const int bm = (bn == 128 && wgs128 >= 512) ? 128 : 64;
    tile = bm * 1000 + bn;
    // the wave-specialised kernel once there are enough 128 x 128 tiles to fill the 256 CUs
    // (MI355_CONV_NO_WS=1 in the environment keeps the auto choice on the 4-wave kernels: an A/B and bisecting aid)
    static const bool no_ws = getenv("MI355_CONV_NO_WS") != nullptr;
    // MI355_CONV_WS_VARIANT=7 keeps the previous wave-specialised kernel (A/B aid); MI355_CONV_WS_MIN_TILES overrides the fill threshold
    static const int ws_var = getenv("MI355_CONV_WS_VARIANT") ? atoi(getenv("MI355_CONV_WS_VARIANT")) : 4;
    static const int ws_feat = getenv("MI355_CONV_WS_FEAT") ? atoi(getenv("MI355_CONV_WS_FEAT")) : 0;
    static const long ws_min = getenv("MI355_CONV_WS_MIN_TILES") ? atol(getenv("MI355_CONV_WS_MIN_TILES")) : 128;
    if (!no_ws && ws_var == 4 && bn == 128 && wgs128 >= ws_min && a.Cin >= 64 && mi355_conv_ws4_eligible(a, vec)) {
      const int rc = mi355_conv_ws4_launch(a, st, ws_feat);
      if (rc != MI355_ERR_UNSUPPORTED) return rc;  // no instantiation for this prologue / epilogue pair: fall through to the older kernels
    }
    if (!no_ws && ws_ok && bn == 128 && wgs128 >= ws_min && a.Cin >= 64) tile = 7128128;
    else if (bn == 128 && wgs128 >= 512) tile = 64128;  // measured: 64-row tiles beat 128-row tiles on the 4-wave kernel
  }
  if (tile % 10000000 == 6128128) {  // ws4, explicit: 6128128 + 10000000 * feature bits (+ 100000000 * ablation bits)
    MI355_REQUIRE(mi355_conv_ws4_eligible(a, vec), "conv_gemm: the ws4 tile needs 16-B aligned channels-last input / output / residual rows, "
                  "Cout %% 4 == 0, a window of <= 256 rows and precision 2 or 4");
    return mi355_conv_ws4_launch(a, st, ((tile / 10000000) % 10) | ((tile / 100000000) << 4));
  }
  if (a.stats_partial) {  // statistics are produced per 64-row wave block: only the 128-row kernels have those
    MI355_REQUIRE(vec, "conv_gemm: fused statistics need the 16-B aligned channels-last input path");
    if (tile != 7128128) tile = 128128;
  }
  if (tile == 7128128) {  // weights through registers, one barrier per chunk
    MI355_REQUIRE(ws_ok, "conv_gemm: the wave-specialised tile needs a 16-B aligned channels-last input and (K-1)*dil <= 64");
    if (ext)  // SnakeBeta / ELU prologue, extended epilogue: the instantiation that carries them (a few spilled registers)
      return a.precision == 2 ? launch_ws3<2, 0, true>(a, st) : (a.precision == 3 ? launch_ws3<3, 0, true>(a, st) : (a.precision == 4 ? launch_ws3<4, 0, true>(a, st) : launch_ws3<1, 0, true>(a, st)));
    return a.precision == 2 ? launch_ws3<2>(a, st) : (a.precision == 3 ? launch_ws3<3>(a, st) : (a.precision == 4 ? launch_ws3<4>(a, st) : launch_ws3<1>(a, st)));
  }
  if (!vec) {
    if (a.precision == 3) return launch<64, 64, 3, false>(a, st);
    if (a.precision == 4) return launch<64, 64, 4, false>(a, st);
    a.precision = 2;  // with bf16 weights the unaligned (tiny C_in) path always runs the hi+lo split
    return launch<64, 64, 2, false>(a, st);
  }
  if (a.precision == 2) {
    switch (tile) {
      case 128128: return launch<128, 128, 2, true>(a, st);
      case 64128: return launch<64, 128, 2, true>(a, st);
      case 64064: return launch<64, 64, 2, true>(a, st);
    }
  } else if (a.precision == 3) {
    switch (tile) {
      case 128128: return launch<128, 128, 3, true>(a, st);
      case 64128: return launch<64, 128, 3, true>(a, st);
      case 64064: return launch<64, 64, 3, true>(a, st);
    }
  } else if (a.precision == 4) {
    switch (tile) {
      case 128128: return launch<128, 128, 4, true>(a, st);
      case 64128: return launch<64, 128, 4, true>(a, st);
      case 64064: return launch<64, 64, 4, true>(a, st);
    }
  } else {
    switch (tile) {
      case 128128: return launch<128, 128, 1, true>(a, st);
      case 64128: return launch<64, 128, 1, true>(a, st);
      case 64064: return launch<64, 64, 1, true>(a, st);
    }
  }
  mi355_set_error("conv_gemm: unsupported tile %d", tile);
  return MI355_ERR_UNSUPPORTED;
}
