// conv_ws4: the wave-specialised implicit-GEMM conv1d / linear / polyphase conv_transpose kernel (gfx950 only).
//
// 512 threads = 8 waves per workgroup, one 128 x 128 output tile, two workgroups per CU (<= 128 VGPRs, <= 64 KB LDS):
//   waves 4-7  PRODUCERS  stream the fp32 activation window of one 32-channel chunk HBM -> registers (float4 per lane, up to two windows in
//              flight), apply the fused prologue once per input element (AdaIN affine, Snake / SnakeBeta / LeakyReLU / ELU), split the value
//              into hi + lo images of the weights' 16-bit type and write them to LDS (64-B rows, 16-B pieces XOR-swizzled by (row >> 2) & 3:
//              the 32x32x16 fragment ds_read_b128 is conflict free for every tap shift);
//   waves 0-3  CONSUMERS  2 x 2 over the tile (64 x 64 each = four 32x32 accumulators).  Weight fragments come straight from L2 into registers
//              (1 KB contiguous per fragment in the packed image, one tap ahead, two static register sets); activation fragments are read
//              from LDS ONE GROUP OF FOUR MFMAs AHEAD (two static register sets), so the matrix pipe never waits for an LDS round trip.
//              One s_barrier per chunk.
// The MFMA runs in the TRANSPOSED orientation (weights = A operand, activations = B operand): a lane then owns ONE output row and four
// consecutive channels per accumulator quad, i.e. the residual / running-sum loads and the output stores are 16-B per lane (4x fewer memory
// instructions than the row-per-register orientation), and the fused instance-norm statistics reduce across lanes with a reduce-scatter
// butterfly (63 shuffles per statistic per wave).
//
// GEMM mode (K == 1, the nn.Linear layers: PL-BERT, LSTM x-projections, 1x1 shortcuts): a "chunk" is 64 channels staged as two 128-row
// blocks of the window and the "taps" walk the blocks, so there are 32 MFMAs per wave between barriers instead of 16.
//
// Reference call sites replaced: see include/mi355audio.h (mi355_conv_gemm).
#include "conv_common.h"

using namespace mi355conv;

namespace {

constexpr int kThreads = 512;

enum { P_NONE = 0, P_LEAKY = 1, P_SNAKE = 2, P_SNAKEBETA = 3, P_ELU = 4 };
enum { E_BASIC = 0, E_GELU = 1, E_SILU = 2, E_GELU_TANH = 3, E_ELU = 4, E_TANH = 5 };

struct ws4_geom {
  int tiles_per_item, P, NT, glog, fold;
  int nch;       // chunks per tile (GEMM mode: 64-channel super-chunks)
  int keff;      // taps per chunk (GEMM mode: 2 sub-chunks)
  int tap_rows;  // LDS row shift per tap (conv: dilation; GEMM mode: 128)
  int R;         // window rows (conv: 128 + (K-1)*dil; GEMM mode: 256)
  int gemm;
  int nslices;   // weight slices of the packed image = ceil(Cin / 32) * K
  int feat;      // bit 0: consumers at s_setprio 1 (A/B aid)
};

template <int EPI>
__device__ __forceinline__ float post_activation(float v, const int post_act, const float slope) {
  if constexpr (EPI == E_BASIC) return (post_act == MI355_ACT_LEAKY && v < 0.f) ? v * slope : v;
  else if constexpr (EPI == E_GELU) return gelu_erf(v);
  else if constexpr (EPI == E_SILU) return v / (1.0f + expf(-v));
  else if constexpr (EPI == E_GELU_TANH) return gelu_tanh(v);
  else if constexpr (EPI == E_ELU) return v > 0.f ? v : expm1f(v);
  else return tanhf(v);
}

// 16 per-lane partial values (slot i) summed over the 32 lanes of a half wave: afterwards lanes hl and hl ^ 16 hold the total of slot
// hl & 15 in v[0] (one full butterfly step, then a reduce-scatter that halves the live values at every step: 31 shuffles).
__device__ __forceinline__ void reduce_scatter16(float (&v)[16], const int hl) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 16, 64);
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const bool up = (hl & m) != 0;
#pragma unroll
    for (int i = 0; i < m; ++i) {
      // opaque scalar copies: LLVM otherwise folds "up ? v[i] : v[i + m]" into a variable-index extract of the promoted vector (a 16-deep
      // v_cndmask chain per element plus an SGPR pair per comparison)
      float lo = v[i], hi = v[i + m];
      asm volatile("" : "+v"(lo), "+v"(hi));
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      v[i] = keep + __shfl_xor(send, m, 64);
    }
  }
}

template <int PREC, int PRE, int EPI, bool GEMM>
__global__ __launch_bounds__(kThreads, 4) void conv_ws4_kernel(const mi355_conv_gemm_args a, const ws4_geom q) {
  constexpr int BM = 128, BN = 128;
  constexpr int NLD = GEMM ? 8 : 6;  // window passes of 32 rows per chunk (conv: R <= 192; GEMM mode: R = 256)
  constexpr int NA = a_images<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup ids go round-robin over the 8 XCDs (id & 7 = XCD, each with its own L2).  An XCD owns runs of 2^glog CONSECUTIVE row tiles (all NT
  // column tiles of a row tile back to back on it): neighbouring tiles share their halo rows through that L2 and the column tiles re-read the
  // same activation window from it, while runs still interleave over the XCDs.
  const int id = blockIdx.x;
  const int kq = id >> 3;
  const int ny = kq % q.NT;
  const int pg = kq / q.NT;
  const int glog = q.glog;
  const int p = (((((pg >> glog) << 3) + (id & 7)) << glog)) | (pg & ((1 << glog) - 1));
  if (p >= q.P) return;
  const int b = p / q.tiles_per_item;
  const int l0 = (p - b * q.tiles_per_item) * BM, n0 = ny * BN;
  const int len_out = a.lens_out ? a.lens_out[b] : a.Lout;
  if (l0 >= len_out) return;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Lin;
  const int R = q.R;
  const int ABYTES = R * 64;
  char* Abase = smem;  // [2 buffers][NA (hi, lo)][R * 64]
  const int nch = q.nch, keff = q.keff;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------ producers
    const int ptid = tid - 256;
    const int c4 = (ptid & 7) * 4;
    const int prow = ptid >> 3;
    const float* xb = a.x + (int64_t)b * a.x_bstride + a.x_off;
    const int wrow0 = (wave - 4) * 8;  // pass i of this wave covers window rows [wrow0 + 32 i, +8): passes entirely past R are skipped
    constexpr int cstride = GEMM ? 64 : 32;
    float4 s0[NLD], s1[NLD];  // two windows in flight: s0 carries the even chunks, s1 the odd ones

    // window row r = prow + 32 i of a chunk: conv mode = input row l0 - pad + r, channels [32 chunk, +32);
    // GEMM mode = input row l0 + (r & 127), channels [64 chunk + 32 (r >> 7), +32)
    auto loadA = [&](float4 (&areg)[NLD], const int chunk) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        if (GEMM || wrow0 + i * 32 < R) {
          int gl = GEMM ? l0 + prow + 32 * (i & 3) : l0 - a.pad + prow + 32 * i;
          int c = chunk * cstride + (GEMM ? 32 * (i >> 2) : 0) + c4;
          gl = gl < 0 ? 0 : (gl >= a.Lin ? a.Lin - 1 : gl);
          if (c >= a.Cin) c = 0;
          areg[i] = *(const float4*)(xb + (int64_t)gl * a.ldx + c);
        }
      }
    };
    auto convertA = [&](const float4 (&areg)[NLD], const int chunk, char* A_hi) {
      char* A_lo = A_hi + ABYTES;
      const int c = chunk * cstride + c4;  // GEMM mode (no prologue coefficients): passes 4..7 carry channels c + 32
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {1.f, 1.f, 1.f, 1.f}, ial[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (!GEMM) {
        if (a.pre_scale) {
          const float4 s4 = *(const float4*)(a.pre_scale + (int64_t)b * a.pre_ld + c);
          const float4 h4 = *(const float4*)(a.pre_shift + (int64_t)b * a.pre_ld + c);
          sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
          sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
        }
      }
      if constexpr (PRE == P_SNAKE || PRE == P_SNAKEBETA) {
        const float4 a4 = *(const float4*)(a.pre_alpha + c);
        al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
        if constexpr (PRE == P_SNAKEBETA) {  // x + sin^2(alpha x) * inv_beta[c]
          const float4 b4 = *(const float4*)(a.pre_inv_beta + c);
          ial[0] = b4.x; ial[1] = b4.y; ial[2] = b4.z; ial[3] = b4.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (PRE == P_SNAKE) {  // 1 / alpha: v_rcp_f32 + one Newton step (<= 1 ulp)
            const float r0 = __builtin_amdgcn_rcpf(al[j]);
            ial[j] = fmaf(fmaf(-al[j], r0, 1.0f), r0, r0);
          }
          al[j] *= 0.15915494309189535f;  // v_sin_f32 takes revolutions
        }
      }
      const float slope = a.pre_slope;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        if (GEMM || wrow0 + i * 32 < R) {
          const int r = prow + i * 32;
          if (GEMM || r < R) {
            const int gl = GEMM ? l0 + prow + 32 * (i & 3) : l0 - a.pad + r;
            const bool rowok = gl >= 0 && gl < len_in;
            const int cb = c + (GEMM ? 32 * (i >> 2) : 0);
            const float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            float tt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float u = v[j];
              if constexpr (!GEMM) u = u * sc[j] + sh[j];
              if constexpr (PRE == P_LEAKY) {
                const float m = u * slope;
                u = u > 0.f ? u : m;
              } else if constexpr (PRE == P_SNAKE || PRE == P_SNAKEBETA) {
                const float sn = __builtin_amdgcn_sinf(al[j] * u);
                u = u + ial[j] * (sn * sn);
              } else if constexpr (PRE == P_ELU) {
                u = u > 0.f ? u : expm1f(u);
              }
              tt[j] = (rowok && (cb + j) < a.Cin) ? u : 0.f;
            }
            const int addr = r * 64 + ((((c4 >> 3) ^ ((r >> 2) & 3))) << 4) + ((c4 & 4) << 1);
            uint2 ph;
            float hi[4];
            if constexpr (PREC >= 3) {
#pragma unroll
              for (int j = 0; j < 4; ++j) hi[j] = split_hi<PREC>(tt[j]);
              ph.x = pack_f16x2(hi[0], hi[1]);
              ph.y = pack_f16x2(hi[2], hi[3]);
            } else {  // one v_cvt_pk_bf16_f32 per pair; the fp32 value of each half is a shift / mask of the packed word
              ph.x = pack_bf16x2(tt[0], tt[1]);
              ph.y = pack_bf16x2(tt[2], tt[3]);
              hi[0] = __builtin_bit_cast(float, ph.x << 16);
              hi[1] = __builtin_bit_cast(float, ph.x & 0xffff0000u);
              hi[2] = __builtin_bit_cast(float, ph.y << 16);
              hi[3] = __builtin_bit_cast(float, ph.y & 0xffff0000u);
            }
            *(uint2*)(A_hi + addr) = ph;
            if constexpr (NA == 2) {
              uint2 pl;
              pl.x = pack_lo<PREC>(tt[0] - hi[0], tt[1] - hi[1]);
              pl.y = pack_lo<PREC>(tt[2] - hi[2], tt[3] - hi[3]);
              *(uint2*)(A_lo + addr) = pl;
            }
          }
        }
      }
    };

    // chunk ci is converted into buffer ci & 1 while the consumers work on chunk ci - 1 (they left that buffer at the barrier that ended
    // chunk ci - 2); the loads of chunk ci + 2 are issued right behind the conversion, i.e. two windows are always in flight
    loadA(s0, 0);
    if (nch > 1) loadA(s1, 1);
    for (int ci = 0; ci < nch; ci += 2) {
      convertA(s0, ci, Abase);
      if (ci + 2 < nch) loadA(s0, ci + 2);
      lds_barrier();  // window ci staged (= the consumers' end-of-chunk barrier of chunk ci - 1)
      if (ci + 1 < nch) {
        convertA(s1, ci + 1, Abase + NA * ABYTES);
        if (ci + 3 < nch) loadA(s1, ci + 3);
        lds_barrier();
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------- consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int hl = lane & 31, hh = lane >> 5;
  if (q.feat & 1) __builtin_amdgcn_s_setprio(1);
  // fragment (nf, kk) of weight slice s: 1 KB at wfrag + s * wstep + (nf * 2 + kk) * 1024
  const int NTp = ((a.Cout + 127) >> 7) << 2;
  const char* wfrag = (const char*)a.w + ((int64_t)((n0 >> 5) + wn * 2)) * 2048 + lane * 16;
  const int64_t wstep = (int64_t)NTp * 2048;
  const int nsteps = nch * keff;
  const int last_slice = q.nslices - 1;
  auto wptr = [&](const int s) { return wfrag + (int64_t)(s < last_slice ? s : last_slice) * wstep; };
  bf16x8 b0[4], b1[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) b0[f] = *(const bf16x8*)(wfrag + f * 1024);

  // acc[mf][nf][r]: output row l0 + wm*64 + mf*32 + hl, channel n0 + wn*64 + nf*32 + 8*(r>>2) + 4*hh + (r&3)
  f32x16 acc[2][2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

  float* yb = a.y + (int64_t)b * a.y_bstride;
  const float* rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
  const int row_w = l0 + wm * 64;           // first output row of this wave
  const int col_w = n0 + wn * 64 + 4 * hh;  // first channel of this lane's quads
  const bool interior = l0 + BM <= len_out && n0 + BN <= a.Cout;
  if (q.fold) {
    // residual and running sum go in as the initial accumulator value (their latency hides under the staging of the first window)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int u = row_w + mf * 32 + hl;
      const bool rok = u < len_out;
      const int uc = rok ? u : len_out - 1;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = col_w + nf * 32 + 8 * qd;
          const bool ok = rok && c < a.Cout;
          const int cc = c < a.Cout ? c : 0;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rb) v = *(const float4*)(rb + (int64_t)uc * a.ldr + cc);
          if (a.accumulate) {
            const float4 o = *(const float4*)(yb + (int64_t)uc * a.ldy + cc);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          acc[mf][nf][4 * qd + 0] = ok ? v.x : 0.f;
          acc[mf][nf][4 * qd + 1] = ok ? v.y : 0.f;
          acc[mf][nf][4 * qd + 2] = ok ? v.z : 0.f;
          acc[mf][nf][4 * qd + 3] = ok ? v.w : 0.f;
        }
    }
  }

  lds_barrier();  // barrier #0
  {
    int ci = 0, tap = 0;
    bf16x8 ah0, al0, ah1, al1;  // two static activation-fragment sets (al*: the lo image, unused for the single-pass precisions)
    // group g of a tap: kk = g >> 1 (16-channel half of the chunk), mf = g & 1 (32-row half of the wave's rows)
    auto rdA = [&](bf16x8& h, bf16x8& l, const int g, const int tp) {
      const int kk = g >> 1, mf = g & 1;
      const int row = wm * 64 + mf * 32 + hl + tp * q.tap_rows;
      const int cidx = kk * 2 + hh;
      const int addr = row * 64 + ((cidx ^ ((row >> 2) & 3)) << 4);
      const char* A_hi = Abase + (ci & 1) * NA * ABYTES;
      h = *(const bf16x8*)(A_hi + addr);
      if constexpr (NA == 2) l = *(const bf16x8*)(A_hi + ABYTES + addr);
    };
    auto mm = [&](const bf16x8& h, const bf16x8& l, const int g, const bf16x8 (&bf)[4]) {
      const int kk = g >> 1, mf = g & 1;
      acc[mf][0] = mfma16<PREC>(bf[kk], h, acc[mf][0]);
      acc[mf][1] = mfma16<PREC>(bf[2 + kk], h, acc[mf][1]);
      if constexpr (NA == 2) {
        acc[mf][0] = mfma16<PREC>(bf[kk], l, acc[mf][0]);
        acc[mf][1] = mfma16<PREC>(bf[2 + kk], l, acc[mf][1]);
      }
    };
    // one tap = four groups; the fragments of group g+1 are requested before the MFMAs of group g are issued
    auto tap_body = [&](const bf16x8 (&bf)[4]) {
      rdA(ah1, al1, 1, tap);
      __builtin_amdgcn_sched_barrier(0);
      mm(ah0, al0, 0, bf);
      rdA(ah0, al0, 2, tap);
      __builtin_amdgcn_sched_barrier(0);
      mm(ah1, al1, 1, bf);
      rdA(ah1, al1, 3, tap);
      __builtin_amdgcn_sched_barrier(0);
      mm(ah0, al0, 2, bf);
      if (tap + 1 < keff) rdA(ah0, al0, 0, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(ah1, al1, 3, bf);
      if (++tap == keff) {  // end of chunk: the next window is staged behind the barrier, this one may be overwritten
        tap = 0;
        ++ci;
        if (ci < nch) {
          lds_barrier();
          rdA(ah0, al0, 0, 0);
        }
      }
    };
    rdA(ah0, al0, 0, 0);
    // The weight prefetch is unconditional (past the last slice it re-reads the last one): a branch around the loads would make hipcc assume
    // they may not have been issued and wait for them right away.
    for (int s = 0; s < nsteps; s += 2) {
      const char* w1 = wptr(s + 1);
#pragma unroll
      for (int f = 0; f < 4; ++f) b1[f] = *(const bf16x8*)(w1 + f * 1024);
      asm volatile("" ::: "memory");  // keep the prefetch AHEAD of the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      tap_body(b0);
      if (s + 1 >= nsteps) break;
      const char* w0 = wptr(s + 2);
#pragma unroll
      for (int f = 0; f < 4; ++f) b0[f] = *(const bf16x8*)(w0 + f * 1024);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      tap_body(b1);
    }
  }
  if (q.feat & 1) __builtin_amdgcn_s_setprio(0);

  // ---------------------------------------------------------------- epilogue
  const bool want_stats = a.stats_partial != nullptr;
  const float oscale = a.out_scale;
  const bool fast = interior && a.up_s == 0 && a.post_act == MI355_ACT_NONE && EPI == E_BASIC && (q.fold || (!a.res && !a.accumulate)) &&
                    !a.post_colscale;
  if (fast) {
    // interior tile, plain store, residual / running sum (if any) already in the accumulators: one wave-uniform base + 32-bit lane offsets
    char* yw = (char*)(yb + (int64_t)row_w * a.ldy + (n0 + wn * 64));
    const uint32_t ldb = (uint32_t)a.ldy * 4u;
    const uint32_t lane_off = (uint32_t)hl * ldb + (uint32_t)hh * 16u;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bv = *(const float4*)(a.bias + col_w + nf * 32 + 8 * qd);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          float4 v;
          v.x = (acc[mf][nf][4 * qd + 0] + bv.x) * oscale;
          v.y = (acc[mf][nf][4 * qd + 1] + bv.y) * oscale;
          v.z = (acc[mf][nf][4 * qd + 2] + bv.z) * oscale;
          v.w = (acc[mf][nf][4 * qd + 3] + bv.w) * oscale;
          *(float4*)(yw + (lane_off + (uint32_t)(mf * 32) * ldb + (uint32_t)(nf * 128 + qd * 32))) = v;
          acc[mf][nf][4 * qd + 0] = v.x; acc[mf][nf][4 * qd + 1] = v.y; acc[mf][nf][4 * qd + 2] = v.z; acc[mf][nf][4 * qd + 3] = v.w;
        }
      }
  } else {
    // generic: bias, activation, column scale, residual (row >> res_shift), running sum, scale; plain or polyphase (conv_transpose) store.
    // Loads on clamped addresses, only the stores are predicated.  Afterwards acc holds the stored values (0 where nothing was stored).
    const int len_up = a.lens_up ? a.lens_up[b] : a.up_Lout;
    const bool folded = q.fold != 0;
    const float* rbg = folded ? nullptr : rb;
    const bool accum = a.accumulate && !folded;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = col_w + nf * 32 + 8 * qd;
        const bool cok = c < a.Cout;
        const int cc = cok ? c : 0;
        int ocol = cc, rph = 0;
        if (a.up_s) { rph = cc / a.up_cout; ocol = cc - rph * a.up_cout; }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.bias) bv = *(const float4*)(a.bias + ocol);
        if (a.post_colscale) cs = *(const float4*)(a.post_colscale + ocol);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          const int u = row_w + mf * 32 + hl;
          bool ok = cok && u < len_out;
          int orow = u < len_out ? u : len_out - 1;
          if (a.up_s) {
            int nc = orow * a.up_s + rph - a.up_p;
            ok = ok && nc >= 0 && nc < len_up;
            nc = nc < 0 ? 0 : (nc >= len_up ? (len_up > 0 ? len_up - 1 : 0) : nc);
            orow = nc + a.up_row_off;
          }
          float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rbg) rv = *(const float4*)(rbg + (int64_t)(orow >> a.res_shift) * a.ldr + ocol);
          float* yp = yb + (int64_t)orow * a.ldy + ocol;
          if (accum) {
            const float4 o = *(const float4*)yp;
            rv.x += o.x; rv.y += o.y; rv.z += o.z; rv.w += o.w;
          }
          float4 v;
          v.x = (post_activation<EPI>(acc[mf][nf][4 * qd + 0] + bv.x, a.post_act, a.post_slope) * cs.x + rv.x) * oscale;
          v.y = (post_activation<EPI>(acc[mf][nf][4 * qd + 1] + bv.y, a.post_act, a.post_slope) * cs.y + rv.y) * oscale;
          v.z = (post_activation<EPI>(acc[mf][nf][4 * qd + 2] + bv.z, a.post_act, a.post_slope) * cs.z + rv.z) * oscale;
          v.w = (post_activation<EPI>(acc[mf][nf][4 * qd + 3] + bv.w, a.post_act, a.post_slope) * cs.w + rv.w) * oscale;
          if (ok) *(float4*)yp = v;
          acc[mf][nf][4 * qd + 0] = ok ? v.x : 0.f; acc[mf][nf][4 * qd + 1] = ok ? v.y : 0.f;
          acc[mf][nf][4 * qd + 2] = ok ? v.z : 0.f; acc[mf][nf][4 * qd + 3] = ok ? v.w : 0.f;
        }
      }
  }
  // Fused instance-norm statistics of what was just stored: this wave's 64 rows (= one MI355_STATS_ROWS block) x 64 channels.  A lane holds
  // 2 rows (mf) x 32 channel slots (slot = nf*16 + r, r the accumulator register); two passes over the registers: sum -> block mean ->
  // sum of squared deviations, each reduced over the 32 lanes of a half wave by a reduce-scatter butterfly (lane hl ends up with slot hl).
  // (sum, M2 about the block mean) per channel goes to stats_partial[b][row block][n]; adain_from_partials merges the blocks in float64.
  if (want_stats && a.up_s == 0 && row_w < len_out) {
    const int cnt = len_out - row_w < 64 ? len_out - row_w : 64;
    const float rcnt = 1.0f / (float)cnt;
    const bool rok0 = row_w + hl < len_out, rok1 = row_w + 32 + hl < len_out;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = acc[0][nf][i] + acc[1][nf][i];  // rows past len_out hold 0
      reduce_scatter16(t, hl);
      const float sum = t[0];
      const float mean = sum * rcnt;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float mi = __shfl(mean, (lane & 32) + i, 64);
        const float d0 = rok0 ? acc[0][nf][i] - mi : 0.f;
        const float d1 = rok1 ? acc[1][nf][i] - mi : 0.f;
        t[i] = d0 * d0 + d1 * d1;
      }
      reduce_scatter16(t, hl);
      // slot r = hl & 15 of this nf  ->  channel n0 + wn*64 + nf*32 + 8*(r>>2) + 4*hh + (r&3)
      const int r = hl & 15;
      const int n = n0 + wn * 64 + nf * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
      if (hl < 16 && n < a.Cout)
        *(float2*)(a.stats_partial + (int64_t)b * a.stats_bstride + ((int64_t)(row_w / MI355_STATS_ROWS) * a.Cout + n) * 2) = make_float2(sum, t[0]);
    }
  }
}

// GEMM mode: pure linear layers (K == 1, no prologue) with at least two 32-channel chunks
bool gemm_mode(const mi355_conv_gemm_args& a) { return a.K == 1 && a.Cin >= 64 && a.pre_act == MI355_ACT_NONE && !a.pre_scale; }

template <int PREC, int PRE, int EPI, bool GEMM>
int launch_ws4(const mi355_conv_gemm_args& a, hipStream_t st, const int feat) {
  ws4_geom q;
  q.gemm = GEMM ? 1 : 0;
  const int chunks32 = (a.Cin + 31) >> 5;
  q.nslices = chunks32 * a.K;
  if (q.gemm) {
    q.nch = (chunks32 + 1) >> 1;
    q.keff = 2;
    q.tap_rows = 128;
    q.R = 256;
  } else {
    q.nch = chunks32;
    q.keff = a.K;
    q.tap_rows = a.dil;
    q.R = 128 + (a.K - 1) * a.dil;
  }
  MI355_REQUIRE(q.R <= (GEMM ? 256 : 192), "conv_gemm(ws4): window of %d rows exceeds %d (K=%d dil=%d)", q.R, GEMM ? 256 : 192, a.K, a.dil);
  const size_t lds = (size_t)2 * a_images<PREC>() * q.R * 64;
  q.tiles_per_item = (a.Lout + 127) / 128;
  q.P = a.B * q.tiles_per_item;
  q.NT = (a.Cout + 127) / 128;
  q.fold = ((a.res || a.accumulate) && a.post_act == MI355_ACT_NONE && a.up_s == 0 && a.res_shift == 0 && !a.post_colscale) ? 1 : 0;
  // runs of 2^glog consecutive row tiles per XCD: long runs share halos in L2, but every XCD must still get several rounds of runs
  q.glog = q.P >= 512 ? 3 : (q.P >= 256 ? 2 : (q.P >= 128 ? 1 : 0));
  q.feat = feat;
  const int per = 8 << q.glog;
  const unsigned grid = (unsigned)(((q.P + per - 1) / per) * per * q.NT);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_ws4_kernel<PREC, PRE, EPI, GEMM>), dim3(grid), dim3(kThreads), lds, st, a, q);
  MI355_LAUNCH_CHECK("conv_gemm(ws4)");
  return MI355_OK;
}

int pre_kind(const mi355_conv_gemm_args& a) {
  switch (a.pre_act) {
    case MI355_ACT_NONE: return P_NONE;
    case MI355_ACT_LEAKY: return P_LEAKY;
    case MI355_ACT_SNAKE: return a.pre_inv_beta ? P_SNAKEBETA : P_SNAKE;
    case MI355_ACT_ELU: return P_ELU;
  }
  return -1;
}
int epi_kind(const mi355_conv_gemm_args& a) {
  switch (a.post_act) {
    case MI355_ACT_NONE:
    case MI355_ACT_LEAKY: return E_BASIC;
    case MI355_ACT_GELU: return E_GELU;
    case MI355_ACT_SILU: return E_SILU;
    case MI355_ACT_GELU_TANH: return E_GELU_TANH;
    case MI355_ACT_ELU: return E_ELU;
    case MI355_ACT_TANH: return E_TANH;
  }
  return -1;
}

// 16-B aligned float rows: base pointer, row pitch and batch pitch
bool aligned4(const float* p, int64_t bstride, int ld) { return ((uintptr_t)p % 16 == 0) && (bstride % 4 == 0) && (ld % 4 == 0); }

}  // namespace

bool mi355_conv_ws4_eligible(const mi355_conv_gemm_args& a, bool vec) {
  if (!vec || a.Lin <= 0) return false;
  if (!gemm_mode(a) && 128 + (a.K - 1) * a.dil > 192) return false;
  if (a.precision != 2 && a.precision != 4) return false;  // the single-pass fast modes stay on the 4-wave kernels
  if (!aligned4(a.y, a.y_bstride, a.ldy)) return false;
  if (a.res && !aligned4(a.res, a.res_bstride, a.ldr)) return false;
  if (a.up_s ? (a.up_cout % 4 != 0) : (a.Cout % 4 != 0)) return false;
  if (a.bias && (uintptr_t)a.bias % 16 != 0) return false;
  if (a.post_colscale && (uintptr_t)a.post_colscale % 16 != 0) return false;
  if (a.stats_partial && a.Cout % 4 != 0) return false;
  return pre_kind(a) >= 0 && epi_kind(a) >= 0;
}

#define WS4_CASE(PREC, PRE, EPI) \
  if (a.precision == PREC && pre == PRE && epi == EPI && !gemm) return launch_ws4<PREC, PRE, EPI, false>(a, st, feat)
#define WS4_GEMM(PREC, EPI) \
  if (a.precision == PREC && epi == EPI && gemm) return launch_ws4<PREC, P_NONE, EPI, true>(a, st, feat)

int mi355_conv_ws4_launch(const mi355_conv_gemm_args& a, hipStream_t st, int feat) {
  const int pre = pre_kind(a), epi = epi_kind(a);
  const bool gemm = gemm_mode(a);
  // Kokoro (bf16 checkpoints): AdaIN resblocks (Snake), AdainResBlk1d / upsamplers (LeakyReLU), plain linears (+ GELU: PL-BERT FFN)
  WS4_CASE(2, P_NONE, E_BASIC);
  WS4_CASE(2, P_LEAKY, E_BASIC);
  WS4_CASE(2, P_SNAKE, E_BASIC);
  WS4_GEMM(2, E_BASIC);
  WS4_GEMM(2, E_GELU);
  // fp16 checkpoints (Whisper: conv stem + GELU, linears)
  WS4_CASE(4, P_NONE, E_BASIC);
  WS4_CASE(4, P_NONE, E_GELU);
  WS4_GEMM(4, E_BASIC);
  WS4_GEMM(4, E_GELU);
  mi355_set_error("conv_gemm(ws4): no instantiation for precision %d, prologue %d, epilogue %d", a.precision, pre, epi);
  return MI355_ERR_UNSUPPORTED;
}
