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
  conv_epilogue<MF, NF, WM, WN, -1>(a, acc, b, l0, n0, wm, wn, lane, len_out, false);
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
  int tile = a.tile;
  if (tile == 0) {
    const int bn = a.Cout <= 64 ? 64 : 128;
    const long wgs128 = (long)a.B * ((a.Lout + 127) / 128) * ((a.Cout + bn - 1) / bn);
    const int bm = (bn == 128 && wgs128 >= 512) ? 128 : 64;
    tile = bm * 1000 + bn;
    // the wave-specialised kernel once there are enough 128 x 128 tiles to occupy the 256 CUs
    // (MI355_CONV_NO_WS=1 in the environment keeps the auto choice on the 4-wave kernels: an A/B and bisecting aid;
    // MI355_CONV_WS_FEAT = feature bits of conv_common.h; MI355_CONV_WS_MIN_TILES overrides the fill threshold)
    static const bool no_ws = getenv("MI355_CONV_NO_WS") != nullptr;
    static const int ws_feat = getenv("MI355_CONV_WS_FEAT") ? atoi(getenv("MI355_CONV_WS_FEAT")) : 0;
    static const long ws_min = getenv("MI355_CONV_WS_MIN_TILES") ? atol(getenv("MI355_CONV_WS_MIN_TILES")) : 128;
    if (!no_ws && bn == 128 && wgs128 >= ws_min && a.Cin >= 64 && mi355_conv_ws4_eligible(a, vec)) {
      const int rc = mi355_conv_ws4_launch(a, st, ws_feat);
      if (rc != MI355_ERR_UNSUPPORTED) return rc;  // no instantiation for this prologue / epilogue pair: fall through to the 4-wave kernels
    }
    if (bn == 128 && wgs128 >= 512) tile = 64128;  // measured: 64-row tiles beat 128-row tiles on the 4-wave kernel
  }
  if (tile % 10000000 == 6128128) {  // ws4, explicit: 6128128 + 10000000 * feature bits (+ 100000000 * ablation bits)
    MI355_REQUIRE(mi355_conv_ws4_eligible(a, vec), "conv_gemm: the wave-specialised tile needs 16-B aligned channels-last input rows and a window of <= 192 rows");
    return mi355_conv_ws4_launch(a, st, ((tile / 10000000) % 10) | ((tile / 100000000) << 4));
  }
  if (a.stats_partial) {  // statistics are produced per 64-row wave block: only the 128-row kernels have those
    MI355_REQUIRE(vec, "conv_gemm: fused statistics need the 16-B aligned channels-last input path");
    tile = 128128;
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
