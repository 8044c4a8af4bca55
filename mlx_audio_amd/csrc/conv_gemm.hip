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

// Split-K geometry (SPLIT instantiations): blockIdx.z = b * ksplit + kz; the workgroup covers chunks [kz * cpz, min(nchunks, (kz + 1) * cpz)) and
// stores its raw partial tile (the host hands it args whose y is the slab workspace [B][ksplit][Lout][Cout] and whose epilogue is switched off);
// conv_split_finish_kernel sums the slabs in a fixed order and applies the epilogue of the original call.
struct split_geom {
  int ksplit, cpz;
  int lo_slices;   // MX images (precision 5): e4m3 tap-pair slices that follow a chunk's K fp16 tap slices (skipped here), 0 otherwise
};

template <int BM, int BN, int PREC, bool VEC, bool SPLIT = false>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(const mi355_conv_gemm_args a, const split_geom sg) {
  constexpr int WM = BM / 2, WN = BN / 2, MF = WM / 32, NF = WN / 32;
  constexpr int BBYTES = (BN / 32) * 2048;  // one (chunk, tap) weight slice of this block
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int b = SPLIT ? (int)blockIdx.z / sg.ksplit : (int)blockIdx.z, l0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
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
  int c_begin = 0, c_end = nchunks;
  if constexpr (SPLIT) {
    c_begin = ((int)blockIdx.z - b * sg.ksplit) * sg.cpz;
    c_end = c_begin + sg.cpz < nchunks ? c_begin + sg.cpz : nchunks;
  }
  const int s_begin = c_begin * K, s_end = c_end * K;
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
    const int wslice = sg.lo_slices ? step + (step / K) * sg.lo_slices : step;
    const char* src = (const char*)a.w + ((int64_t)wslice * NTp + (n0 >> 5)) * 2048;
#pragma unroll
    for (int i = 0; i < BN / 64; ++i) {
      const int off = (i * 4 + wave) * 1024;
      glds16(src + off + lane * 16, Bs + buf * BBYTES + off);
    }
  };

  auto stage_A = [&](int chunk) {
    const int c4 = (tid & 7) * 4;
    const int c = chunk * 32 + c4;
    // the prologue's per-channel operands, requested together and unconditionally (an absent one reads the weight image -- valid, 16-byte aligned --
    // and is dropped by a select): as loads under `if (ptr)` with constants on the other side each was its own round trip in front of the window
    const float* const safe = (const float*)a.w;
    const bool has_aff = a.pre_scale != nullptr, has_snake = a.pre_act == MI355_ACT_SNAKE, has_beta = has_snake && a.pre_inv_beta != nullptr;
    const float4 s4 = *(const float4*)(has_aff ? a.pre_scale + (int64_t)b * a.pre_ld + c : safe);
    const float4 h4 = *(const float4*)(has_aff ? a.pre_shift + (int64_t)b * a.pre_ld + c : safe);
    const float4 a4 = *(const float4*)(has_snake ? a.pre_alpha + c : safe);
    const float4 b4 = *(const float4*)(has_beta ? a.pre_inv_beta + c : safe);
    // ... and the first four window rows of this thread with them (VEC): one round trip per chunk for windows of <= 128 rows
    constexpr int NR = 4, RS = kThreads / 8;
    const int cc = c < a.Cin ? c : 0;
    auto load_rows = [&](const int rb, float4 (&t)[NR], bool (&rin)[NR]) {
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const int r = rb + u * RS;
        const int gl = l0 - a.pad + r;
        int64_t base = (int64_t)a.x_off + (int64_t)gl * a.ldx + cc;
        if (a.flat_valid > 0) {   // flattened strided conv on 16-byte aligned runs: validity by flat element index, every bound a multiple of 4
          rin[u] = r < R && c < a.Cin && base >= 0 && base + 3 < flat_hi;
          base = base < 0 ? 0 : (base + 3 < flat_hi ? base : (flat_hi >= 4 ? flat_hi - 4 : 0));   // an empty item (flat_hi = 0) reads element 0, dropped by rin
        } else {
          rin[u] = r < R && c < a.Cin && gl >= 0 && gl < len_in;
          const int glc = gl < 0 ? 0 : (gl < len_in ? gl : (len_in > 0 ? len_in - 1 : 0));   // lens_in[b] == 0: row 0 (inside the allocation), dropped by rin
          base = (int64_t)a.x_off + (int64_t)glc * a.ldx + cc;
        }
        t[u] = *(const float4*)(xb + base);
      }
    };
    float4 t0[NR];
    bool rin0[NR];
    if constexpr (VEC) load_rows(tid >> 3, t0, rin0);
    float sc[4], sh[4], al[4], ial[4];
    sc[0] = has_aff ? s4.x : 1.f; sc[1] = has_aff ? s4.y : 1.f; sc[2] = has_aff ? s4.z : 1.f; sc[3] = has_aff ? s4.w : 1.f;
    sh[0] = has_aff ? h4.x : 0.f; sh[1] = has_aff ? h4.y : 0.f; sh[2] = has_aff ? h4.z : 0.f; sh[3] = has_aff ? h4.w : 0.f;
    al[0] = has_snake ? a4.x : 1.f; al[1] = has_snake ? a4.y : 1.f; al[2] = has_snake ? a4.z : 1.f; al[3] = has_snake ? a4.w : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ial[i] = 1.0f / al[i];
    if (has_beta) { ial[0] = b4.x; ial[1] = b4.y; ial[2] = b4.z; ial[3] = b4.w; }   // SnakeBeta: x + sin^2(alpha x) * inv_beta[c]
    // quantising prologue (a.pre_fq): the value of conv_common.h's fq_pre_value, then the utterance's dynamic uint8 quantise / dequantise
    const bool fqon = a.pre_fq != nullptr;
    const FakeQuant fq = fqon ? FakeQuant(-a.pre_fq[2 * b], a.pre_fq[2 * b + 1]) : FakeQuant(0.f, 0.f);
    fq_coef fk[4];
    if (fqon) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fk[i] = fq_load_coef(a.pre_scale, a.pre_shift, (int64_t)b * a.pre_ld, a.pre_act, a.pre_alpha, (c + i) < a.Cin ? c + i : 0);
    }
    // one row of the window: prologue, hi / lo split, LDS store
    auto finish_row = [&](const int r, const float (&v)[4], const bool (&ok)[4]) {
      float hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = v[i] * sc[i] + sh[i];
        if (fqon) t = fq(fq_pre_value(v[i], fk[i], a.pre_scale != nullptr, a.pre_act, a.pre_slope));
        else if (a.pre_act == MI355_ACT_LEAKY) t = t > 0.f ? t : t * a.pre_slope;
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
    };
    if constexpr (VEC) {
      // FOUR rows' loads in flight per thread, unconditional (a row / channel group outside the input reads a clamped, valid address and is zeroed
      // through ok[]): as `if (inside) v = load` per row every row was its own round trip -- three to six SERIAL ones per chunk, which is what a
      // launch of few tiles (one utterance per call, the split-K passes) spends its time on
      auto process_rows = [&](const int rb, const float4 (&t)[NR], const bool (&rin)[NR]) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
          const int r = rb + u * RS;
          if (r >= R) break;
          const float v[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
          bool ok[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) ok[i] = rin[u] && (c + i) < a.Cin;
          finish_row(r, v, ok);
        }
      };
      process_rows(tid >> 3, t0, rin0);
      for (int rb = (tid >> 3) + NR * RS; rb < R; rb += NR * RS) {
        float4 t[NR];
        bool rin[NR];
        load_rows(rb, t, rin);
        process_rows(rb, t, rin);
      }
    } else {
      for (int r = tid >> 3; r < R; r += kThreads / 8) {
        const int gl = l0 - a.pad + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        bool ok[4] = {false, false, false, false};
        const int64_t base = (int64_t)a.x_off + (int64_t)gl * a.ldx + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool o = (c + i) < a.Cin;
          if (a.flat_valid > 0) o = o && (base + i) >= 0 && (base + i) < flat_hi;
          else o = o && gl >= 0 && gl < len_in;
          ok[i] = o;
          if (o) v[i] = xb[base + i];
        }
        finish_row(r, v, ok);
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

  issue_B(s_begin, 0);
  int chunk = c_begin, tap = 0;
  for (int s = s_begin; s < s_end; ++s) {
    if (tap == 0) {
      if (s != s_begin) __syncthreads();  // every wave is done reading the previous chunk's window
      stage_A(chunk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's weight slice has landed in LDS
    __syncthreads();
    if (s + 1 < s_end) issue_B(s + 1, (s + 1 - s_begin) & 1);
    compute(tap, (s - s_begin) & 1);
    if (++tap == K) { tap = 0; ++chunk; }
  }

  // ---------------------------------------------------------------- epilogue (SPLIT: a plain store of the partial tile into slab blockIdx.z)
  conv_epilogue<MF, NF, WM, WN, SPLIT ? 0 : -1>(a, acc, SPLIT ? (int)blockIdx.z : b, l0, n0, wm, wn, lane, len_out, false);
}

// e4m3 tap-pair slices per chunk of the weight image of the call being dispatched (MX images handed to these kernels: they run the precision-4
// arithmetic on the fp16 slices and skip the rest); set by mi355_conv_gemm for the duration of one call
thread_local int t_mx_lo_slices = 0;

template <int BM, int BN, int PREC, bool VEC>
int launch(const mi355_conv_gemm_args& a, hipStream_t st) {
  const int R = BM + (a.K - 1) * a.dil;
  const size_t lds = (size_t)R * 64 * a_images<PREC>() + 2 * (BN / 32) * 2048;
  MI355_REQUIRE(lds <= 64 * 1024, "conv_gemm: window too large for LDS (K=%d dil=%d)", a.K, a.dil);
  dim3 grid((a.Lout + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, PREC, VEC>), grid, dim3(kThreads), lds, st, a, split_geom{1, 0, t_mx_lo_slices});
  MI355_LAUNCH_CHECK("conv_gemm");
  return MI355_OK;
}

// ---------------------------------------------------------------------------------------------- split-K for launches of few tiles
// One utterance per call (the latency configuration) leaves most convs of the network with 2..90 output tiles and 24..100 dependent
// (chunk, tap) steps each: the kernel then runs at the latency of its serial K loop on a fraction of the CUs (profiles/r3_kernel_stats_b1_call14.txt:
// 60 us for an 80 x 768 x 768 PL-BERT projection on 12 workgroups).  With a workspace in the args the dispatcher cuts the chunk range over
// ksplit workgroups per tile (grid z) and a second kernel sums the partial tiles in slab order and applies the epilogue of the call -- bias,
// activation, LayerScale, residual (row >> res_shift), running sum, out_scale, plain or polyphase store and the fused instance-norm statistics
// of the stored rows -- so every feature of mi355_conv_gemm keeps its meaning.  Sums are in a fixed order (deterministic), but not the order
// of the unsplit kernel: results differ from it by fp32 rounding of the accumulation.
// One workgroup per 64 x 64 block of GEMM outputs; thread (r16, c4) owns rows r16 + 16 i (i < 4) and columns 4 c4 .. + 4: every slab read is a
// 16-byte load, the four rows and the groups are independent loads in flight.  ldp = slab row pitch (C_out padded to 4).
__global__ __launch_bounds__(256) void conv_split_finish_kernel(const mi355_conv_gemm_args a, const float* __restrict__ part, const int ksplit,
                                                                const int64_t slab, const int ldp) {
  __shared__ float sst[16][64][3];
  const int tid = threadIdx.x, c4 = tid & 15, r16 = tid >> 4;
  const int b = blockIdx.z, nb = blockIdx.y * 64 + 4 * c4, row0 = blockIdx.x * 64;
  const int len_out = a.lens_out ? a.lens_out[b] : a.Lout;
  if (row0 >= len_out) return;
  const int len_up = a.lens_up ? a.lens_up[b] : a.up_Lout;
  int ocol[4], rph[4];
  float bias[4], cscale[4];
  bool nok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = nb + e;
    nok[e] = n < a.Cout;
    ocol[e] = nok[e] ? n : 0;
    rph[e] = 0;
    if (a.up_s) { rph[e] = ocol[e] / a.up_cout; ocol[e] -= rph[e] * a.up_cout; }
    bias[e] = (a.bias && nok[e]) ? a.bias[ocol[e]] : 0.f;
    cscale[e] = (a.post_colscale && nok[e]) ? a.post_colscale[ocol[e]] : 1.f;
  }
  float* const yb = a.y + (int64_t)b * a.y_bstride;
  const float* const rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
  const float* const pb = part + (int64_t)b * ksplit * slab + nb;
  const bool colok = nb < ldp;   // the 16-byte piece lies inside the slab row
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rok[i] = colok && (row0 + r16 + 16 * i) < len_out;
  auto add4 = [](float4& s, const float4 t) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; };
  int g = 0;
  for (; g + 2 <= ksplit; g += 2) {   // eight independent loads in flight, added in slab order
    float4 t[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[u][i] = rok[i] ? *(const float4*)(pb + (int64_t)(g + u) * slab + (int64_t)(row0 + r16 + 16 * i) * ldp) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) add4(acc[i], t[u][i]);
  }
  for (; g < ksplit; ++g) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (rok[i]) add4(acc[i], *(const float4*)(pb + (int64_t)g * slab + (int64_t)(row0 + r16 + 16 * i) * ldp));
  }
  float sK[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = row0 + r16 + 16 * i;
    const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bool valid = nok[e] && u < len_out;
      int orow = u;
      if (a.up_s) {
        const int nc = u * a.up_s + rph[e] - a.up_p;
        valid = valid && nc >= 0 && nc < len_up;
        orow = nc + a.up_row_off;
      }
      if (!valid) continue;
      float v = av[e] + bias[e];
      switch (a.post_act) {
        case MI355_ACT_LEAKY: v = v > 0.f ? v : v * a.post_slope; break;
        case MI355_ACT_GELU: v = gelu_erf(v); break;
        case MI355_ACT_SILU: v = v / (1.0f + expf(-v)); break;
        case MI355_ACT_GELU_TANH: v = gelu_tanh(v); break;
        case MI355_ACT_ELU: v = v > 0.f ? v : expm1f(v); break;
        case MI355_ACT_TANH: v = tanhf(v); break;
        default: break;
      }
      float rv = rb ? rb[(int64_t)(orow >> a.res_shift) * a.ldr + ocol[e]] : 0.f;
      float* const dst = yb + (int64_t)orow * a.ldy + ocol[e];
      if (a.accumulate) rv += *dst;
      v = (v * cscale[e] + rv) * a.out_scale;
      *dst = v;
      sK[e] = cnt[e] == 0 ? v : sK[e];
      const float d = v - sK[e];
      s1[e] += d;
      s2[e] += d * d;
      ++cnt[e];
    }
  }
  if (!a.stats_partial) return;
  // (count, mean, M2) of this thread's <= 4 stored rows per column; the 16 row threads of a column merge with Chan's formula in thread order
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float cl = (float)cnt[e];
    sst[r16][4 * c4 + e][0] = cl;
    sst[r16][4 * c4 + e][1] = cnt[e] ? sK[e] + s1[e] / cl : 0.f;
    sst[r16][4 * c4 + e][2] = cnt[e] ? s2[e] - s1[e] * s1[e] / cl : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y * 64 + tid;
  if (tid < 64 && n < a.Cout) {
    float ct = sst[0][tid][0], mt = sst[0][tid][1], vt = sst[0][tid][2];
    for (int j = 1; j < 16; ++j) {
      const float cj = sst[j][tid][0], mj = sst[j][tid][1], vj = sst[j][tid][2];
      if (cj > 0.f) {
        const float cn = ct + cj, dm = mj - mt;
        vt = vt + vj + (ct > 0.f ? dm * dm * ct * cj / cn : 0.f);
        mt = (mt * ct + mj * cj) / cn;
        ct = cn;
      }
    }
    *(float2*)(a.stats_partial + (int64_t)b * a.stats_bstride + ((int64_t)blockIdx.x * a.Cout + n) * 2) = make_float2(mt * ct, vt);
  }
}

// conv_split_finish_lean_kernel: the same sums, epilogue and statistics for the plain store (no polyphase upsampling) with C_out a multiple of 4
// and 16-byte aligned operands -- every linear / conv of the one-utterance configuration except the generator's two transposed convs -- as
// straight-line code.  In conv_split_finish_kernel every guarded load (`rok ? *p : 0`, `a.bias ? ... : 0`, `rb ? rb[...] : 0`, the accumulate
// read) compiles to a branch with its own s_waitcnt vmcnt(0): ~70 serial round trips in the generated code, 11-14 us per launch for a 64 x 64
// block, 116 launches = 1.36 ms of the 7.6 ms one-utterance pass (profiles/r3_kernel_stats_b1_bygrid_call44.txt).  Here every load is
// unconditional (rows / columns past the block clamp to valid ones, null operands read the slab instead) and 16 bytes wide, all requested before
// the first use; only the stores are predicated.
__global__ __launch_bounds__(256) void conv_split_finish_lean_kernel(const mi355_conv_gemm_args a, const float* __restrict__ part, const int ksplit,
                                                                     const int64_t slab, const int ldp) {
  __shared__ float sst[16][64][3];
  const int tid = threadIdx.x, c4 = tid & 15, r16 = tid >> 4;
  const int b = blockIdx.z, nb = blockIdx.y * 64 + 4 * c4, row0 = blockIdx.x * 64;
  const int len_out = a.lens_out ? a.lens_out[b] : a.Lout;
  if (row0 >= len_out) return;   // workgroup-uniform
  const bool colok = nb < a.Cout;                       // C_out % 4 == 0: the piece is whole or absent
  const int nbc = colok ? nb : a.Cout - 4;
  const float* const pb = part + (int64_t)b * ksplit * slab + nbc;
  const bool has_bias = a.bias != nullptr, has_cs = a.post_colscale != nullptr, has_res = a.res != nullptr;
  int urow[4];
  bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = row0 + r16 + 16 * i;
    rok[i] = colok && u < len_out;
    urow[i] = u < len_out ? u : len_out - 1;
  }
  const float4 bias4 = *(const float4*)(has_bias ? a.bias + nbc : pb + (int64_t)urow[0] * ldp);
  const float4 cs4 = *(const float4*)(has_cs ? a.post_colscale + nbc : pb + (int64_t)urow[0] * ldp);
  float* const yb = a.y + (int64_t)b * a.y_bstride + nbc;
  float4 res4[4], old4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* const dummy = pb + (int64_t)urow[i] * ldp;
    res4[i] = *(const float4*)(has_res ? a.res + (int64_t)b * a.res_bstride + (int64_t)(urow[i] >> a.res_shift) * a.ldr + nbc : dummy);
    old4[i] = *(const float4*)(a.accumulate ? yb + (int64_t)urow[i] * a.ldy : dummy);
  }
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add4 = [](float4& s, const float4 t) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; };
  int g = 0;
  for (; g + 2 <= ksplit; g += 2) {   // eight loads in flight, added in slab order
    float4 t[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[u][i] = *(const float4*)(pb + (int64_t)(g + u) * slab + (int64_t)urow[i] * ldp);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) add4(acc[i], t[u][i]);
  }
  for (; g < ksplit; ++g) {
    float4 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = *(const float4*)(pb + (int64_t)g * slab + (int64_t)urow[i] * ldp);
#pragma unroll
    for (int i = 0; i < 4; ++i) add4(acc[i], t[i]);
  }
  const float bias[4] = {bias4.x, bias4.y, bias4.z, bias4.w}, cscale[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
  float pv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pv[i][0] = acc[i].x + (has_bias ? bias[0] : 0.f); pv[i][1] = acc[i].y + (has_bias ? bias[1] : 0.f);
    pv[i][2] = acc[i].z + (has_bias ? bias[2] : 0.f); pv[i][3] = acc[i].w + (has_bias ? bias[3] : 0.f);
  }
  auto each = [&](auto f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) pv[i][e] = f(pv[i][e]);
  };
  switch (a.post_act) {   // one switch around the 16 values, not 16 switches
    case MI355_ACT_LEAKY: each([&](float v) { return v > 0.f ? v : v * a.post_slope; }); break;
    case MI355_ACT_GELU: each([](float v) { return gelu_erf(v); }); break;
    case MI355_ACT_SILU: each([](float v) { return v / (1.0f + expf(-v)); }); break;
    case MI355_ACT_GELU_TANH: each([](float v) { return gelu_tanh(v); }); break;
    case MI355_ACT_ELU: each([](float v) { return v > 0.f ? v : expm1f(v); }); break;
    case MI355_ACT_TANH: each([](float v) { return tanhf(v); }); break;
    default: break;
  }
  float sK[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rv4[4] = {res4[i].x, res4[i].y, res4[i].z, res4[i].w}, ov4[4] = {old4[i].x, old4[i].y, old4[i].z, old4[i].w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float rv = has_res ? rv4[e] : 0.f;
      if (a.accumulate) rv += ov4[e];
      const float v = (pv[i][e] * (has_cs ? cscale[e] : 1.f) + rv) * a.out_scale;
      o[e] = v;
      if (rok[i]) {   // the running statistics of conv_split_finish_kernel, same order
        sK[e] = cnt[e] == 0 ? v : sK[e];
        const float d = v - sK[e];
        s1[e] += d;
        s2[e] += d * d;
        ++cnt[e];
      }
    }
    if (rok[i]) *(float4*)(yb + (int64_t)urow[i] * a.ldy) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (!a.stats_partial) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float cl = (float)cnt[e];
    sst[r16][4 * c4 + e][0] = cl;
    sst[r16][4 * c4 + e][1] = cnt[e] ? sK[e] + s1[e] / cl : 0.f;
    sst[r16][4 * c4 + e][2] = cnt[e] ? s2[e] - s1[e] * s1[e] / cl : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.y * 64 + tid;
  if (tid < 64 && n < a.Cout) {
    float ct = sst[0][tid][0], mt = sst[0][tid][1], vt = sst[0][tid][2];
    for (int j = 1; j < 16; ++j) {
      const float cj = sst[j][tid][0], mj = sst[j][tid][1], vj = sst[j][tid][2];
      if (cj > 0.f) {
        const float cn = ct + cj, dm = mj - mt;
        vt = vt + vj + (ct > 0.f ? dm * dm * ct * cj / cn : 0.f);
        mt = (mt * ct + mj * cj) / cn;
        ct = cn;
      }
    }
    *(float2*)(a.stats_partial + (int64_t)b * a.stats_bstride + ((int64_t)blockIdx.x * a.Cout + n) * 2) = make_float2(mt * ct, vt);
  }
}

// chunk groups for a launch of `wgs` tiles (0 = do not split): enough workgroups for two per CU, at least `min_steps` (chunk, tap) steps per group
int split_groups(const long wgs, const int nchunks, const int K, const long rows_total, const int Cout, const int64_t ws_bytes, const int forced) {
  if (nchunks < 2) return 0;
  static const int off = getenv("MI355_CONV_SPLIT") ? atoi(getenv("MI355_CONV_SPLIT")) : -1;   // 0 = never split (A/B aid)
  if (off == 0 && !forced) return 0;
  static const int min_steps = getenv("MI355_CONV_SPLIT_MINSTEPS") ? atoi(getenv("MI355_CONV_SPLIT_MINSTEPS")) : 4;   // A/B knob.  Round 4, calls 15 / 18: 2 is 0.27 ms faster on the 7.7 ms one-utterance pass (1 is not), but a different grouping between a
  // batch and a lone utterance moves F0 by 5e-6 relative, which is enough to wrap a harmonic phase feature (atan2 at +-pi) in the free-running
  // batch-vs-single tests (tools/diag_batch_single.py): kept at 4 until those tests inject the source
  const int min_chunks = K >= min_steps ? 1 : (min_steps + K - 1) / K;       // >= min_steps (chunk, tap) steps per group
  int ks = forced;                                            // > 0: that many groups; -1: the rule's count whatever the tile count; 0: the rule
  if (ks <= 0) {
    // measured (profiles/r3_bench_codecs_b1_call22): between ~130 and 256 tiles two or three groups do not pay for the slab traffic and the second
    // launch (EnCodec's 512 -> 8 x 256 transposed conv at one utterance: 192 tiles, +19 % with the split) -- split only when at least half the CUs idle
    // ... unless the K loop is deep enough for the cut to pay (Kokoro's first generator stage at one utterance: 166 tiles x 24 / 56 / 88 steps,
    // 79.5 -> 38.5 us with three groups at 88 steps; the same-box A/B of call 27 showed the 192-tile / 32-step case within noise, not slower)
    if (forced == 0 && ((wgs > 128 && !(wgs < 256 && (long)nchunks * K >= 24)) || (long)nchunks * K < 8)) return 0;
    ks = (int)((512 + wgs - 1) / wgs);
    if (ks < 2) ks = 2;
  }
  if (ks > nchunks / min_chunks) ks = nchunks / min_chunks;
  const int64_t per = rows_total * ((Cout + 3) & ~3) * 4;     // one slab of every item
  if (per <= 0 || ws_bytes / per < 2) return 0;
  if (ks > ws_bytes / per) ks = (int)(ws_bytes / per);
  if (ks < 2) return 0;
  const int cpz = (nchunks + ks - 1) / ks;
  ks = (nchunks + cpz - 1) / cpz;                             // groups that actually hold chunks
  return ks >= 2 ? ks : 0;
}

template <int BM, int BN, int PREC>
int launch_split(const mi355_conv_gemm_args& a, hipStream_t st, const int ks) {
  const int R = BM + (a.K - 1) * a.dil;
  const size_t lds = (size_t)R * 64 * a_images<PREC>() + 2 * (BN / 32) * 2048;
  MI355_REQUIRE(lds <= 64 * 1024, "conv_gemm: window too large for LDS (K=%d dil=%d)", a.K, a.dil);
  const int nchunks = (a.Cin + 31) >> 5;
  const split_geom sg{ks, (nchunks + ks - 1) / ks, t_mx_lo_slices};
  const int ldp = (a.Cout + 3) & ~3;
  const int64_t slab = (int64_t)a.Lout * ldp;
  mi355_conv_gemm_args p = a;   // the partial pass: same input side, raw store into slab (b, kz)
  p.y = (float*)a.split_ws;
  p.y_bstride = slab;
  p.ldy = ldp;
  p.bias = nullptr; p.post_act = MI355_ACT_NONE; p.post_colscale = nullptr;
  p.res = nullptr; p.accumulate = 0; p.out_scale = 1.f;
  p.up_s = 0; p.lens_up = nullptr; p.stats_partial = nullptr;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, PREC, true, true>), dim3((a.Lout + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B * ks), dim3(kThreads), lds, st, p, sg);
  MI355_LAUNCH_CHECK("conv_gemm (split)");
  static const bool lean_off = getenv("MI355_CONV_FINISH_OLD") != nullptr && getenv("MI355_CONV_FINISH_OLD")[0] == '1';   // A/B knob
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const bool lean = !lean_off && !a.up_s && a.Cout % 4 == 0 && al16(a.bias) && al16(a.post_colscale) && al16(a.res) && al16(a.y) && al16(a.split_ws) &&
                    a.ldy % 4 == 0 && a.y_bstride % 4 == 0 && (!a.res || (a.ldr % 4 == 0 && a.res_bstride % 4 == 0));
  if (lean) hipLaunchKernelGGL(conv_split_finish_lean_kernel, dim3((a.Lout + 63) / 64, (a.Cout + 63) / 64, a.B), dim3(256), 0, st, a, (const float*)a.split_ws, ks, slab, ldp);
  else hipLaunchKernelGGL(conv_split_finish_kernel, dim3((a.Lout + 63) / 64, (a.Cout + 63) / 64, a.B), dim3(256), 0, st, a, (const float*)a.split_ws, ks, slab, ldp);
  MI355_LAUNCH_CHECK("conv_gemm (split finish)");
  return MI355_OK;
}

template <int BM, int BN>
int launch_split_p(const mi355_conv_gemm_args& a, hipStream_t st, const int ks) {
  switch (a.precision) {
    case 1: return launch_split<BM, BN, 1>(a, st, ks);
    case 3: return launch_split<BM, BN, 3>(a, st, ks);
    case 4: return launch_split<BM, BN, 4>(a, st, ks);
    default: return launch_split<BM, BN, 2>(a, st, ks);
  }
}

}  // namespace

namespace {
bool conv_vec_path(const mi355_conv_gemm_args& a) {
  return (a.flat_valid % 4 == 0) && (a.ldx % 4 == 0) && (a.x_off % 4 == 0) && (a.x_bstride % 4 == 0) && (((uintptr_t)a.x) % 16 == 0);
}
// launches that can write per-block extrema: the quantising-prologue instantiations of the wave-specialised kernel (conv_ws4_fq.hip)
bool conv_ext_supported(const mi355_conv_gemm_args& a) {
  if (!a.pre_fq || (a.precision != 0 && a.precision != 2) || a.up_s != 0 || a.flat_valid != 0 || a.K == 1) return false;
  if (a.pre_act != MI355_ACT_NONE && a.pre_act != MI355_ACT_LEAKY && !(a.pre_act == MI355_ACT_SNAKE && !a.pre_inv_beta)) return false;
  if (a.post_act != MI355_ACT_NONE && a.post_act != MI355_ACT_LEAKY) return false;
  if (a.Cin < (a.Cout <= 64 ? 32 : 64)) return false;
  mi355_conv_gemm_args t = a;
  t.precision = 2;
  return mi355_conv_ws4_eligible(t, conv_vec_path(a));
}
}  // namespace

extern "C" int mi355_conv_gemm_ext_supported(const mi355_conv_gemm_args* ap) { return ap && conv_ext_supported(*ap) ? 1 : 0; }

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
  MI355_REQUIRE(a.precision >= 1 && a.precision <= 6, "conv_gemm: precision must be 1 .. 6");
  MI355_REQUIRE((a.precision != 5 && a.precision != 6) || !a.pre_fq, "conv_gemm: precisions 5 / 6 have no quantising prologue");
  MI355_REQUIRE(!a.pre_fq || a.pre_act == MI355_ACT_NONE || a.pre_act == MI355_ACT_LEAKY || (a.pre_act == MI355_ACT_SNAKE && !a.pre_inv_beta),
                "conv_gemm: a quantising prologue (pre_fq) takes no activation, LeakyReLU or Snake");
  if (a.out_scale == 0.f) a.out_scale = 1.f;
  hipStream_t st = (hipStream_t)stream;
  // 16-byte loads of four channels: aligned rows; a flattened strided conv (flat_valid > 0) qualifies when its run bounds are multiples of 4 too
  static const bool novec_flat = getenv("MI355_CONV_NOVEC_FLAT") != nullptr && getenv("MI355_CONV_NOVEC_FLAT")[0] == '1';   // A/B knob
  const bool vec = (a.flat_valid % 4 == 0) && (a.ldx % 4 == 0) && (a.x_off % 4 == 0) && (a.x_bstride % 4 == 0) &&
                   (((uintptr_t)a.x) % 16 == 0) && !(novec_flat && a.flat_valid > 0);
  if (a.stats_partial) {
    MI355_REQUIRE(a.up_s == 0, "conv_gemm: fused statistics need a plain (non-polyphase) store");
    MI355_REQUIRE(a.stats_bstride % 2 == 0 && ((uintptr_t)a.stats_partial) % 8 == 0, "conv_gemm: stats_partial must be 8-byte aligned");
  }
  int tile = a.tile;
  t_mx_lo_slices = 0;
  if (a.y_split) {
    MI355_REQUIRE(a.y_split == 2 || a.y_split == 4, "conv_gemm: y_split must be 0, 2 (bfloat16 hi | lo) or 4 (IEEE half hi | lo)");
    MI355_REQUIRE(!a.accumulate, "conv_gemm: y_split cannot accumulate into y (y holds split words)");
    a.split_ws = nullptr;   // (the split-K finish kernel stores floats)
  }
  if (a.x_split) {   // pre-split activations: the wave-specialised kernel's producers only (whatever the tile count)
    MI355_REQUIRE(a.x_split == a.precision && (a.precision == 2 || a.precision == 4), "conv_gemm: x_split must equal the launch's precision, 2 or 4");
    MI355_REQUIRE(a.pre_act == MI355_ACT_NONE && !a.pre_scale && !a.pre_fq, "conv_gemm: a split input takes no prologue");
    MI355_REQUIRE(!a.ext_partial && mi355_conv_ws4_eligible(a, vec), "conv_gemm: x_split needs 16-byte aligned channels-last rows and a window of <= 192 rows "
                  "(the wave-specialised kernel)");
    static const int ws_feat_x = getenv("MI355_CONV_WS_FEAT") ? atoi(getenv("MI355_CONV_WS_FEAT")) : 0;
    const int rc = a.Cout <= 64 ? mi355_conv_ws4_launch(a, st, ws_feat_x & 3, 64) : mi355_conv_ws4_launch(a, st, tile % 10000000 == 6128128 ? ((tile / 10000000) % 10) : ws_feat_x);
    return rc;
  }
  if (a.ext_partial) {   // per-block extrema: only the quantising instantiations of the wave-specialised kernel write them -- no other path may take the launch
    MI355_REQUIRE(conv_ext_supported(a), "conv_gemm: ext_partial needs a launch mi355_conv_gemm_ext_supported accepts (pre_fq, precision 2, plain store, K > 1)");
    MI355_REQUIRE(a.ext_bstride % 2 == 0 && ((uintptr_t)a.ext_partial) % 8 == 0, "conv_gemm: ext_partial must be 8-byte aligned");
    static const int ws_feat_e = getenv("MI355_CONV_WS_FEAT") ? atoi(getenv("MI355_CONV_WS_FEAT")) : 0;
    return mi355_conv_ws4_launch(a, st, ws_feat_e & 3, a.Cout <= 64 ? 64 : 128);
  }
  if (a.precision == 5 || a.precision == 6) {
    // the wave-specialised kernel takes the launch when it fills the chip (the same rule as the auto choice below) or when asked for by tile
    // code; everything else runs the precision-4 arithmetic of the 4-wave kernels on the image's fp16 slices
    static const long ws_min5 = getenv("MI355_CONV_WS_MIN_TILES") ? atol(getenv("MI355_CONV_WS_MIN_TILES")) : 128;
    static const bool no_ws5 = getenv("MI355_CONV_NO_WS") != nullptr;
    const long wgs128 = (long)a.B * ((a.Lout + 127) / 128) * ((a.Cout + 127) / 128);
    const bool want = tile % 10000000 == 6128128 || (tile == 0 && !no_ws5 && a.Cout > 64 && wgs128 >= ws_min5 && a.Cin >= 64);
    if (want && mi355_conv_ws4_eligible(a, vec)) {
      const int rc = mi355_conv_ws4_launch(a, st, tile == 0 ? 0 : (((tile / 10000000) % 10) | ((tile / 100000000) << 4)));   // feature / probe / ablation bits
      if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    MI355_REQUIRE(tile % 10000000 != 6128128, "conv_gemm: precisions 5 / 6 on the wave-specialised tile needs K = 3 (mod 4), a plain / LeakyReLU / Snake prologue, "
                  "no epilogue activation beyond LeakyReLU and 16-B aligned channels-last rows");
    t_mx_lo_slices = (a.K + 1) >> 1;
    a.precision = 4;   // (the precision-4 instantiations of the wave-specialised kernel do not know the MX slice layout: mx_image below)
  }
  const bool mx_image = t_mx_lo_slices != 0;
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
    // (shallow inputs, C_in < 64: a tile is one chunk of work and an epilogue -- store bound.  From 2048 tiles on the persistent kernel's interior epilogue wins
    // 2.2x over the 4-wave kernels: Kokoro's 22 -> 128 source conv at 2 M rows 0.60 -> 0.26 ms, profiles/r6_conv_thin_b64_call19.txt)
    if (!no_ws && !mx_image && bn == 128 && wgs128 >= ws_min && (a.Cin >= 64 || wgs128 >= 2048) && mi355_conv_ws4_eligible(a, vec)) {
      const int rc = mi355_conv_ws4_launch(a, st, ws_feat);
      if (rc != MI355_ERR_UNSUPPORTED) return rc;  // no instantiation for this prologue / epilogue pair: fall through to the 4-wave kernels
    }
    // thin outputs (C_out <= 64): the same kernel on 128 x 64 tiles (MI355_CONV_NO_WS64=1 keeps them on the 4-wave 64 x 64 kernel: A/B aid)
    static const bool no_ws64 = getenv("MI355_CONV_NO_WS64") != nullptr;
    if (!no_ws && !no_ws64 && !mx_image && bn == 64 && wgs128 >= ws_min && a.Cin >= 32 && mi355_conv_ws4_eligible(a, vec)) {
      const int rc = mi355_conv_ws4_launch(a, st, ws_feat & 3, 64);
      if (rc != MI355_ERR_UNSUPPORTED) return rc;
    }
    if (bn == 128 && wgs128 >= 512) tile = 64128;  // measured: 64-row tiles beat 128-row tiles on the 4-wave kernel
    if (a.split_ws && vec) {   // few tiles and a deep K loop: cut the chunk range over several workgroups per tile (64-row tiles)
      const long wgs64 = (long)a.B * ((a.Lout + 63) / 64) * ((a.Cout + bn - 1) / bn);
      const int ks = split_groups(wgs64, (a.Cin + 31) >> 5, a.K, (long)a.B * a.Lout, a.Cout, a.split_ws_bytes, 0);
      if (ks) return bn == 128 ? launch_split_p<64, 128>(a, st, ks) : launch_split_p<64, 64>(a, st, ks);
    }
  }
  if (tile % 10000000 == 2064128 || tile % 10000000 == 2064064) {  // split-K, explicit: + 10000000 * groups (0 = as many as the rule gives)
    MI355_REQUIRE(vec && a.split_ws, "conv_gemm: the split-K tiles need the 16-B aligned channels-last input path and a workspace (split_ws)");
    const bool wide = tile % 10000000 == 2064128;
    const long wgs64 = (long)a.B * ((a.Lout + 63) / 64) * ((a.Cout + (wide ? 127 : 63)) / (wide ? 128 : 64));
    const int ks = split_groups(wgs64, (a.Cin + 31) >> 5, a.K, (long)a.B * a.Lout, a.Cout, a.split_ws_bytes, tile / 10000000 ? tile / 10000000 : -1);
    MI355_REQUIRE(ks >= 2, "conv_gemm: nothing to split (C_in %d, K %d, workspace %lld bytes)", a.Cin, a.K, (long long)a.split_ws_bytes);
    return wide ? launch_split_p<64, 128>(a, st, ks) : launch_split_p<64, 64>(a, st, ks);
  }
  if (tile % 10000000 == 6128064) {  // ws4 on 128 x 64 tiles, explicit (+ 10000000 * feature bits 0..3)
    MI355_REQUIRE(!mx_image, "conv_gemm: MX images (precision 5) have no 128 x 64 wave-specialised tile");
    MI355_REQUIRE(mi355_conv_ws4_eligible(a, vec), "conv_gemm: the wave-specialised tile needs 16-B aligned channels-last input rows and a window of <= 192 rows");
    return mi355_conv_ws4_launch(a, st, (tile / 10000000) & 11, 64);
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
