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
// The epilogue is the row-per-register one shared with the 4-wave kernels (conv_common.h): a half wave stores 128 contiguous bytes of one
// output row per instruction.  (Measured, profiles/r2_conv_ab_call1_*.txt: the transposed MFMA orientation -- a lane owning one row and four
// consecutive channels, 16-B loads / stores at a 512-B row pitch -- needs 4x fewer memory instructions but touches 32 cache lines per
// instruction instead of 2; the residual fold got 7-11 % slower and WRITE_SIZE grew 15 % from partially written lines.)
//
// GEMM mode (K == 1, the nn.Linear layers: PL-BERT, LSTM x-projections, 1x1 shortcuts): a "chunk" is 64 channels staged as two 128-row
// blocks of the window and the "taps" walk the blocks, so there are 32 MFMAs per wave between barriers instead of 16.
//
// PREC 5 (fp16 hi pass + block-scaled e4m3 lo pass, conv mode only): the producers keep the lo residual t - fp16(t) as e4m3 bytes (32-byte rows,
// 16-byte halves XOR-swizzled by (row >> 3) & 1) with one E8M0 scale byte per window row; the weight stream of a chunk is K fp16 tap slices followed
// by ceil(K / 2) e4m3 tap-PAIR slices of the same size, so the consumers see ONE stream of 16-register items: K hi items (16 MFMAs of 32x32x16 each)
// and then ceil(K / 2) lo items (4 x v_mfma_scale_f32_32x32x64_f8f6f4: K block b of the instruction = tap 2 p + b) into the same accumulators.
//
// Reference call sites replaced: see include/mi355audio.h (mi355_conv_gemm).
#pragma once
#include "conv_common.h"

namespace mi355conv {

extern int g_ws4_resident;   // conv_ws4.hip; 0 = two workgroups per CU

constexpr int kWs4Threads = 512;

enum { P_NONE = 0, P_LEAKY = 1, P_SNAKE = 2, P_SNAKEBETA = 3, P_ELU = 4 };

struct ws4_geom {
  int tiles_per_item, P, NT, glog, fold;
  int nch;        // chunks per tile (GEMM mode: 64-channel super-chunks)
  int keff;       // taps per chunk (GEMM mode: 2 sub-chunks)
  int tap_rows;   // LDS row shift per tap (conv: dilation; GEMM mode: 128)
  int R;          // window rows (conv: 128 + (K-1)*dil; GEMM mode: 256)
  int gemm;
  int nslices;    // weight slices of the packed image = ceil(Cin / 32) * K
  int bn;         // tile columns: 128, or 64 (the BN = 64 instantiations: C_out <= 64 per tile column, 2 x 2 consumer waves of 64 x 32)
  int feat;       // bit 0: consumers at default priority (A/B aid)
  int total_ids;  // virtual workgroup ids (tile slots incl. the XCD-run padding); a workgroup walks ids blockIdx.x + i * gridDim.x
  unsigned long long* dbg;  // timeline probe buffer (DBG instantiation only)
};

// timeline probe (DBG instantiation only): s_memtime stamps of consumer wave 0 of every 16th workgroup, 4 per tile for its first 8 tiles
constexpr int kDbgSlots = 48;   // 0..31: 4 stamps x 8 tiles, 32..33: wall clock, 34..41: cycles waited at the chunk barriers per tile (PREC 5 probe),
                                 // 42..47: producer wave 4: cycles in convertA / at barriers / in loadA, items, first / last stamp (PREC 5 probe)

struct tile_t { int b, l0, n0, len_out, len_in; };

// Virtual workgroup ids go round-robin over the 8 XCDs (id & 7 = XCD, each with its own L2; gridDim.x is a multiple of 8, so a workgroup's ids all
// map to its own XCD).  An XCD owns runs of 2^glog CONSECUTIVE row tiles (all NT column tiles of a row tile back to back on it): neighbouring
// tiles share their halo rows through that L2 and the column tiles re-read the same activation window from it.
__device__ __forceinline__ bool decode_tile(const mi355_conv_gemm_args& a, const ws4_geom& q, const int id, tile_t& t) {
  const int kq = id >> 3;
  const int ny = kq % q.NT;
  const int pg = kq / q.NT;
  const int glog = q.glog;
  const int p = (((((pg >> glog) << 3) + (id & 7)) << glog)) | (pg & ((1 << glog) - 1));
  if (p >= q.P) return false;
  t.b = p / q.tiles_per_item;
  t.l0 = (p - t.b * q.tiles_per_item) * 128;
  t.n0 = ny * q.bn;
  t.len_out = a.lens_out ? a.lens_out[t.b] : a.Lout;
  if (t.l0 >= t.len_out) return false;
  t.len_in = a.lens_in ? a.lens_in[t.b] : a.Lin;
  return true;
}
// the next id of this workgroup (after `id`) that has work, or -1
__device__ __forceinline__ int next_work(const mi355_conv_gemm_args& a, const ws4_geom& q, int id, tile_t& t) {
  for (id += (int)gridDim.x; id < q.total_ids; id += (int)gridDim.x)
    if (decode_tile(a, q, id, t)) return id;
  return -1;
}

// ABL (ablation bits, timing experiments only -- results are WRONG when non-zero; reachable only through the explicit tile codes
// ABL * 100000000 + 6128128 of tools/bench_conv.py --ablate): 1 = no weight-fragment loads after the first, 2 = no activation-fragment LDS
// reads, 4 = the producers only take part in the barriers, 8 = no residual fold and no epilogue.
// FQ: the prologue ends with the dynamic uint8 fake quantisation of its value (a.pre_fq: the utterance's extrema); the prologue value is then
// conv_common.h's fq_pre_value, the function the extrema pass evaluated.
template <int PREC, int PRE, int EPI, bool GEMM, bool DBG = false, int ABL = 0, int BN = 128, bool FQ = false, bool CW = true>
__global__ __launch_bounds__(kWs4Threads, 4) void conv_ws4_kernel(const mi355_conv_gemm_args a, const ws4_geom q) {
  static_assert(!FQ || (!GEMM && (PRE == P_NONE || PRE == P_LEAKY || PRE == P_SNAKE)), "quantising prologues: conv mode, none / LeakyReLU / Snake");
  constexpr int BM = 128;
  static_assert(BN == 128 || BN == 64, "tile columns");
  constexpr bool MXP = PREC == 5 || PREC == 6;   // fp16 hi pass + block-scaled lo pass on an MX image: e4m3 (5) or FP4 e2m1 (6) elements
  constexpr bool FP4 = PREC == 6;
  static_assert(!MXP || (!GEMM && !FQ && BN == 128), "precisions 5 / 6: conv mode, 128-column tiles");
  static_assert(!FP4 || CW, "precision 6: the column-wave consumer layout only");
  constexpr int NLD = GEMM ? 8 : 6;  // window passes of 32 rows per chunk (conv: R <= 192; GEMM mode: R = 256)
  constexpr int NA = a_images<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane_k = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = q.R;
  // PREC 5 keeps its planes UNSWIZZLED at padded pitches -- 80-byte hi rows, 48-byte lo rows: 20 r mod 64 / 12 r mod 64 walk the 4-bank groups
  // bijectively over any 16 consecutive row residues, so the 16 lanes of a ds_read_b128 group (rows covering all residues mod 16, one 16-byte piece
  // each) never meet -- and an address is then base + row * pitch: one add per read instead of the ~10 VALU operations of the XOR swizzle
  // PREC 6: 16-byte lo rows (32 channels x 4 bits) at their natural pitch: a lane reads ONE row's 16 bytes (its scale block = tap 2 p + lane half), and
  // 16 consecutive rows are 16 distinct 16-byte slots of the 256-byte bank row
  constexpr int HP = MXP ? 80 : 64, LP = FP4 ? 16 : 48;
  const int ABYTES = R * HP;
  const int WBYTES = window_bytes<PREC>(R);  // one staged window: NA 16-bit images (PREC 5: hi image, e4m3 lo image, scale bytes)
  char* Abase = smem;  // [2 buffers][WBYTES]
  const int nch = q.nch, keff = q.keff;
  // PERSISTENT: the workgroup walks its tiles; the (tile, chunk) items form one stream.  Item j is staged in LDS buffer j & 1; one s_barrier per
  // item: the producers arrive when item j is converted, the consumers when they are done with item j - 1 (and with the previous tile's epilogue
  // and the next tile's fold loads), so the window of a tile's first chunk is already waiting when its MFMAs may start.
  tile_t t0;
  int id0 = (int)blockIdx.x - (int)gridDim.x;
  id0 = next_work(a, q, id0, t0);
  if (id0 < 0) return;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------ producers
    const int ptid = tid - 256;
    const int c4 = (ptid & 7) * 4;
    const int prow = ptid >> 3;
    const float* xbase = a.x + a.x_off;
    const int wrow0 = (wave - 4) * 8;  // pass i of this wave covers window rows [wrow0 + 32 i, +8): passes entirely past R are skipped
    constexpr int cstride = GEMM ? 64 : 32;
    float4 s0[NLD], s1[NLD];  // two windows in flight: s0 carries the even items, s1 the odd ones
    // ... and with each window the prologue's per-channel coefficients of ITS chunk (scale, shift, alpha, 1 / beta): loaded inside convertA -- under
    // `if (a.pre_scale)` with constants on the other side, then alpha behind them -- they were two SERIAL L2 round trips (s_waitcnt vmcnt(0) each) at
    // the head of every item, ~1.3 us per chunk and producer wave: more than the consumers need for a 3-tap chunk
    constexpr int NKC = 2 + ((PRE == P_SNAKE || PRE == P_SNAKEBETA) ? 1 : 0) + (PRE == P_SNAKEBETA ? 1 : 0);
    float4 k0[NKC], k1[NKC];
    struct item_t { int id, ci, b, l0, len_in; };
    auto advance = [&](item_t& it) {  // next (tile, chunk) item of this workgroup; id = -1 past the end
      if (it.ci + 1 < nch) { ++it.ci; return; }
      tile_t t;
      it.id = next_work(a, q, it.id, t);
      it.ci = 0; it.b = t.b; it.l0 = t.l0; it.len_in = t.len_in;
    };

    // precision 5: windows that lie inside their utterance in rows and channels take straight-line loads and an unmasked conversion (FASTW)
    // ... and GEMM mode (round 6): the wide linears' producers are bound by VALU ISSUE slots (they get one every ~16 cycles under the consumers' priority),
    // and most of their instructions were row / channel masks and address arithmetic, not the conversion (profiles/r6_conv_big_gemm_split_b64_call13.txt)
    // Same-box A / B (profiles/r6_conv_ab_fastw_b64_call17.txt, r6_conv_big_gemm_fastw_b64_call17.txt): GEMM mode -10 .. -13 %, LeakyReLU convs -6 %, plain
    // convs -1 %; the Snake convs of the 16-bit splits get SLOWER (+2 .. +8 % at 7 / 11 taps: a second copy of the sin / rcp body in the instruction
    // cache), so they keep the one masked body -- as do the quantising prologues (KittenTTS' launches are small)
    constexpr bool FASTW = !FQ && (MXP || GEMM || PRE == P_NONE || PRE == P_LEAKY);
    const bool fastw_off = !MXP && (q.feat & 2);   // feature bit 1 on the 16-bit-split kernels: masked body only (A / B aid; precision 5: the 2 x 2 layout)
    auto interior_window = [&](const item_t& it) {
      if (fastw_off) return false;
      if constexpr (GEMM) return it.l0 + 128 <= it.len_in && it.l0 + 128 <= a.Lin && it.ci * 64 + 64 <= a.Cin;   // wave-uniform
      const int r0 = it.l0 - a.pad;
      return r0 >= 0 && r0 + R <= it.len_in && r0 + R <= a.Lin && it.ci * 32 + 32 <= a.Cin;   // wave-uniform
    };
    // window row r = prow + 32 i of a chunk: conv mode = input row l0 - pad + r, channels [32 chunk, +32);
    // GEMM mode = input row l0 + (r & 127), channels [64 chunk + 32 (r >> 7), +32)
    auto loadA = [&](float4 (&areg)[NLD], float4 (&kreg)[NKC], const item_t& it) {
      const int chunk = it.ci, l0 = it.l0;
      const float* xb = xbase + (int64_t)it.b * a.x_bstride;
      {
        const int c = chunk * cstride + c4;
        const float* const safe = (const float*)a.w;   // an absent operand reads the weight image (valid, 16-byte aligned) and is dropped by a select
        const bool aff = !GEMM && a.pre_scale != nullptr;
        kreg[0] = *(const float4*)(aff ? a.pre_scale + (int64_t)it.b * a.pre_ld + c : safe);
        kreg[1] = *(const float4*)(aff ? a.pre_shift + (int64_t)it.b * a.pre_ld + c : safe);
        if constexpr (PRE == P_SNAKE || PRE == P_SNAKEBETA) kreg[2] = *(const float4*)(a.pre_alpha + c);
        if constexpr (PRE == P_SNAKEBETA) kreg[3] = *(const float4*)(a.pre_inv_beta + c);
      }
      // every pass loads, unconditionally: a pass entirely past the window (wave-uniform) repeats pass 0's address (the same cache lines, never
      // converted).  With the passes under `if (wrow0 + 32 i < R)` the number of loads per window was unknown to the compiler's wait-count
      // bookkeeping, and the wait for window j became s_waitcnt vmcnt(0): it also waited for window j + 1, issued half an item earlier -- one
      // window of prefetch instead of two
      if (FASTW && interior_window(it)) {
        // every row and channel of the window exists (the resblock convs' interior tiles: all but the two edge tiles of an utterance): one wave-uniform
        // base per pass + ONE 32-bit lane offset, no clamps -- ~6 VALU operations per load less in the producers' issue stream (round 6: the
        // producers are starved of VALU issue slots, 16 cycles per instruction under the consumers' priority)
        const char* wb = (const char*)(xb + (int64_t)(GEMM ? l0 : l0 - a.pad) * a.ldx + chunk * cstride);
        const uint32_t loff = ((uint32_t)prow * (uint32_t)a.ldx + (uint32_t)c4) * 4u;
        const size_t pstep = (size_t)a.ldx * 128u;   // 32 rows
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int ii = (GEMM || wrow0 + i * 32 < R) ? i : 0;
          if constexpr (GEMM) areg[i] = *(const float4*)(wb + (ii & 3) * pstep + (ii >> 2) * 128 + loff);   // passes 4..7: the second 32 channels of the super-chunk
          else areg[i] = *(const float4*)(wb + ii * pstep + loff);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int ii = (GEMM || wrow0 + i * 32 < R) ? i : 0;
        int gl = GEMM ? l0 + prow + 32 * (ii & 3) : l0 - a.pad + prow + 32 * ii;
        int c = chunk * cstride + (GEMM ? 32 * (ii >> 2) : 0) + c4;
        gl = gl < 0 ? 0 : (gl >= a.Lin ? a.Lin - 1 : gl);
        if (c >= a.Cin) c = 0;
        areg[i] = *(const float4*)(xb + (int64_t)gl * a.ldx + c);
      }
    };
    auto convertA_body = [&](const float4 (&areg)[NLD], const float4 (&kreg)[NKC], const item_t& it, char* A_hi, auto mask_tag) {
      constexpr bool MASK = decltype(mask_tag)::value;   // false: an interior window (FASTW) -- no row / channel masks
      const int chunk = it.ci, l0 = it.l0, b = it.b, len_in = it.len_in;
      char* A_lo = A_hi + ABYTES;
      const int c = chunk * cstride + c4;  // GEMM mode (no prologue coefficients): passes 4..7 carry channels c + 32
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {1.f, 1.f, 1.f, 1.f}, ial[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (!GEMM) {
        const bool aff = a.pre_scale != nullptr;
        const float4 s4 = kreg[0], h4 = kreg[1];
        sc[0] = aff ? s4.x : 1.f; sc[1] = aff ? s4.y : 1.f; sc[2] = aff ? s4.z : 1.f; sc[3] = aff ? s4.w : 1.f;
        sh[0] = aff ? h4.x : 0.f; sh[1] = aff ? h4.y : 0.f; sh[2] = aff ? h4.z : 0.f; sh[3] = aff ? h4.w : 0.f;
      }
      if constexpr (PRE == P_SNAKE || PRE == P_SNAKEBETA) {
        const float4 a4 = kreg[2];
        al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
        if constexpr (PRE == P_SNAKEBETA) {  // x + sin^2(alpha x) * inv_beta[c]
          const float4 b4 = kreg[3];
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
      // FQ: this utterance's quantiser and the channel coefficients in the form fq_pre_value takes
      const FakeQuant fq = FQ ? FakeQuant(-a.pre_fq[2 * b], a.pre_fq[2 * b + 1]) : FakeQuant(0.f, 0.f);
      fq_coef fk[4];
      if constexpr (FQ) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fk[j] = fq_load_coef(a.pre_scale, a.pre_shift, (int64_t)b * a.pre_ld, PRE == P_SNAKE ? MI355_ACT_SNAKE : MI355_ACT_NONE, a.pre_alpha, c + j);
      }
      const bool chan[4] = {c < a.Cin, c + 1 < a.Cin, c + 2 < a.Cin, c + 3 < a.Cin};   // conv mode: the lane's four channels are the same in every pass
      // a.x_split (wave-uniform): x holds SPLIT words -- hi | lo << 16 of every value, produced once by the layer in front (a conv epilogue, LayerNorm,
      // mi355_split16) -- so the conversion the C_out / 128 column tiles of a row tile would each repeat is two v_perm_b32 per pair of values
      const bool xs = PRE == P_NONE && NA == 2 && !FQ && !MXP && a.x_split != 0;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        if (GEMM || wrow0 + i * 32 < R) {
          const int r = prow + i * 32;
          if (GEMM || r < R) {
            const int gl = GEMM ? l0 + prow + 32 * (i & 3) : l0 - a.pad + r;
            const bool rowok = gl >= 0 && gl < len_in;
            const int cb = c + (GEMM ? 32 * (i >> 2) : 0);
            const float v[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
            if constexpr (PRE == P_NONE && NA == 2 && !FQ && !MXP) {
              if (xs) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (!MASK || (rowok && (cb + j) < a.Cin)) ? __builtin_bit_cast(uint32_t, v[j]) : 0u;
                const int xaddr = r * 64 + ((((c4 >> 3) ^ ((r >> 2) & 3))) << 4) + ((c4 & 4) << 1);
                uint2 xh, xl;   // v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first
                xh.x = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u);
                xh.y = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
                xl.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);
                xl.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
                *(uint2*)(A_hi + xaddr) = xh;
                *(uint2*)(A_lo + xaddr) = xl;
                continue;
              }
            }
            float tt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float u = v[j];
              if constexpr (FQ) {
                u = fq(fq_pre_value(u, fk[j], a.pre_scale != nullptr, PRE == P_SNAKE ? MI355_ACT_SNAKE : (PRE == P_LEAKY ? MI355_ACT_LEAKY : MI355_ACT_NONE), slope));
                tt[j] = (rowok && (cb + j) < a.Cin) ? u : 0.f;
                continue;
              }
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
              if constexpr (GEMM) {
                tt[j] = (!MASK || (rowok && (cb + j) < a.Cin)) ? u : 0.f;
              } else {
                // the prologue value is computed UNCONDITIONALLY (opaque to the optimiser) and masked by one v_cndmask: left to itself hipcc sinks
                // the whole affine + sin chain of every element under its own s_and_saveexec / s_cbranch_execz pair (skipping work for padding rows
                // that almost never occur), i.e. ~10 SALU instructions, two branches and a s_waitcnt per element in the producers' issue stream
                if constexpr (MASK) {
                  asm volatile("" : "+v"(u));
                  tt[j] = (rowok && chan[j]) ? u : 0.f;
                } else {
                  tt[j] = u;
                }
              }
            }
            const int addr = MXP ? r * HP + c4 * 2 : r * 64 + ((((c4 >> 3) ^ ((r >> 2) & 3))) << 4) + ((c4 & 4) << 1);
            uint2 ph;
            float hi[4];
            if constexpr (MXP || PREC == 4) {  // v_cvt_pk_f16_f32 on the clamped value, v_cvt_f32_f16 back
              typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
              typedef float f2_t __attribute__((ext_vector_type(2)));
              float cl[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) cl[j] = __builtin_amdgcn_fmed3f(tt[j], -65504.f, 65504.f);   // one v_med3_f32 instead of min + max
              if constexpr (PREC == 4) {   // the lo image is the residual of the CLAMPED value: finite for every input (a value beyond fp16's range is wrong either way)
#pragma unroll
                for (int j = 0; j < 4; ++j) tt[j] = cl[j];
              }
              const h2_t ha = __builtin_convertvector((f2_t){cl[0], cl[1]}, h2_t), hb = __builtin_convertvector((f2_t){cl[2], cl[3]}, h2_t);
              ph.x = __builtin_bit_cast(uint32_t, ha);
              ph.y = __builtin_bit_cast(uint32_t, hb);
              hi[0] = (float)ha[0]; hi[1] = (float)ha[1]; hi[2] = (float)hb[0]; hi[3] = (float)hb[1];
            } else if constexpr (PREC >= 3) {
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
            if constexpr (PREC == 4) {
              // lo = t - fp16(t) as ONE v_fma_mix_f32 per element (the half operand read in place, see the MX branch below: the product is exact, so
              // this is the subtraction's single rounding), two elements per v_cvt_pk_f16_f32: 12 VALU operations per four elements instead of ~24
              // (round 6: the wide linears are bound by this conversion -- profiles/r6_conv_big_gemm_b64_call11.txt)
              typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
              typedef float f2_t __attribute__((ext_vector_type(2)));
              float l0, l1, l2, l3;
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph.x), "v"(tt[0]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph.x), "v"(tt[1]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(ph.y), "v"(tt[2]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(ph.y), "v"(tt[3]));
              uint2 pl;
              pl.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){l0, l1}, h2_t));
              pl.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){l2, l3}, h2_t));
              *(uint2*)(A_lo + addr) = pl;
            } else if constexpr (NA == 2) {
              uint2 pl;
              pl.x = pack_lo<PREC>(tt[0] - hi[0], tt[1] - hi[1]);
              pl.y = pack_lo<PREC>(tt[2] - hi[2], tt[3] - hi[3]);
              *(uint2*)(A_lo + addr) = pl;
            }
            if constexpr (MXP) {
              // MX block = the 32 channels of this window row (the 8 lanes ptid & 7): shared exponent = floor(log2(max |lo|)) - 7, so the scaled
              // elements stay below 256 < 448 (no saturation); E8M0 byte 0 (2^-127) for an all-zero / denormal-sized row
              // lo = t - fp16(t) as ONE v_fma_mix_f32 per element (the half operand is read in place: no v_cvt_f32_f16 back to fp32 first); the
              // product is exact, so this is the same single rounding as the subtraction
              float l0, l1, l2, l3;   // src0 = the half selected by op_sel[0] of the packed word (op_sel_hi[0] = 1: an f16 source), src1 = -1.0, src2 = t (f32)
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ph.x), "v"(tt[0]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ph.x), "v"(tt[1]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(ph.y), "v"(tt[2]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(ph.y), "v"(tt[3]));
              // max |lo| of the lane's four elements in TWO instructions (source modifiers; left to itself hipcc spends four: it canonicalises each
              // |x| through v_max_f32 |x|, |x| first).  (The DPP read behind it wants two wait states after a VALU write: hipcc's hazard pass counts an
              // asm statement's register definitions and pads with its own s_nop 1 -- checked in the generated code.)
              float m4;
              asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32_e64 %0, %0, |%4|" : "=&v"(m4) : "v"(l0), "v"(l1), "v"(l2), "v"(l3));
              uint32_t mb = __builtin_bit_cast(uint32_t, m4);  // non-negative floats order like unsigned integers
              mb = max(mb, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mb, 0xB1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
              mb = max(mb, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mb, 0x4E, 0xf, 0xf, true));   // quad_perm [2, 3, 0, 1]
              mb = max(mb, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mb, 0x141, 0xf, 0xf, true));  // row_half_mirror: the other quad of the 8 lanes
              // (floor 1, not 0: the scale then is a NORMAL float for the scaled conversion below; a row that small -- padding, lo == 0 -- converts to 0 anyway)
              // PREC 6: the e2m1 grid tops out at 6 = 1.5 x 2^2: shared exponent = floor(log2(max |lo|)) - 2 (OCP MX: a scaled maximum in (6, 8) saturates)
              const int sb = max((int)(mb >> 23) - (FP4 ? 2 : 7), 1);
              // v_cvt_scalef32_pk_fp8_f32 divides both inputs by 2^(exponent of the scale operand - 127) inside the conversion (probed on gfx950:
              // tools/probes/cvt_scale_probe.hip, profiles/r5_cvt_scale_probe_call9.txt: bit-identical to multiplying by the exact reciprocal first):
              // four v_mul_f32 and the 254 - sb arithmetic per four elements less in the producers' issue stream
              typedef short s16x2 __attribute__((ext_vector_type(2)));
              const float scl = __builtin_bit_cast(float, (uint32_t)sb << 23);   // 2^(sb - 127)
              char* A_l8 = A_hi + ABYTES;
              if constexpr (FP4) {
                // v_cvt_scalef32_pk_fp4_f32: src0 -> the low nibble, src1 -> the high nibble of the byte op_sel picks; divides by 2^(exponent of the
                // scale - 127), nearest with ties to the even code, saturating at +-6 (probed: tools/src/mfma_fp4_probe.hip, r6_mfma_fp4_probe_call2.jsonl).
                // The lane's four channels are two bytes: nibble j of a row's 16 bytes = channel j of the chunk
                unsigned p4 = 0;
                p4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(p4, l0, l1, scl, 0);
                p4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(p4, l2, l3, scl, 1);
                *(uint16_t*)(A_l8 + r * LP + (c4 >> 1)) = (uint16_t)p4;
              } else {
                s16x2 p8 = {0, 0};
                p8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(p8, l0, l1, scl, false);
                p8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(p8, l2, l3, scl, true);
                const int pk = __builtin_bit_cast(int, p8);
                *(int*)(A_l8 + r * LP + c4) = pk;
              }
              if ((ptid & 7) == 0) *(uint8_t*)(A_l8 + R * LP + r) = (uint8_t)sb;
            }
          }
        }
      }
    };

    auto convertA = [&](const float4 (&areg)[NLD], const float4 (&kreg)[NKC], const item_t& it, char* A_hi) {
      if constexpr (FASTW) {
        if (interior_window(it)) { convertA_body(areg, kreg, it, A_hi, std::false_type{}); return; }
      }
      convertA_body(areg, kreg, it, A_hi, std::true_type{});
    };

    // chunk ci is converted into buffer ci & 1 while the consumers work on chunk ci - 1 (they left that buffer at the barrier that ended
    // chunk ci - 2); the loads of chunk ci + 2 are issued right behind the conversion, i.e. two windows are always in flight

    item_t ia{id0, 0, t0.b, t0.l0, t0.len_in};
    item_t ib = ia;
    advance(ib);
    constexpr bool work = (ABL & 4) == 0;
    // PREC 5 probe: producer wave 4 of every 16th workgroup sums the cycles it spends converting, at the barriers and issuing loads
    unsigned long long* pdbg = nullptr;
    unsigned long long pc_conv = 0, pc_bar = 0, pc_load = 0, pc_items = 0, pc_first = 0, pc_t = 0;
    if constexpr (DBG && MXP) {
      if (wave == 4 && (blockIdx.x & 15) == 0 && q.dbg) pdbg = q.dbg + (size_t)(blockIdx.x >> 4) * kDbgSlots;
      if (pdbg) pc_first = pc_t = __builtin_amdgcn_s_memtime();
    }
    auto pstamp = [&](unsigned long long& acc_c) {
      if constexpr (DBG && MXP) {
        if (pdbg) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc_c += n - pc_t; pc_t = n; }
      }
    };
    if (work) loadA(s0, k0, ia);
    if (work && ib.id >= 0) loadA(s1, k1, ib);
    pstamp(pc_load);
    while (true) {
      if (work) convertA(s0, k0, ia, Abase);
      if constexpr (DBG && MXP) { if (pdbg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      pstamp(pc_conv);
      item_t na = ib;
      if (na.id >= 0) advance(na);
      if (work) loadA(s0, k0, na.id >= 0 ? na : ia);   // past the last item: the current one again (unconditional: see loadA), never converted
      pstamp(pc_load);
      lds_barrier();  // even item staged
      pstamp(pc_bar);
      ++pc_items;
      if (ib.id < 0) break;
      if (work) convertA(s1, k1, ib, Abase + WBYTES);
      if constexpr (DBG && MXP) { if (pdbg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      pstamp(pc_conv);
      item_t nb = na;
      if (nb.id >= 0) advance(nb);
      if (work) loadA(s1, k1, nb.id >= 0 ? nb : ib);
      pstamp(pc_load);
      lds_barrier();  // odd item staged
      pstamp(pc_bar);
      ++pc_items;
      if (na.id < 0) break;
      ia = na;
      ib = nb;
    }
    if constexpr (DBG && MXP) {
      if (pdbg && lane_k == 0) {
        pdbg[42] = pc_conv; pdbg[43] = pc_bar; pdbg[44] = pc_load; pdbg[45] = pc_items; pdbg[46] = pc_first; pdbg[47] = pc_t;
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------- consumers
  // consumer waves over the tile: 2 x 2 blocks of 64 x (BN / 2) -- or, precision 5, four COLUMN waves of 128 rows x 32 columns (see there)
  constexpr bool COLW = MXP && CW;   // CW = false: the 2 x 2 layout (A / B aid, precision 5 only)
  constexpr int WM = COLW ? 128 : 64, WN = COLW ? 32 : BN / 2, MF = WM / 32, NF = WN / 32;   // BN = 64: every activation fragment feeds one 32-column fragment instead of two
  constexpr int NB = 2 * NF;                                  // weight fragments of a wave per slice: (nf, kk)
  const int wm = COLW ? 0 : wave >> 1, wn = COLW ? wave : wave & 1;
  if (!(q.feat & 1)) __builtin_amdgcn_s_setprio(1);  // MFMA issuers outrank the producers' VALU work (measured +1..8 %)
  const int NTp = ((a.Cout + 127) >> 7) << 2;
  const int64_t wstep = (int64_t)NTp * 2048;
  const int nsteps = nch * keff;
  const int last_slice = q.nslices - 1;
  const int fold = q.fold;
  unsigned long long* dbg = nullptr;
  if constexpr (DBG) {
    if (wave == 0 && (blockIdx.x & 15) == 0 && q.dbg) dbg = q.dbg + (size_t)(blockIdx.x >> 4) * kDbgSlots;
  }
  int jbuf = 0;   // parity of the item stream = LDS buffer of the current chunk
  int ntile = 0;
  tile_t t = t0;
  for (int id = id0; id >= 0; id = next_work(a, q, id, t), ++ntile) {
    const int b = t.b, l0 = t.l0, n0 = t.n0, len_out = t.len_out;
    // an opaque per-tile copy of the lane id: without it LICM hoists the ~200 lane-constant 64-bit fold / store offsets out of the tile loop and
    // parks them in scratch (a persistent kernel's classic: recompute per tile instead, it is a handful of VALU ops)
    int lane = lane_k;
    asm volatile("" : "+v"(lane));
    const int hl = lane & 31, hh = lane >> 5;
    if constexpr (DBG) {
      if (dbg && lane == 0 && ntile < 8) dbg[4 * ntile] = __builtin_amdgcn_s_memtime();  // tile start
      if (dbg && lane == 0 && (ntile == 0 || ntile == 7)) dbg[32 + (ntile == 7)] = wall_clock64();  // 100 MHz constant-rate counter: gives the shader clock
    }
    // fragment (nf, kk) of weight slice s: 1 KB at wfrag + s * wstep + (nf * 2 + kk) * 1024
    const char* wfrag = (const char*)a.w + ((int64_t)((n0 >> 5) + wn * NF)) * 2048 + lane * 16;
    auto wptr = [&](const int s) { return wfrag + (int64_t)(s < last_slice ? s : last_slice) * wstep; };
    bf16x8 b0[NB], b1[NB];
    if constexpr (!MXP) {
#pragma unroll
      for (int f = 0; f < NB; ++f) b0[f] = *(const bf16x8*)(wfrag + f * 1024);
    }

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

    float* yb = a.y + (int64_t)b * a.y_bstride;
    const float* rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
    const bool interior = l0 + BM <= len_out && n0 + BN <= a.Cout;
    // residual and running sum go in as the initial accumulator value
    if ((ABL & 8) == 0 && fold && interior) {  // no clamping, 32-bit lane offsets from wave-uniform bases
      const char* rw = rb ? (const char*)(rb + (int64_t)(l0 + wm * WM) * a.ldr + (n0 + wn * WN)) : nullptr;
      const char* yr = (const char*)(yb + (int64_t)(l0 + wm * WM) * a.ldy + (n0 + wn * WN));
      const uint32_t rpb = (uint32_t)a.ldr * 4u, ypb = (uint32_t)a.ldy * 4u;
      const uint32_t roff = (uint32_t)(4 * hh) * rpb + (uint32_t)hl * 4u;
      const uint32_t yoff = (uint32_t)(4 * hh) * ypb + (uint32_t)hl * 4u;
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
    } else if ((ABL & 8) == 0 && fold) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int n = n0 + wn * WN + nf * 32 + hl;
          const bool nok = n < a.Cout;
          const int ncl = nok ? n : a.Cout - 1;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float rv[8];
            int us[8];
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
              const int r = h * 8 + qq;
              us[qq] = l0 + wm * WM + mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              rv[qq] = 0.f;
            }
            if (rb) {
#pragma unroll
              for (int qq = 0; qq < 8; ++qq) rv[qq] = rb[(int64_t)(us[qq] < len_out ? us[qq] : len_out - 1) * a.ldr + ncl];
            }
            if (a.accumulate) {
              float yv[8];
#pragma unroll
              for (int qq = 0; qq < 8; ++qq) yv[qq] = yb[(int64_t)(us[qq] < len_out ? us[qq] : len_out - 1) * a.ldy + ncl];
#pragma unroll
              for (int qq = 0; qq < 8; ++qq) rv[qq] += yv[qq];
            }
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) acc[mf][nf][h * 8 + qq] = (nok && us[qq] < len_out) ? rv[qq] : 0.f;
          }
        }
    }

    if constexpr (MXP && COLW) {
      // ---- fp16 hi taps, then the e4m3 lo tap pairs of the chunk: one stream of 8-register weight items, one s_barrier per chunk.
      // COLUMN-WAVE layout (round 6): consumer wave w owns ALL 128 rows x the 32 columns [32 w, +32) of the tile = four 32x32 accumulators (mf = 0..3).
      // A weight item is then 8 registers per lane (hi item: the kk 0 and kk 1 fragments of ONE 32-column group; lo item: its 32-byte e4m3 operand)
      // and no two waves of a workgroup load the same fragment: half the L2 -> L1 -> register traffic of the 2 x 2 layout, whose row-pair waves each
      // pulled the same 4 KB per tap (round 6 call 1: the weight loads cost 20 % of this kernel; profiles/r6_conv_ws4_p5_ablation_b64_call1.txt).  The
      // activation fragments (LDS, 4x the L1's bandwidth) are read twice as often instead: 8 per tap and wave, each feeding one MFMA.
      // Per accumulator element the order of additions is the one of the 2 x 2 layout (chunks, taps, kk 0 then kk 1, then the lo pairs): bit-identical.
      // Register plan (128 per lane, 64 of them accumulators): two weight items w0 / w1, four 8-register activation tuples qv[0..3]
      // (hi: group g = the two 32-row fragments mf = 2 (g & 1), + 1 under kk = g >> 1; lo: qv[mf] = the e4m3 operand of rows [32 mf, +32)).
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      const int K = keff, NP = (K + 1) >> 1, dil = q.tap_rows;   // K = 3 (mod 4): K odd, NP even (the dispatcher's eligibility rule)
      // E8M0 scale of this wave's 32-column fragment (one per output column, after the last slice of the image)
      const uint8_t* wsc = (const uint8_t*)a.w + (int64_t)q.nslices * wstep + (n0 + wn * WN) + hl;
      const int bsc = (int)wsc[0];
      // weight item s: a wave-uniform base (SGPR pair) + the lane's 32-bit offset: no 64-bit VALU address arithmetic, no pointer VGPRs
      const char* wtile = (const char*)a.w + ((int64_t)((n0 >> 5) + wn * NF)) * 2048;
      const uint32_t wlane = (uint32_t)lane * 16u;
      auto ldW = [&](i32x8& w, const int s, const bool first = false, const bool lo_item = false) {
        const char* src = wtile + (int64_t)(s < last_slice ? s : last_slice) * wstep;
        if constexpr ((ABL & 1) != 0) {   // timing ablation: only the tile's first weight item is loaded
          if (!first) { asm volatile("" : "+v"(w) : "s"(src)); return; }
        }
        const i32x4 lo4 = *(const i32x4*)(src + wlane);
        if (FP4 && lo_item) {   // an FP4 lo item is ONE kilobyte (16 bytes per lane: the 32 channels of tap 2 p + lane half for the lane's column)
          w = __builtin_shufflevector(lo4, lo4, 0, 1, 2, 3, 4, 5, 6, 7);
          return;
        }
        const i32x4 hi4 = *(const i32x4*)(src + 1024 + wlane);
        w = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      i32x8 w0, w1;
      ldW(w0, 0, true);
      if constexpr ((ABL & 1) != 0) w1 = w0;
      i32x8 qv[4];
      int sv[4] = {0, 0, 0, 0};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[g][e] = 0;
      // per-item opaque copies of the lane coordinates: LICM otherwise parks partly computed LDS addresses in VGPRs across the whole tile and
      // the allocator pays for them with an accumulator in scratch; recomputing them costs a handful of VALU operations per 8 MFMAs
      int hlx = hl, hhx = hh;
      int lbH = 0, lbL = 0, lbS = 0;   // the lane's byte offsets into the hi / lo planes and the scale bytes; everything else of an address is wave-uniform
      auto fresh = [&]() {
        asm volatile("" : "+v"(hlx), "+v"(hhx));
        lbH = hlx * HP + hhx * 16;
        if constexpr (FP4) {   // a lane reads the whole 16-byte row under ITS tap: lane half 1 sits one tap (dil rows) further down
          lbL = (hlx + hhx * dil) * LP;
          lbS = hlx + hhx * dil;
        } else {
          lbL = hlx * LP + hhx * 16;
          lbS = hlx;
        }
      };
      auto lo4 = [](const i32x8& v) { return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(v, v, 0, 1, 2, 3)); };
      auto hi4 = [](const i32x8& v) { return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(v, v, 4, 5, 6, 7)); };
      // fragments of group g under tap tp: kk = g >> 1 (16-channel half of the chunk), rows [32 m0, +32) and [32 (m0 + 1), +32), m0 = 2 (g & 1)
      auto rdH = [&](const int g, const int tp) {
        const int kk = g >> 1, m0 = 2 * (g & 1);
        const int uoff = jbuf * WBYTES + (m0 * 32 + tp * dil) * HP + kk * 32;   // wave-uniform
        int la = lbH;
        asm volatile("" : "+v"(la));   // one add per group, recomputed: a shared (hoisted) address per (group, tap) costs a register each
        i32x4 v0, v1;
        if constexpr ((ABL & 2) != 0) { v0 = (i32x4){la, la, la, la}; v1 = v0; }   // timing ablation: no LDS read
        else {
          v0 = *(const i32x4*)(Abase + (la + uoff));
          v1 = *(const i32x4*)(Abase + (la + uoff) + 32 * HP);
        }
        qv[g] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      // lo operand of rows [mf * 32, +32) for tap pair p: bytes 0..15 = channels [16 hh, +16) of the row under tap 2 p, bytes 16..31 = the same
      // channels under tap 2 p + 1 (K is odd: the last pair takes its last tap twice, that half of the weight item is zero, the data stay
      // finite); scale byte: lanes 0..31 carry K block 0 (tap 2 p), lanes 32..63 K block 1
      auto rdL = [&](const int mf, const int p) {
        const int t1 = 2 * p + 1 < K ? 2 * p + 1 : K - 1;
        const int u0 = mf * 32 + 2 * p * dil, u1 = mf * 32 + t1 * dil;          // wave-uniform row offsets of the two taps
        const int ubase = jbuf * WBYTES + ABYTES;
        int la = lbL, ls = lbS;
        asm volatile("" : "+v"(la), "+v"(ls));
        i32x4 x0, x1;
        int sc;
        if constexpr (FP4) {
          // ONE 16-byte read: the lane's row under tap 2 p + lane half (fresh() put the half's tap shift into the lane offset; the odd tap
          // count's last pair takes its last tap twice: there the shift is taken back, the weight half is zero)
          const int back = (2 * p + 1 < K) ? 0 : dil;   // wave-uniform
          if (back) { la -= hhx * back * LP; ls -= hhx * back; }
          x0 = *(const i32x4*)(Abase + (la + (ubase + u0 * LP)));
          x1 = x0;
          sc = (int)*(const uint8_t*)(Abase + (ls + (ubase + R * LP + u0)));
        } else if constexpr ((ABL & 2) != 0) {
          x0 = (i32x4){la, la, la, la}; x1 = x0; sc = 120 + (ls & 7);   // timing ablation: no LDS read
        } else {
          x0 = *(const i32x4*)(Abase + (la + (ubase + u0 * LP)));
          x1 = *(const i32x4*)(Abase + (la + (ubase + u1 * LP)));
          sc = (int)*(const uint8_t*)(Abase + (ls + (ubase + R * LP + (hhx ? u1 : u0))));
        }
        qv[mf] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        sv[mf] = sc;
      };
      auto mmH = [&](const int g, const i32x8& w) {
        const int kk = g >> 1, m0 = 2 * (g & 1);
        const bf16x8 bfrag = kk ? hi4(w) : lo4(w);
        acc[m0][0] = mfma16<PREC>(lo4(qv[g]), bfrag, acc[m0][0]);
        acc[m0 + 1][0] = mfma16<PREC>(hi4(qv[g]), bfrag, acc[m0 + 1][0]);
      };
      auto mmL = [&](const int mf, const i32x8& w) {
        // (inline asm, accumulator tied in place: with the builtin hipcc picks the three-address form under register pressure -- D in 16 OTHER
        // registers -- and then shuffles whole accumulators through scratch.  Hazards, which hipcc does not pad for an asm statement: a VALU
        // result as an operand (the scale byte's mask) wants two wait states = the leading s_nop 1; the next reader of D is always another
        // MFMA taking it whole as C (0 wait states) until the pad in front of the fold / epilogue below.)
        if constexpr (FP4) {   // 4-register operands, cbsz / blgp 4 = e2m1 on both sides: 8 passes instead of 16
          const i32x4 a4 = __builtin_shufflevector(qv[mf], qv[mf], 0, 1, 2, 3), b4 = __builtin_shufflevector(w, w, 0, 1, 2, 3);
          asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(acc[mf][0]) : "v"(a4), "v"(b4), "v"(sv[mf]), "v"(bsc));
        } else {
          asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc[mf][0]) : "v"(qv[mf]), "v"(w), "v"(sv[mf]), "v"(bsc));
        }
      };
      // hi tap tp on weight item w (its two fragments are in flight or landed); `last`: the chunk's last hi tap requests the operands of the
      // first lo tap pair instead of the next tap's fragments
      auto hi_tap = [&](const int tp, const i32x8& w, auto last) {
        constexpr bool LAST = decltype(last)::value;
        fresh();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          mmH(g, w);
          if constexpr (LAST) rdL(g, 0);
          else rdH(g, tp + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // lo tap pair p on weight item w
      auto lo_pair = [&](const int p, const i32x8& w, const bool more) {
        fresh();
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          mmL(mf, w);
          if (more) rdL(mf, p + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto prefetch = [&](i32x8& w, const int s, const bool lo_item = false) {
        ldW(w, s, false, lo_item);
        asm volatile("" ::: "memory");  // keep the prefetch AHEAD of the MFMAs
        __builtin_amdgcn_sched_barrier(0);
      };
      // One chunk: K hi items, then NP lo items; wa holds item s on entry, wb holds the next chunk's first item on exit (K odd, NP even: the
      // roles of the two register sets swap from chunk to chunk, hence the two instances below).
      int s = 0;
      unsigned long long bar_wait = 0;
      bool first_chunk = true;
      auto chunk_body = [&](i32x8& wa, i32x8& wb) {
        if constexpr (DBG) {
          unsigned long long tb0 = 0;
          if (dbg) tb0 = __builtin_amdgcn_s_memtime();
          lds_barrier();
          if (dbg) {
            const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
            bar_wait += tb1 - tb0;
            if (first_chunk && lane == 0 && ntile < 8) dbg[4 * ntile + 1] = tb1;  // first window staged
            first_chunk = false;
          }
        } else {
          lds_barrier();  // the chunk's window is staged behind this barrier (and the producers may refill the buffer just left)
        }
        fresh();
        rdH(0, 0);
        rdH(1, 0);
        rdH(2, 0);
        rdH(3, 0);
        int tp = 0;
        for (; tp + 1 < K; tp += 2) {
          prefetch(wb, s + 1);
          hi_tap(tp, wa, std::false_type{});
          prefetch(wa, s + 2);
          hi_tap(tp + 1, wb, std::false_type{});
          s += 2;
        }
        prefetch(wb, s + 1, true);                    // the chunk's first lo item
        hi_tap(tp, wa, std::true_type{});
        s += 1;
        if constexpr (FP4) {
          // lo items two at a time (NP is even); the LAST two are peeled so that every prefetch knows at compile time whether it fetches a lo item (one
          // kilobyte per wave under precision 6) or the next chunk's first hi item: a run-time choice between one and two loads would cost the
          // compiler's wait-count bookkeeping its precision (it then waits for everything in flight)
          int p = 0;
          for (; p + 2 < NP; p += 2) {
            prefetch(wa, s + 1, true);
            lo_pair(p, wb, true);
            prefetch(wb, s + 2, true);
            lo_pair(p + 1, wa, true);
            s += 2;
          }
          prefetch(wa, s + 1, true);
          lo_pair(p, wb, true);
          prefetch(wb, s + 2, false);                   // the next chunk's first hi item (past the image's end: the last slice again, never used)
          lo_pair(p + 1, wa, false);
          s += 2;
        } else {
          for (int p = 0; p < NP; p += 2) {
            prefetch(wa, s + 1);
            lo_pair(p, wb, true);
            prefetch(wb, s + 2);
            lo_pair(p + 1, wa, p + 2 < NP);
            s += 2;
          }
        }
        jbuf ^= 1;
      };
      for (int c = 0; c < nch; c += 2) {
        chunk_body(w0, w1);
        if (c + 1 >= nch) break;
        chunk_body(w1, w0);
      }
      if constexpr (DBG) {
        if (dbg && lane == 0 && ntile < 8) dbg[34 + ntile] = bar_wait;
      }
      // the tile's last MFMAs are asm statements: 16 passes -> 19 wait states before anything but an MFMA may touch their D
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]));
    } else if constexpr (PREC == 5) {
      // ---- (the 2 x 2 consumer layout of rounds 4 - 5, kept selectable for same-box A / B: tile code feature bit 1) fp16 hi taps, then the e4m3 lo tap pairs of the chunk: one stream of 16-register weight items, one s_barrier per chunk.
      // Register plan (128 per lane, 64 of them accumulators): two weight items as 8-register tuples w[nf] (hi item: low half = the kk 0 fragment,
      // high half = kk 1; lo item: the 32-byte e4m3 operand of column fragment nf), ONE 8-register activation set `qa` shared by the two phases
      // (hi: its low half is fragment set 0; lo: the whole e4m3 operand) and a 4-register second hi set.
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      // An "a" constraint anywhere in the kernel makes hipcc split the 128 registers into 64 VGPRs + 64 AGPRs and keep the MFMA accumulators in
      // the AGPR half: two bins (4 x 16 accumulator registers | the 8-register operand tuples), instead of one in which the tuples fragment.
      // { int agpr_hint; asm volatile("" : "=a"(agpr_hint)); }
      const int K = keff, NP = (K + 1) >> 1, dil = q.tap_rows;   // K = 3 (mod 4): K odd, NP even (the dispatcher's eligibility rule)
      // E8M0 scales of this wave's two 32-column fragments (one per output column, after the last slice of the image): bytes 0 / 1 = nf 0 / 1
      const uint8_t* wsc = (const uint8_t*)a.w + (int64_t)q.nslices * wstep + (n0 + wn * WN) + hl;
      const int bsc = (int)wsc[0] | ((int)wsc[32] << 8);
      // weight item s: a wave-uniform base (SGPR pair) + the lane's 32-bit offset: no 64-bit VALU address arithmetic, no pointer VGPRs
      const char* wtile = (const char*)a.w + ((int64_t)((n0 >> 5) + wn * NF)) * 2048;
      const uint32_t wlane = (uint32_t)lane * 16u;
      auto ldW = [&](i32x8 (&w)[2], const int s, const bool first = false) {
        const char* src = wtile + (int64_t)(s < last_slice ? s : last_slice) * wstep;
        if constexpr ((ABL & 1) != 0) {   // timing ablation: only the tile's first weight item is loaded
          if (!first) { asm volatile("" : "+v"(w[0]), "+v"(w[1]) : "s"(src)); return; }
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const i32x4 lo4 = *(const i32x4*)(src + (2 * nf) * 1024 + wlane), hi4 = *(const i32x4*)(src + (2 * nf + 1) * 1024 + wlane);
          w[nf] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      };
      i32x8 w0[2], w1[2];
      ldW(w0, 0, true);
      if constexpr ((ABL & 1) != 0) { w1[0] = w0[0]; w1[1] = w0[1]; }
      // Activation operands: two 8-register tuples.  hi phase: four fragment sets (qa low / high half, qb low / high half = groups 0..3 of a tap), each
      // refilled with the NEXT tap's fragment right behind the MFMAs that read it: a read is four groups (8 MFMAs) ahead of its use.  lo phase:
      // qa = the e4m3 operand of rows 0..31 of the wave's block, qb = rows 32..63, each refilled for the next tap pair behind its two MFMAs.
      i32x8 qa, qb;
      int sa = 0, sb2 = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) { qa[e] = 0; qb[e] = 0; }
      // per-item opaque copies of the lane coordinates: LICM otherwise parks ~12 partly computed LDS addresses in VGPRs across the whole tile and
      // the allocator pays for them with an accumulator in scratch; recomputing them costs a handful of VALU operations per 8 MFMAs
      int hlx = hl, hhx = hh;
      int lbH = 0, lbL = 0, lbS = 0;   // the lane's byte offsets into the hi / lo planes and the scale bytes; everything else of an address is wave-uniform
      auto fresh = [&]() {
        asm volatile("" : "+v"(hlx), "+v"(hhx));
        lbH = (wm * WM + hlx) * HP + hhx * 16;
        lbL = (wm * WM + hlx) * LP + hhx * 16;
        lbS = wm * WM + hlx;
      };
      auto lo4 = [](const i32x8& v) { return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(v, v, 0, 1, 2, 3)); };
      auto hi4 = [](const i32x8& v) { return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(v, v, 4, 5, 6, 7)); };
      auto set_lo = [](i32x8& q8, const i32x4 v) { q8 = __builtin_shufflevector(v, (i32x4)__builtin_shufflevector(q8, q8, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7); };
      auto set_hi = [](i32x8& q8, const i32x4 v) { q8 = __builtin_shufflevector((i32x4)__builtin_shufflevector(q8, q8, 0, 1, 2, 3), v, 0, 1, 2, 3, 4, 5, 6, 7); };
      // fragment of group g (kk = g >> 1: 16-channel half of the chunk, mf = g & 1: 32-row half of the wave's rows) under tap tp
      auto rdH = [&](const int g, const int tp) {
        const int kk = g >> 1, mf = g & 1;
        const int uoff = jbuf * WBYTES + (mf * 32 + tp * dil) * HP + kk * 32;   // wave-uniform
        int la = lbH;
        asm volatile("" : "+v"(la));   // one add per read, recomputed: a shared (hoisted) address per (group, tap) costs a register each
        i32x4 v;
        if constexpr ((ABL & 2) != 0) v = (i32x4){la, la, la, la};   // timing ablation: no LDS read
        else v = *(const i32x4*)(Abase + (la + uoff));
        if (g == 0) set_lo(qa, v);
        else if (g == 1) set_hi(qa, v);
        else if (g == 2) set_lo(qb, v);
        else set_hi(qb, v);
      };
      // lo operand of rows [mf * 32, +32) for tap pair p: bytes 0..15 = channels [16 hh, +16) of the row under tap 2 p, bytes 16..31 = the same
      // channels under tap 2 p + 1 (K is odd: the last pair takes its last tap twice, that half of the weight item is zero, the data stay
      // finite); scale byte: lanes 0..31 carry K block 0 (tap 2 p), lanes 32..63 K block 1
      auto rdL = [&](const int mf, const int p) {
        const int t1 = 2 * p + 1 < K ? 2 * p + 1 : K - 1;
        const int u0 = mf * 32 + 2 * p * dil, u1 = mf * 32 + t1 * dil;          // wave-uniform row offsets of the two taps
        const int ubase = jbuf * WBYTES + ABYTES;
        int la = lbL, ls = lbS;
        asm volatile("" : "+v"(la), "+v"(ls));
        i32x4 x0, x1;
        int sc;
        if constexpr ((ABL & 2) != 0) {
          x0 = (i32x4){la, la, la, la}; x1 = x0; sc = 120 + (ls & 7);   // timing ablation: no LDS read
        } else {
          x0 = *(const i32x4*)(Abase + (la + (ubase + u0 * LP)));
          x1 = *(const i32x4*)(Abase + (la + (ubase + u1 * LP)));
          sc = (int)*(const uint8_t*)(Abase + (ls + (ubase + R * LP + (hhx ? u1 : u0))));
        }
        if (mf == 0) { qa = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7); sa = sc; }
        else { qb = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7); sb2 = sc; }
      };
      auto mmH = [&](const int g, const i32x8 (&w)[2]) {
        const int kk = g >> 1, mf = g & 1;
        const bf16x8 h = g == 0 ? lo4(qa) : (g == 1 ? hi4(qa) : (g == 2 ? lo4(qb) : hi4(qb)));
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16<PREC>(h, kk ? hi4(w[nf]) : lo4(w[nf]), acc[mf][nf]);
      };
      auto mmL = [&](const int mf, const i32x8 (&w)[2]) {
        // (inline asm, accumulator tied in place: with the builtin hipcc picks the three-address form under register pressure -- D in 16 OTHER
        // registers -- and then shuffles whole accumulators through scratch.  Hazards, which hipcc does not pad for an asm statement: a VALU
        // result as an operand (the scale byte's mask) wants two wait states = the leading s_nop 1; the next reader of D is always another
        // MFMA taking it whole as C (0 wait states) until the pad in front of the fold / epilogue below.)
        if (mf == 0) {
          asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc[0][0]) : "v"(qa), "v"(w[0]), "v"(sa), "v"(bsc));
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,0]" : "+v"(acc[0][1]) : "v"(qa), "v"(w[1]), "v"(sa), "v"(bsc));
        } else {
          asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc[1][0]) : "v"(qb), "v"(w[0]), "v"(sb2), "v"(bsc));
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,0]" : "+v"(acc[1][1]) : "v"(qb), "v"(w[1]), "v"(sb2), "v"(bsc));
        }
      };
      // hi tap tp on weight item w (its four fragments are in flight or landed); `last`: the chunk's last hi tap requests the two operands of the
      // first lo tap pair instead of the next tap's fragments
      auto hi_tap = [&](const int tp, const i32x8 (&w)[2], auto last) {
        constexpr bool LAST = decltype(last)::value;
        fresh();
        mmH(0, w);
        if constexpr (!LAST) rdH(0, tp + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmH(1, w);
        if constexpr (LAST) rdL(0, 0);
        else rdH(1, tp + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmH(2, w);
        if constexpr (!LAST) rdH(2, tp + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmH(3, w);
        if constexpr (LAST) rdL(1, 0);
        else rdH(3, tp + 1);
        __builtin_amdgcn_sched_barrier(0);
      };
      // lo tap pair p on weight item w
      auto lo_pair = [&](const int p, const i32x8 (&w)[2], const bool more) {
        fresh();
        mmL(0, w);
        if (more) rdL(0, p + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmL(1, w);
        if (more) rdL(1, p + 1);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto prefetch = [&](i32x8 (&w)[2], const int s) {
        ldW(w, s);
        asm volatile("" ::: "memory");  // keep the prefetch AHEAD of the MFMAs
        __builtin_amdgcn_sched_barrier(0);
      };
      // One chunk: K hi items, then NP lo items; wa holds item s on entry, wb holds the next chunk's first item on exit (K odd, NP even: the
      // roles of the two register sets swap from chunk to chunk, hence the two instances below).
      int s = 0;
      unsigned long long bar_wait = 0;
      bool first_chunk = true;
      auto chunk_body = [&](i32x8 (&wa)[2], i32x8 (&wb)[2]) {
        if constexpr (DBG) {
          unsigned long long tb0 = 0;
          if (dbg) tb0 = __builtin_amdgcn_s_memtime();
          lds_barrier();
          if (dbg) {
            const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
            bar_wait += tb1 - tb0;
            if (first_chunk && lane == 0 && ntile < 8) dbg[4 * ntile + 1] = tb1;  // first window staged
            first_chunk = false;
          }
        } else {
          lds_barrier();  // the chunk's window is staged behind this barrier (and the producers may refill the buffer just left)
        }
        fresh();
        rdH(0, 0);
        rdH(1, 0);
        rdH(2, 0);
        rdH(3, 0);
        int tp = 0;
        for (; tp + 1 < K; tp += 2) {
          prefetch(wb, s + 1);
          hi_tap(tp, wa, std::false_type{});
          prefetch(wa, s + 2);
          hi_tap(tp + 1, wb, std::false_type{});
          s += 2;
        }
        prefetch(wb, s + 1);
        hi_tap(tp, wa, std::true_type{});
        s += 1;
        for (int p = 0; p < NP; p += 2) {
          prefetch(wa, s + 1);
          lo_pair(p, wb, true);
          prefetch(wb, s + 2);
          lo_pair(p + 1, wa, p + 2 < NP);
          s += 2;
        }
        jbuf ^= 1;
      };
      for (int c = 0; c < nch; c += 2) {
        chunk_body(w0, w1);
        if (c + 1 >= nch) break;
        chunk_body(w1, w0);
      }
      if constexpr (DBG) {
        if (dbg && lane == 0 && ntile < 8) dbg[34 + ntile] = bar_wait;
      }
      // the tile's last MFMAs are asm statements: 16 passes -> 19 wait states before anything but an MFMA may touch their D
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    } else {
      int tap = 0;
      bf16x8 ah0 = b0[0], al0 = b0[1], ah1 = b0[0], al1 = b0[1];  // two static activation-fragment sets (al*: the lo image, unused for the single-pass precisions)
      // group g of a tap: kk = g >> 1 (16-channel half of the chunk), mf = g & 1 (32-row half of the wave's rows)
      auto rdA = [&](bf16x8& h, bf16x8& l, const int g, const int tp) {
        const int kk = g >> 1, mf = g & 1;
        const int row = wm * WM + mf * 32 + hl + tp * q.tap_rows;
        const int cidx = kk * 2 + hh;
        const int addr = row * 64 + ((cidx ^ ((row >> 2) & 3)) << 4);
        const char* A_hi = Abase + jbuf * WBYTES;
        if constexpr ((ABL & 2) != 0) {
          asm volatile("" : "+v"(h), "+v"(l) : "v"(addr));  // opaque: no LDS read
        } else {
          h = *(const bf16x8*)(A_hi + addr);
          if constexpr (NA == 2) l = *(const bf16x8*)(A_hi + ABYTES + addr);
        }
      };
      auto mm = [&](const bf16x8& h, const bf16x8& l, const int g, const bf16x8 (&bf)[NB]) {
        const int kk = g >> 1, mf = g & 1;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16<PREC>(h, bf[2 * nf + kk], acc[mf][nf]);
        if constexpr (NA == 2) {
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16<PREC>(l, bf[2 * nf + kk], acc[mf][nf]);
        }
      };
      // one tap = four groups; the fragments of group g+1 are requested before the MFMAs of group g are issued
      auto tap_body = [&](const bf16x8 (&bf)[NB], const bool first) {
        if (tap == 0) {  // new chunk: its window is staged behind this barrier (and the producers may refill the buffer just left)
          lds_barrier();
          if constexpr (DBG) {
            if (first && dbg && lane == 0 && ntile < 8) dbg[4 * ntile + 1] = __builtin_amdgcn_s_memtime();  // first window staged
          }
          rdA(ah0, al0, 0, 0);
        }
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
        if (++tap == keff) {  // end of chunk
          tap = 0;
          jbuf ^= 1;
        }
      };
      // The weight prefetch is unconditional (past the last slice it re-reads the last one): a branch around the loads would make hipcc assume
      // they may not have been issued and wait for them right away.
      for (int s = 0; s < nsteps; s += 2) {
        const char* w1 = wptr(s + 1);
#pragma unroll
        for (int f = 0; f < NB; ++f) {
          if constexpr ((ABL & 1) != 0) { b1[f] = b0[f]; asm volatile("" : "+v"(b1[f]) : "v"(w1)); }
          else b1[f] = *(const bf16x8*)(w1 + f * 1024);
        }
        asm volatile("" ::: "memory");  // keep the prefetch AHEAD of the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        tap_body(b0, s == 0);
        if (s + 1 >= nsteps) break;
        const char* w0 = wptr(s + 2);
#pragma unroll
        for (int f = 0; f < NB; ++f) {
          if constexpr ((ABL & 1) != 0) { b0[f] = b1[f]; asm volatile("" : "+v"(b0[f]) : "v"(w0)); }
          else b0[f] = *(const bf16x8*)(w0 + f * 1024);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tap_body(b1, false);
      }
    }
    if constexpr (DBG) {
      if (dbg && lane == 0 && ntile < 8) dbg[4 * ntile + 2] = __builtin_amdgcn_s_memtime();  // main loop done
    }

    if constexpr ((ABL & 8) != 0) {
      float tsum = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int r = 0; r < 16; ++r) tsum += acc[mf][nf][r];
      if (tsum == 1.2345e-30f) yb[0] = tsum;  // keeps the MFMAs alive without an epilogue
      continue;
    }
    const bool plain = a.up_s == 0 && a.post_act == MI355_ACT_NONE && (fold || (!a.res && !a.accumulate)) && !a.post_colscale && !a.y_split;
    if constexpr (COLW) {
      // a column wave holds two 64-row statistics blocks: the epilogue of the 2 x 2 layout, once per half (accumulators 2 h, 2 h + 1 as its block h)
      if (plain && interior) {
        conv_epilogue_interior<2, NF, 64, WN, FQ, MF, 0>(a, acc, b, l0, n0, 0, wn, lane);
        conv_epilogue_interior<2, NF, 64, WN, FQ, MF, 2>(a, acc, b, l0, n0, 1, wn, lane);
      } else {
        conv_epilogue<2, NF, 64, WN, EPI, FQ, MF, 0>(a, acc, b, l0, n0, 0, wn, lane, len_out, fold != 0);
        conv_epilogue<2, NF, 64, WN, EPI, FQ, MF, 2>(a, acc, b, l0, n0, 1, wn, lane, len_out, fold != 0);
      }
    } else {
      bool up_interior = false;
      if constexpr (EPI == 0 && !FQ && BN == 128 && PRE == P_LEAKY && !GEMM) {   // polyphase store (the conv_transpose upsamplers behind a LeakyReLU): tiles whose rows all land inside the output
        if (a.up_s != 0 && interior && a.post_act == MI355_ACT_NONE && !a.accumulate && !a.post_colscale && !a.y_split && !a.stats_partial && a.res_shift == 0 &&
            (a.up_cout & 31) == 0) {
          const int len_up = a.lens_up ? a.lens_up[b] : a.up_Lout;
          up_interior = (int64_t)l0 * a.up_s - a.up_p >= 0 && (int64_t)(l0 + BM - 1) * a.up_s + (a.up_s - 1) - a.up_p < len_up;
        }
      }
      if (plain && interior) conv_epilogue_interior<MF, NF, WM, WN, FQ>(a, acc, b, l0, n0, wm, wn, lane);   // FQ instantiations: + per-block extrema (EXT)
      else if (up_interior) conv_epilogue_up_interior<MF, NF, WM, WN>(a, acc, b, l0, n0, wm, wn, lane);
      else conv_epilogue<MF, NF, WM, WN, EPI, FQ>(a, acc, b, l0, n0, wm, wn, lane, len_out, fold != 0);
    }
    if constexpr (DBG) {
      if (dbg && lane == 0 && ntile < 8) dbg[4 * ntile + 3] = __builtin_amdgcn_s_memtime();  // stores issued
    }
  }
  if (!(q.feat & 1)) __builtin_amdgcn_s_setprio(0);
}

// GEMM mode: pure linear layers (K == 1, no prologue) with at least two 32-channel chunks
inline bool gemm_mode(const mi355_conv_gemm_args& a) { return a.K == 1 && a.Cin >= 64 && a.pre_act == MI355_ACT_NONE && !a.pre_scale && !a.pre_fq; }

template <int PREC, int PRE, int EPI, bool GEMM, bool DBG = false, int ABL = 0, int BN = 128, bool FQ = false, bool CW = true>
int launch_ws4(const mi355_conv_gemm_args& a, hipStream_t st, const int feat, unsigned long long* dbg = nullptr) {
  ws4_geom q;
  q.bn = BN;
  q.gemm = GEMM ? 1 : 0;
  const int chunks32 = (a.Cin + 31) >> 5;
  q.nslices = chunks32 * ((PREC == 5 || PREC == 6) ? a.K + ((a.K + 1) >> 1) : a.K);
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
  const size_t lds = (size_t)2 * window_bytes<PREC>(q.R);
  q.tiles_per_item = (a.Lout + 127) / 128;
  q.P = a.B * q.tiles_per_item;
  q.NT = (a.Cout + BN - 1) / BN;
  q.fold = ((a.res || a.accumulate) && a.post_act == MI355_ACT_NONE && a.up_s == 0 && a.res_shift == 0 && !a.post_colscale) ? 1 : 0;
  // runs of 2^glog consecutive row tiles per XCD: long runs share halos in L2, but every XCD must still get several rounds of runs
  q.glog = q.P >= 512 ? 3 : (q.P >= 256 ? 2 : (q.P >= 128 ? 1 : 0));
  q.feat = feat;
  q.dbg = dbg;
  const int per = 8 << q.glog;
  q.total_ids = ((q.P + per - 1) / per) * per * q.NT;
  // persistent: two resident workgroups per CU (128 VGPRs x 8 waves each) walk the tile list; small problems launch one workgroup per tile
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  static const int wg_per_cu = getenv("MI355_CONV_WS_WG_PER_CU") ? atoi(getenv("MI355_CONV_WS_WG_PER_CU")) : 2;
  // mi355_conv_ws4_resident(n): the next launches take at most n persistent workgroups (a share of the chip: several convs on separate streams)
  const int resident = g_ws4_resident > 0 ? ((g_ws4_resident + 7) / 8) * 8 : ((cus * wg_per_cu) / 8) * 8;
  const unsigned grid = (unsigned)((feat & 8) || q.total_ids <= resident ? q.total_ids : resident);  // feat bit 3: one workgroup per tile (A/B aid)
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL((conv_ws4_kernel<PREC, PRE, EPI, GEMM, DBG, ABL, BN, FQ, CW>), dim3(grid), dim3(kWs4Threads), lds, st, a, q);
  MI355_LAUNCH_CHECK("conv_gemm(ws4)");
  return MI355_OK;
}

// epilogue family of a launch: 0 = none / LeakyReLU, 1 = GELU (erf), 2 = SiLU, 3 = GELU-tanh; -1 = ELU / tanh epilogues (the one-column final
// convs of DAC / SNAC: never this kernel's shapes, they stay on the 4-wave kernels).  One instantiation per activation: the transcendental
// bodies are inlined 64 times per wave, and a kernel carrying all of them no longer fits the instruction cache.
inline int epi_family(const mi355_conv_gemm_args& a) {
  switch (a.post_act) {
    case MI355_ACT_NONE:
    case MI355_ACT_LEAKY: return 0;
    case MI355_ACT_GELU: return 1;
    case MI355_ACT_SILU: return 2;
    case MI355_ACT_GELU_TANH: return 3;
  }
  return -1;
}
inline int pre_kind(const mi355_conv_gemm_args& a) {
  switch (a.pre_act) {
    case MI355_ACT_NONE: return P_NONE;
    case MI355_ACT_LEAKY: return P_LEAKY;
    case MI355_ACT_SNAKE: return a.pre_inv_beta ? P_SNAKEBETA : P_SNAKE;
    case MI355_ACT_ELU: return P_ELU;
  }
  return -1;
}

#define WS4_CASE(PREC, PRE, EPI) \
  if (bn == 128 && pre == PRE && epi == EPI && !gemm) return launch_ws4<PREC, PRE, EPI, false>(a, st, feat)
#define WS4_GEMM(PREC, EPI) \
  if (bn == 128 && epi == EPI && gemm) return launch_ws4<PREC, P_NONE, EPI, true>(a, st, feat)
// the 64-column tile (bn == 64: asked for by the dispatcher when the last 128-column tile would be at most half full)
#define WS4_CASE_N64(PREC, PRE, EPI) \
  if (bn == 64 && pre == PRE && epi == EPI && !gemm) return launch_ws4<PREC, PRE, EPI, false, false, 0, 64>(a, st, feat & 15)
#define WS4_GEMM_N64(PREC, EPI) \
  if (bn == 64 && epi == EPI && gemm) return launch_ws4<PREC, P_NONE, EPI, true, false, 0, 64>(a, st, feat & 15)

}  // namespace mi355conv

// per-precision instantiation sets (one translation unit each: they compile in parallel); MI355_ERR_UNSUPPORTED = no such instantiation
int mi355_conv_ws4_p2(const mi355_conv_gemm_args& a, hipStream_t st, int feat, unsigned long long* dbg, int bn);
int mi355_conv_ws4_p4(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn);
int mi355_conv_ws4_p13(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn);
int mi355_conv_ws4_fq(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn);   // quantising prologues (pre_fq), precision 2
int mi355_conv_ws4_p5(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn);   // fp16 hi + MX e4m3 lo (conv mode, 128-column tiles)
int mi355_conv_ws4_p6(const mi355_conv_gemm_args& a, hipStream_t st, int feat, int bn);   // fp16 hi + MX FP4 lo (conv mode, 128-column tiles)
int mi355_conv_ws4_p5_probe(const mi355_conv_gemm_args& a, hipStream_t st, int feat, unsigned long long* dbg);   // its probe / ablation builds
