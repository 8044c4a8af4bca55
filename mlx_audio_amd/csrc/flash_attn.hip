// Scaled-dot-product attention for the transformer blocks under mlx_audio/stt and mlx_audio/tts (gfx950).
//
// Replaces (reference call sites):
//   - Whisper MultiHeadAttention.qkv_attention: q@k, + causal mask, softmax(precise=True), w@v
//     (stt/models/whisper/whisper.py:371-385) for the encoder (1500 x 1500), the decoder prefill and decode steps;
//   - mx.fast.scaled_dot_product_attention with GQA + additive causal mask + KVCache
//     (tts/models/qwen3_tts/talker.py:307, speech_tokenizer.py transformer, codec/models/mimi/modules/transformer.py:109,
//     lm/models/llama.py / sesame/attention.py for CSM).
//
// Two kernels, both exact-f32 arithmetic (activations in this library are fp32, SURVEY section 8 header):
//
//   flash_attn_kernel<DH>  (Tq > 8): one workgroup = 128 queries x one head, 4 waves x 32 queries.  K / V stream
//     through LDS in 64-key (DH = 64) / 32-key (DH = 128) stages (register prefetch of the next stage under the MFMAs).  Both contractions run on
//     v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation, S^T = K Q^T and O^T += V^T P^T, so that a lane owns ONE
//     query column (l & 31) of every accumulator: the online-softmax max / sum / rescale are per-lane scalars plus a
//     single xor-32 exchange between the two half-waves (which hold interleaved key rows of the same query), and the
//     probabilities feed the second MFMA straight from the accumulator registers -- with the key order of step s
//     chosen as the C-layout row order ((s&3) + 8*(s>>2) + 4*(lane>>5)) no data movement is needed at all.
//     LDS rows are padded to DH+1 floats: the A-operand ds_read_b32 of both phases is bank-conflict free.
//
//   attn_decode_kernel<DH> (Tq <= 8, the KV-cache decode step): one workgroup per (query, head, item); its 4 (or, for
//     more than 256 keys, 16) waves take interleaved 64-key chunks, lanes own keys for q.k (float4 row reads) and own channels for p.V, online
//     softmax per wave, merged through LDS at the end.  An MFMA tile would be 31/32 idle here; this path is bound by
//     reading the KV cache once.
//
// Visibility of key j for query i of item b (len_q / len_k = valid rows, queries are the LAST len_q positions):
//   k_start <= j < len_k,  causal: j <= i + (len_k - len_q),  window W > 0: j > i + (len_k - len_q) - W.
// Invisible keys get probability exactly 0 (the reference adds -1e9 / -inf style masks: identical after softmax).
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __forceinline__ void wave_lds_sync2() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr float kLog2e = 1.4426950408889634f;

// K / V element types: KVT 0 = float32, 1 = bfloat16, 2 = float16 (mi355_flash_attn_args.kv_dtype = MI355_KV_F32 / MI355_KV_BF16 / MI355_KV_F16).  A 16-bit cache
// is what the reference keeps (K / V live in the checkpoint dtype: whisper.py:360-361, lm/models/cache.py:104-176) and halves the bytes of the
// decode-step attention, which is bound by reading the cache once.  Strides are in ELEMENTS for every type.
template <int KVT> struct kv_t { using type = float; };
template <> struct kv_t<1> { using type = uint16_t; };
template <> struct kv_t<2> { using type = uint16_t; };

template <int KVT>
__device__ __forceinline__ float kv_f32(const uint16_t u) {
  if constexpr (KVT == 1) return __builtin_bit_cast(float, (uint32_t)u << 16);
  else return (float)__builtin_bit_cast(_Float16, u);
}

// the value a cache of this type would hand back (round to nearest even into the 16-bit type and back; identity for float32)
template <int KVT>
__device__ __forceinline__ float kv_round(const float v) {
  if constexpr (KVT == 0) return v;
  else if constexpr (KVT == 1) return kv_f32<1>((uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu));
  else return kv_f32<2>((uint16_t)(pack_f16x2(v, 0.f) & 0xffffu));
}

// four consecutive elements at p (16-byte aligned for float32, 8-byte aligned for the 16-bit types)
template <int KVT>
__device__ __forceinline__ float4 kv_load4(const typename kv_t<KVT>::type* p) {
  if constexpr (KVT == 0) {
    return *(const float4*)p;
  } else {
    const uint2 u = *(const uint2*)p;
    return make_float4(kv_f32<KVT>((uint16_t)(u.x & 0xffffu)), kv_f32<KVT>((uint16_t)(u.x >> 16)), kv_f32<KVT>((uint16_t)(u.y & 0xffffu)),
                       kv_f32<KVT>((uint16_t)(u.y >> 16)));
  }
}

template <int DH, int KVT>
__global__ __launch_bounds__(256) void flash_attn_kernel(const mi355_flash_attn_args a) {
  using kvp = const typename kv_t<KVT>::type*;
  constexpr int KB = DH == 64 ? 64 : 32;  // keys per LDS stage (two padded fp32 tiles must fit 64 KB of static LDS)
  constexpr int LD = DH + 1;   // padded LDS row, floats
  constexpr int NDB = DH / 32; // 32-channel blocks of the output
  constexpr int NLD = (KB * DH / 4) / 256;  // float4 loads per thread per stage and tensor
  __shared__ float Ks[KB * LD];
  __shared__ float Vs[KB * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int g = h / (a.heads / a.kv_heads);
  const int len_q = a.lens_q ? a.lens_q[b] : a.Tq;
  const int len_k = a.lens_k ? a.lens_k[b] : a.Tk;
  const int q0 = blockIdx.x * 128;
  if (q0 >= len_q || len_k <= 0) return;
  const int qoff = len_k - len_q;
  const int qi = q0 + wave * 32 + (lane & 31);
  const bool wave_active = (q0 + wave * 32) < len_q;
  const int qic = qi < len_q ? qi : len_q - 1;
  const int qpos = qic + qoff;  // absolute position of this lane's query

  // Q fragment: B operand of S^T = K Q^T; step s needs Q[q][2s + half] (pre-scaled, log2 domain)
  float qreg[DH / 2];
  {
    const float* qrow = a.q + (int64_t)b * a.q_bstride + (int64_t)qic * a.ldq + h * DH;
    const float sc = a.scale * kLog2e;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) {
      const float2 t = *(const float2*)(qrow + 2 * s);
      qreg[s] = (half ? t.y : t.x) * sc;
    }
  }

  // key range this workgroup needs
  int kend = len_k, kbeg = 0;
  if (a.causal) {
    const int last_q = (q0 + 127 < len_q ? q0 + 127 : len_q - 1) + qoff;
    kend = last_q + 1 < len_k ? last_q + 1 : len_k;
    if (kend < 1) kend = 1;
  }
  if (a.window > 0) {
    kbeg = q0 + qoff - a.window + 1;
    if (kbeg < 0) kbeg = 0;
  }
  const int kstart = a.k_start ? a.k_start[b] : 0;  // left-padded rows: keys before k_start[b] are padding
  if (kstart > kbeg) kbeg = kstart;
  kbeg &= ~(KB - 1);

  // heads packed inside a row (g * DH) or head-major planes (k_hstride: a head's keys contiguous -- the layout for long key ranges:
  // with rows of 2 * heads * DH floats every key of one head sits 6-8 KB from the next and lands on the same one or two L2 channels)
  kvp kbase = (kvp)a.k + (int64_t)b * a.k_bstride + (a.k_hstride ? (int64_t)g * a.k_hstride : (int64_t)g * DH);
  kvp vbase = (kvp)a.v + (int64_t)b * a.v_bstride + (a.v_hstride ? (int64_t)g * a.v_hstride : (int64_t)g * DH);

  float4 kpre[NLD], vpre[NLD];
  auto prefetch = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid;
      const int row = e / (DH / 4), c4 = e % (DH / 4);
      int j = kb + row;
      j = j < len_k ? j : len_k - 1;  // clamp: finite data, masked below
      kpre[i] = kv_load4<KVT>(kbase + (int64_t)j * a.ldk + c4 * 4);
      vpre[i] = kv_load4<KVT>(vbase + (int64_t)j * a.ldv + c4 * 4);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid;
      const int row = e / (DH / 4), c4 = e % (DH / 4);
      float* kd = Ks + row * LD + c4 * 4;
      float* vd = Vs + row * LD + c4 * 4;
      kd[0] = kpre[i].x; kd[1] = kpre[i].y; kd[2] = kpre[i].z; kd[3] = kpre[i].w;
      vd[0] = vpre[i].x; vd[1] = vpre[i].y; vd[2] = vpre[i].z; vd[3] = vpre[i].w;
    }
  };

  f32x16 o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  prefetch(kbeg);
  for (int kb = kbeg; kb < kend; kb += KB) {
    __syncthreads();  // everyone is done reading the previous stage
    commit();
    __syncthreads();
    if (kb + KB < kend) prefetch(kb + KB);
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < KB / 32; ++sub) {
      const int kb32 = kb + sub * 32;
      if (kb32 >= kend) break;
      // ---- S^T block (32 keys x 32 queries)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* krow = Ks + (sub * 32 + (lane & 31)) * LD + half;
#pragma unroll
      for (int s = 0; s < DH / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * s], qreg[s], acc, 0, 0, 0);
      // ---- mask + online softmax (per-lane query)
      float bm = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kb32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bool vis = j < len_k && j >= kstart;
        if (a.causal) vis = vis && j <= qpos;
        if (a.window > 0) vis = vis && j > qpos - a.window;
        acc[r] = vis ? acc[r] : -INFINITY;
        bm = fmaxf(bm, acc[r]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float m_new = fmaxf(m, bm);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = exp2f(m - m_safe);  // m = -inf -> 0
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = exp2f(acc[r] - m_safe);  // -inf -> 0
        ps += acc[r];
      }
      lsum = lsum * alpha + ps;
      m = m_new;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      // ---- O^T += V^T P^T : step s contracts keys (s&3) + 8*(s>>2) + 4*half, which is where acc[s] lives
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float* vrow = Vs + (sub * 32 + (s & 3) + 8 * (s >> 2) + 4 * half) * LD + (lane & 31);
#pragma unroll
        for (int d = 0; d < NDB; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[d * 32], acc[s], o[d], 0, 0, 0);
      }
    }
  }

  if (!wave_active) return;
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  if (qi < len_q) {
    float* orow = a.out + (int64_t)b * a.out_bstride + (int64_t)qi * a.ldo + h * DH;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = make_float4(o[d][c * 4] * inv, o[d][c * 4 + 1] * inv, o[d][c * 4 + 2] * inv, o[d][c * 4 + 3] * inv);
        *(float4*)(orow + d * 32 + 8 * c + 4 * half) = t;
      }
  }
}

// ---------------------------------------------------------------------------------------------------- 16-bit K / V on the 16-bit matrix pipe
// flash_attn16_kernel<DH, KVT>: the prefill / encoder kernel when K and V are held in the checkpoint's 16-bit type.  Same tiling, visibility
// rule and online softmax as flash_attn_kernel, but both contractions run on v_mfma_f32_32x32x16_{bf16,f16} (16 k per instruction at half the
// issue cost of the fp32 32x32x2: 16x the MACs per cycle):
//   * K tile [64 keys][DH] goes global -> LDS untouched (16-byte pieces; rows padded by 16 bytes so the A-operand ds_read_b128 is conflict free);
//     V goes in TRANSPOSED ([DH][64 keys]) because the second contraction is over keys: its A operand needs 8 keys of one channel per lane.
//   * Q (fp32, pre-scaled) is split once into hi + lo images of the K / V type and lives in registers as the B operand of S^T = K Q^T; the
//     probabilities come out of the accumulator in C layout, are split hi + lo and are already in B-operand order for O^T += V^T P^T when the
//     keys of a step are taken in C-row order {0..3, 8..11} + 4 * (lane >> 5) (+16 for the second step) -- the V^T fragment is read in that order.
//   So K and V enter exactly (they ARE 16-bit), Q and P carry ~16 (bf16) / ~22 (fp16) mantissa bits: fp32-grade results at 4 MFMAs per product pair.
template <bool F16>
__device__ __forceinline__ f32x16 mfma32_16(const uint4 a, const uint4 b, const f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ void split_pair(const float x, const float y, uint32_t& hi, uint32_t& lo) {
  if constexpr (F16) {
    // round-toward-zero pack (one v_cvt_pkrtz_f16_f32 for two values): hi need not be the NEAREST half, only a half whose remainder the lo
    // image can hold -- x - hi is exact in fp32 and |x - hi| < 1 ulp_half(x), so hi + lo carries the same ~22 bits
    typedef __attribute__((ext_vector_type(2))) __fp16 h2;
    const h2 hv = __builtin_amdgcn_cvt_pkrtz(x, y);
    hi = __builtin_bit_cast(uint32_t, hv);
    const h2 lv = __builtin_amdgcn_cvt_pkrtz(x - (float)hv[0], y - (float)hv[1]);
    lo = __builtin_bit_cast(uint32_t, lv);
  } else {
    hi = pack_bf16x2(x, y);
    const float hx = __builtin_bit_cast(float, hi << 16), hy = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16x2(x - hx, y - hy);
  }
}

template <int DH, int KVT>
__global__ __launch_bounds__(256) void flash_attn16_kernel(const mi355_flash_attn_args a) {
  constexpr bool F16 = KVT == 2;
  constexpr int KB = 64;            // keys per LDS stage
  constexpr int KLD = DH + 8;       // K row stride (halves): 144 / 272 bytes
  constexpr int VLD = KB + 8;       // V^T row stride (halves): 144 bytes
  constexpr int NDB = DH / 32;      // 32-channel blocks of the output
  constexpr int NST = DH / 16;      // k steps of S^T = K Q^T
  constexpr int NLD = (KB * DH / 8) / 256;  // 16-byte pieces per thread per stage and tensor
  __shared__ __attribute__((aligned(16))) uint16_t Ks[KB * KLD];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[DH * VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g2 = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int g = h / (a.heads / a.kv_heads);
  const int len_q = a.lens_q ? a.lens_q[b] : a.Tq;
  const int len_k = a.lens_k ? a.lens_k[b] : a.Tk;
  const int q0 = blockIdx.x * 128;
  if (q0 >= len_q || len_k <= 0) return;
  const int qoff = len_k - len_q;
  const int qi = q0 + wave * 32 + (lane & 31);
  const bool wave_active = (q0 + wave * 32) < len_q;
  const int qic = qi < len_q ? qi : len_q - 1;
  const int qpos = qic + qoff;

  // Q fragments (B operand): step s holds Q[q][16 s + 8 g2 .. + 8], pre-scaled into the log2 domain, as hi + lo images
  uint4 qh[NST], ql[NST];
  {
    const float* qrow = a.q + (int64_t)b * a.q_bstride + (int64_t)qic * a.ldq + h * DH;
    const float sc = a.scale * kLog2e;
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const float4 t0 = *(const float4*)(qrow + 16 * s + 8 * g2), t1 = *(const float4*)(qrow + 16 * s + 8 * g2 + 4);
      split_pair<F16>(t0.x * sc, t0.y * sc, qh[s].x, ql[s].x);
      split_pair<F16>(t0.z * sc, t0.w * sc, qh[s].y, ql[s].y);
      split_pair<F16>(t1.x * sc, t1.y * sc, qh[s].z, ql[s].z);
      split_pair<F16>(t1.z * sc, t1.w * sc, qh[s].w, ql[s].w);
    }
  }

  int kend = len_k, kbeg = 0;
  if (a.causal) {
    const int last_q = (q0 + 127 < len_q ? q0 + 127 : len_q - 1) + qoff;
    kend = last_q + 1 < len_k ? last_q + 1 : len_k;
    if (kend < 1) kend = 1;
  }
  if (a.window > 0) {
    kbeg = q0 + qoff - a.window + 1;
    if (kbeg < 0) kbeg = 0;
  }
  const int kstart = a.k_start ? a.k_start[b] : 0;
  if (kstart > kbeg) kbeg = kstart;
  kbeg &= ~(KB - 1);

  const uint16_t* kbase = (const uint16_t*)a.k + (int64_t)b * a.k_bstride + (a.k_hstride ? (int64_t)g * a.k_hstride : (int64_t)g * DH);
  const uint16_t* vbase = (const uint16_t*)a.v + (int64_t)b * a.v_bstride + (a.v_hstride ? (int64_t)g * a.v_hstride : (int64_t)g * DH);

  uint4 kpre[NLD], vpre[NLD];
  auto prefetch = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid;
      const int row = e / (DH / 8), c8 = e % (DH / 8);
      int j = kb + row;
      j = j < len_k ? j : len_k - 1;  // clamp: finite data, masked below
      kpre[i] = *(const uint4*)(kbase + (int64_t)j * a.ldk + c8 * 8);
      vpre[i] = *(const uint4*)(vbase + (int64_t)j * a.ldv + c8 * 8);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = i * 256 + tid;
      const int row = e / (DH / 8), c8 = e % (DH / 8);
      *(uint4*)(Ks + row * KLD + c8 * 8) = kpre[i];
      uint16_t* vd = Vt + (c8 * 8) * VLD + row;
      vd[0 * VLD] = (uint16_t)(vpre[i].x & 0xffffu); vd[1 * VLD] = (uint16_t)(vpre[i].x >> 16);
      vd[2 * VLD] = (uint16_t)(vpre[i].y & 0xffffu); vd[3 * VLD] = (uint16_t)(vpre[i].y >> 16);
      vd[4 * VLD] = (uint16_t)(vpre[i].z & 0xffffu); vd[5 * VLD] = (uint16_t)(vpre[i].z >> 16);
      vd[6 * VLD] = (uint16_t)(vpre[i].w & 0xffffu); vd[7 * VLD] = (uint16_t)(vpre[i].w >> 16);
    }
  };

  f32x16 o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  prefetch(kbeg);
  for (int kb = kbeg; kb < kend; kb += KB) {
    __syncthreads();  // everyone is done reading the previous stage
    commit();
    __syncthreads();
    if (kb + KB < kend) prefetch(kb + KB);
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < KB / 32; ++sub) {
      const int kb32 = kb + sub * 32;
      if (kb32 >= kend) break;
      // ---- S^T block (32 keys x 32 queries): A = K rows, B = Q hi / lo
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const uint16_t* krow = Ks + (sub * 32 + (lane & 31)) * KLD + 8 * g2;
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        const uint4 kf = *(const uint4*)(krow + 16 * s);
        acc = mfma32_16<F16>(kf, qh[s], acc);
        acc = mfma32_16<F16>(kf, ql[s], acc);
      }
      // ---- mask + online softmax (per-lane query).  The VALU work of this section, not the MFMAs, is what bounds the kernel at DH = 64
      // (~10 lane-instructions per (query, key) against one MFMA cycle), so everything that is not needed on an interior block is skipped
      // wave-uniformly: the visibility mask (only blocks that touch len_k / k_start / the causal diagonal / the window edge), the rescale of
      // O (only when some lane's running maximum moved), and exp2 is the bare v_exp_f32 (arguments <= 0: no range handling needed).
      const int wq0 = q0 + wave * 32 + qoff;   // positions of this wave's first / last query
      const int wq1 = (q0 + wave * 32 + 31 < len_q ? q0 + wave * 32 + 31 : len_q - 1) + qoff;
      const bool interior = kb32 + 32 <= len_k && kb32 >= kstart && (!a.causal || kb32 + 31 <= wq0) && (a.window <= 0 || kb32 > wq1 - a.window);
      float bm = -INFINITY;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, acc[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kb32 + (r & 3) + 8 * (r >> 2) + 4 * g2;
          bool vis = j < len_k && j >= kstart;
          if (a.causal) vis = vis && j <= qpos;
          if (a.window > 0) vis = vis && j > qpos - a.window;
          acc[r] = vis ? acc[r] : -INFINITY;
          bm = fmaxf(bm, acc[r]);
        }
      }
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float m_new = fmaxf(m, bm);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] = __builtin_amdgcn_exp2f(acc[r] - m_safe);  // -inf -> 0
        ps += acc[r];
      }
      if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {   // some query's maximum moved: rescale (wave-uniform branch)
        const float alpha = __builtin_amdgcn_exp2f(m - m_safe);  // m = -inf -> 0
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      } else {
        lsum += ps;
      }
      m = m_new;
      // ---- O^T += V^T P^T: step s2 contracts the keys 16 s2 + {0..3, 8..11} + 4 g2, which is where acc[8 s2 .. 8 s2 + 7] live
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 ph, pl;
        split_pair<F16>(acc[8 * s2 + 0], acc[8 * s2 + 1], ph.x, pl.x);
        split_pair<F16>(acc[8 * s2 + 2], acc[8 * s2 + 3], ph.y, pl.y);
        split_pair<F16>(acc[8 * s2 + 4], acc[8 * s2 + 5], ph.z, pl.z);
        split_pair<F16>(acc[8 * s2 + 6], acc[8 * s2 + 7], ph.w, pl.w);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const uint16_t* vrow = Vt + (d * 32 + (lane & 31)) * VLD + sub * 32 + 16 * s2 + 4 * g2;
          const uint2 v0 = *(const uint2*)vrow, v1 = *(const uint2*)(vrow + 8);
          const uint4 vf = make_uint4(v0.x, v0.y, v1.x, v1.y);
          o[d] = mfma32_16<F16>(vf, ph, o[d]);
          o[d] = mfma32_16<F16>(vf, pl, o[d]);
        }
      }
    }
  }

  if (!wave_active) return;
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  if (qi < len_q) {
    float* orow = a.out + (int64_t)b * a.out_bstride + (int64_t)qi * a.ldo + h * DH;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = make_float4(o[d][c * 4] * inv, o[d][c * 4 + 1] * inv, o[d][c * 4 + 2] * inv, o[d][c * 4 + 3] * inv);
        *(float4*)(orow + d * 32 + 8 * c + 4 * g2) = t;
      }
  }
}

template <int DH, int NW, int KVT>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const mi355_flash_attn_args a) {
  using kvp = const typename kv_t<KVT>::type*;
  constexpr int ND = DH / 64;  // channels per lane in the p.V phase
  __shared__ float qs[DH];
  __shared__ float ps[NW][64];
  __shared__ float red_m[NW], red_l[NW];
  __shared__ float red_o[NW][DH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsplit = a.nsplit > 1 ? a.nsplit : 1;  // key range split over nsplit workgroups (partials merged by the last one to finish)
  const int qi = blockIdx.x / nsplit, sp = blockIdx.x - qi * nsplit, h = blockIdx.y, b = blockIdx.z;
  const int g = h / (a.heads / a.kv_heads);
  const int len_q = a.lens_q ? a.lens_q[b] : a.Tq;
  const int len_k = a.lens_k ? a.lens_k[b] : a.Tk;
  if (qi >= len_q) return;
  const int qpos = qi + (len_k - len_q);
  // Fused decode step (new_k != null; Tq == 1): q arrives RAW from the q|k|v projection and gets its per-head RMSNorm + rotary embedding here;
  // the new position's k and v arrive raw in new_k / new_v (NOT in the cache), k gets the same treatment, both enter the softmax as one more
  // online-softmax update after the cached keys, and the first query head of each kv group stores them into the cache row Tk - 1 -- which
  // nobody reads in this launch.  One launch less per layer than projection -> head_norm_rope -> attention (talker.py:264-307).
  __shared__ float ks_new[DH], vs_new[DH];
  const bool fused = a.new_k != nullptr;
  const float qsc = a.scale * kLog2e;
  // The per-head norm weights and the rotary table entries of this lane's pair are requested HERE, with the raw q | k | v loads, and
  // unconditionally (a missing operand reads the q row instead -- valid memory -- and is dropped by a select): behind the first barrier they were
  // two more dependent L2 round trips of a launch whose whole work is a few of them (one decode step of CSM's depth decoder: 8 workgroups).
  constexpr int half_dh = DH / 2;
  const bool pf_act = lane < half_dh;
  const int pf_i0 = pf_act ? (a.rope_mode == 1 ? 2 * lane : lane) : 0, pf_i1 = pf_act ? (a.rope_mode == 1 ? 2 * lane + 1 : lane + half_dh) : 0;
  const float* const pf_dummy = a.q + (int64_t)b * a.q_bstride + (int64_t)qi * a.ldq + h * DH;
  const bool pf_has_qnw = fused && a.q_norm_w != nullptr, pf_has_knw = fused && a.k_norm_w != nullptr, pf_has_rope = fused && a.rope_cos != nullptr;
  int pf_pos = 0;
  if (pf_has_rope) {
    pf_pos = (a.lens_k ? len_k - 1 : a.rope_pos) - (a.k_start ? a.k_start[b] : 0);   // slot caches: every item is at its own position
    pf_pos = pf_pos < 0 ? 0 : (pf_pos >= a.rope_rows ? a.rope_rows - 1 : pf_pos);
  }
  const float pf_qnw0 = (pf_has_qnw ? a.q_norm_w : pf_dummy)[pf_i0], pf_qnw1 = (pf_has_qnw ? a.q_norm_w : pf_dummy)[pf_i1];
  const float pf_knw0 = (pf_has_knw ? a.k_norm_w : pf_dummy)[pf_i0], pf_knw1 = (pf_has_knw ? a.k_norm_w : pf_dummy)[pf_i1];
  const float pf_cos = (pf_has_rope ? a.rope_cos + (int64_t)pf_pos * half_dh : pf_dummy)[pf_act ? lane : 0];
  const float pf_sin = (pf_has_rope ? a.rope_sin + (int64_t)pf_pos * half_dh : pf_dummy)[pf_act ? lane : 0];
  if (!fused) {
    for (int t = tid; t < DH; t += NW * 64) qs[t] = a.q[(int64_t)b * a.q_bstride + (int64_t)qi * a.ldq + h * DH + t] * qsc;
  } else {
    // NT elements per thread (2 for the one-wave workgroup at DH = 128), every load of a stage requested before the first is used; the scales /
    // biases of an fp8 / biased projection come unconditionally too (an absent one reads q and is dropped by a select)
    constexpr int NT = (DH + NW * 64 - 1) / (NW * 64);
    int64_t iq[NT], ik[NT];
    int tt[NT];
    float q0[NT], k0[NT], v0[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = tid + j * NW * 64;
      tt[j] = t < DH ? t : DH - 1;
      iq[j] = (int64_t)b * a.q_bstride + (int64_t)qi * a.ldq + h * DH + tt[j];
      ik[j] = (int64_t)b * a.new_bstride + g * DH + tt[j];
      q0[j] = a.q[iq[j]]; k0[j] = a.new_k[ik[j]]; v0[j] = a.new_v[ik[j]];
    }
    const bool has_ws = a.q_wscale != nullptr, has_qb = a.q_bias != nullptr, has_kb = a.k_bias != nullptr, has_vb = a.v_bias != nullptr;
    float wq[NT], wk[NT], wv[NT], bq[NT], bk[NT], bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      wq[j] = (has_ws ? a.q_wscale + h * DH : pf_dummy)[tt[j]]; wk[j] = (has_ws ? a.k_wscale + g * DH : pf_dummy)[tt[j]];
      wv[j] = (has_ws ? a.v_wscale + g * DH : pf_dummy)[tt[j]];
      bq[j] = (has_qb ? a.q_bias + h * DH : pf_dummy)[tt[j]]; bk[j] = (has_kb ? a.k_bias + g * DH : pf_dummy)[tt[j]];
      bv[j] = (has_vb ? a.v_bias + g * DH : pf_dummy)[tt[j]];
    }
    // rows pipeline: the projection arrives as K-group slabs (summed in slab order: deterministic); four slabs' loads in flight at a time
    int sl = 1;
    for (; sl + 4 <= a.in_kgroups; sl += 4) {
      float tq[NT][4], tk[NT][4], tv[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          tq[j][u] = a.q[iq[j] + (sl + u) * a.in_kg_stride];
          tk[j][u] = a.new_k[ik[j] + (sl + u) * a.in_kg_stride];
          tv[j][u] = a.new_v[ik[j] + (sl + u) * a.in_kg_stride];
        }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) { q0[j] += tq[j][u]; k0[j] += tk[j][u]; v0[j] += tv[j][u]; }
    }
    for (; sl < a.in_kgroups; ++sl) {
      float tq[NT], tk[NT], tv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        tq[j] = a.q[iq[j] + sl * a.in_kg_stride]; tk[j] = a.new_k[ik[j] + sl * a.in_kg_stride]; tv[j] = a.new_v[ik[j] + sl * a.in_kg_stride];
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) { q0[j] += tq[j]; k0[j] += tk[j]; v0[j] += tv[j]; }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = tid + j * NW * 64;
      if (t >= DH) continue;
      float qv = q0[j], kv2 = k0[j], vv2 = v0[j];
      if (has_ws) { qv *= wq[j]; kv2 *= wk[j]; vv2 *= wv[j]; }
      if (has_qb) qv += bq[j];
      if (has_kb) kv2 += bk[j];
      if (has_vb) vv2 += bv[j];
      qs[t] = qv;
      ks_new[t] = kv2;
      vs_new[t] = kv_round<KVT>(vv2);   // as the cache will hold it (the reference attends over the cache)
    }
    __syncthreads();
    // q and k: the arithmetic of head_norm_rope_kernel (transformer.hip), same order.  Four-wave workgroups give q to wave 0 and k to wave 1;
    // the one-wave workgroup of short key ranges does both
    auto norm_rope = [&](float* vec, const bool is_q) {
      const bool has_nw = is_q ? pf_has_qnw : pf_has_knw;
      const float nw0 = is_q ? pf_qnw0 : pf_knw0, nw1 = is_q ? pf_qnw1 : pf_knw1;
      constexpr int half = DH / 2;
      const bool act = lane < half;
      const int i0 = a.rope_mode == 1 ? 2 * lane : lane, i1 = a.rope_mode == 1 ? 2 * lane + 1 : lane + half;
      float x0 = 0.f, x1 = 0.f;
      if (act) { x0 = vec[i0]; x1 = vec[i1]; }
      if (has_nw) {
        const float ss = wave_sum_fast(sumsq2(x0, x1));   // (the reduction of head_norm_rope_kernel: the two are held bit-equal)
        const float r = rsqrtf(ss / (float)DH + a.norm_eps);
        if (act) { x0 = x0 * r * nw0; x1 = x1 * r * nw1; }
      }
      if (a.rope_cos && act) {
        const float c = pf_cos, sn = pf_sin;
        float y0, y1;
        rope_pair(x0, x1, c, sn, y0, y1);
        x0 = y0; x1 = y1;
      }
      wave_lds_sync2();
      if (act) {
        if (is_q) { vec[i0] = x0 * qsc; vec[i1] = x1 * qsc; }
        else { vec[i0] = kv_round<KVT>(x0); vec[i1] = kv_round<KVT>(x1); }
      }
    };
    if constexpr (NW == 1) {
      norm_rope(qs, true);
      norm_rope(ks_new, false);
    } else {
      if (wave == 0) norm_rope(qs, true);
      else if (wave == 1) norm_rope(ks_new, false);
    }
  }
  __syncthreads();
  int kend = len_k, kbeg = 0;
  if (a.causal) kend = qpos + 1 < len_k ? qpos + 1 : len_k;
  if (fused) kend = len_k - 1;   // the cached keys; the new one (position len_k - 1 = the query's own) is added from ks_new / vs_new below
  if (a.window > 0) { kbeg = qpos - a.window + 1; if (kbeg < 0) kbeg = 0; }
  if (a.k_start && a.k_start[b] > kbeg) kbeg = a.k_start[b];
  if (nsplit > 1) {  // this workgroup's share: contiguous, a multiple of 64 keys
    const int span = kend > kbeg ? kend - kbeg : 0;
    const int per = (((span + nsplit - 1) / nsplit) + 63) & ~63;
    kbeg += sp * per;
    if (kbeg + per < kend) kend = kbeg + per;
  }
  // heads packed inside a row (g * DH) or head-major planes (k_hstride: a head's keys contiguous -- the layout for long key ranges:
  // with rows of 2 * heads * DH floats every key of one head sits 6-8 KB from the next and lands on the same one or two L2 channels)
  kvp kbase = (kvp)a.k + (int64_t)b * a.k_bstride + (a.k_hstride ? (int64_t)g * a.k_hstride : (int64_t)g * DH);
  kvp vbase = (kvp)a.v + (int64_t)b * a.v_bstride + (a.v_hstride ? (int64_t)g * a.v_hstride : (int64_t)g * DH);
  float m = -INFINITY, l = 0.f, o[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) o[i] = 0.f;
  for (int kb = kbeg + wave * 64; kb < kend; kb += NW * 64) {
    // q.k with FOUR lanes per key: each load instruction then touches 16 keys x 64 contiguous bytes (16 cache lines) instead of 64 keys x
    // 16 bytes (64 lines) -- the row-per-lane pattern is bound by the address unit (one line per clock), not by bandwidth.  Lane (g, sub) of a
    // 16-key pass owns floats [i*16 + sub*4, +4) of key g for i = 0 .. DH/16; two xor-shuffles finish the dot product; the 64 scores of the
    // chunk are redistributed one per lane through LDS.
    // fp32 caches: the chunk's first 16 value rows are requested together with its keys (their addresses depend on neither the scores nor the
    // softmax): a short key range -- a decode step of CSM's depth decoder attends over <= 32 positions -- then costs one round trip, not three
    constexpr int VB = (NW == 16 && DH == 128) ? 8 : 16;   // the 1024-thread instantiation has 128 registers per lane
    float vpre[VB][ND];
    if constexpr (KVT == 0) {
      const int nn = kend - kb < 64 ? kend - kb : 64;
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        const float* vrow = vbase + (int64_t)(kb + (u < nn ? u : nn - 1)) * a.ldv;
#pragma unroll
        for (int i = 0; i < ND; ++i) vpre[u][i] = vrow[i * 64 + lane];
      }
    }
    {
      // The four 16-key passes of a chunk go in groups of PG whose loads are ALL requested before the first is used (as one rolled sequence the
      // passes shared their registers and every pass waited for its own loads: four serial round trips per chunk); a group with no visible key
      // is skipped (wave-uniform) and scores -inf.
      constexpr int PG = DH == 64 ? 4 : (NW == 16 ? 1 : 2);   // (the 1024-thread instantiation has 128 registers per lane)
      const int sub = lane & 3, grp = lane >> 2;
#pragma unroll
      for (int pg = 0; pg < 4; pg += PG) {
        if (kb + pg * 16 >= kend) {
          if (sub == 0) {
#pragma unroll
            for (int u = 0; u < PG; ++u) ps[wave][(pg + u) * 16 + grp] = -INFINITY;
          }
          continue;
        }
        float4 kv[PG][DH / 16];
        bool valid[PG];
#pragma unroll
        for (int u = 0; u < PG; ++u) {
          const int key = kb + (pg + u) * 16 + grp;
          valid[u] = key < kend;
          kvp krow = kbase + (int64_t)(valid[u] ? key : kend - 1) * a.ldk + sub * 4;
#pragma unroll
          for (int i = 0; i < DH / 16; ++i) kv[u][i] = kv_load4<KVT>(krow + i * 16);
        }
#pragma unroll
        for (int u = 0; u < PG; ++u) {
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int i = 0; i < DH / 16; ++i) {
            const int d = i * 16 + sub * 4;
            t0 = fmaf(qs[d], kv[u][i].x, t0);
            t1 = fmaf(qs[d + 1], kv[u][i].y, t1);
            t0 = fmaf(qs[d + 2], kv[u][i].z, t0);
            t1 = fmaf(qs[d + 3], kv[u][i].w, t1);
          }
          float t = t0 + t1;
          t += __shfl_xor(t, 1, 64);
          t += __shfl_xor(t, 2, 64);
          if (sub == 0) ps[wave][(pg + u) * 16 + grp] = valid[u] ? t : -INFINITY;
        }
      }
    }
    wave_lds_sync2();
    const float s = ps[wave][lane];
    wave_lds_sync2();  // every lane has its score before ps is overwritten with the probabilities
    const float m_new = fmaxf(m, wave_max_fast(s));  // finite: key kb itself is visible (DPP reductions: every lane of the wave is here)
    const float alpha = exp2f(m - m_new);
    const float p = exp2f(s - m_new);
    l = l * alpha + wave_sum_fast(p);
    m = m_new;
    ps[wave][lane] = p;
    wave_lds_sync2();
    const int n = kend - kb < 64 ? kend - kb : 64;
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] *= alpha;
    // p.V: 8 value rows in flight per step (a rolled loop keeps ONE load in flight and serialises 64 L2 latencies per chunk: that, not
    // bandwidth, was what made the 1500-key cross-attention step take 42 us)
    if constexpr (KVT == 0) {
#pragma unroll
      for (int u = 0; u < VB; ++u) {   // the rows requested with the keys
        const float pj = u < n ? ps[wave][u] : 0.f;
#pragma unroll
        for (int i = 0; i < ND; ++i) o[i] = fmaf(pj, vpre[u][i], o[i]);
      }
      for (int jj = VB; jj < n; jj += VB) {
        float vv[VB][ND];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
          const int j = jj + u < n ? jj + u : n - 1;
          const float* vrow = vbase + (int64_t)(kb + j) * a.ldv;
#pragma unroll
          for (int i = 0; i < ND; ++i) vv[u][i] = vrow[i * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
          const float pj = jj + u < n ? ps[wave][jj + u] : 0.f;
#pragma unroll
          for (int i = 0; i < ND; ++i) o[i] = fmaf(pj, vv[u][i], o[i]);
        }
      }
    } else {
      // 16-bit values: a lane loads one 4-byte PAIR of channels, so a wave load covers KPL = 128 / DH whole rows (two 128-byte rows at DH = 64,
      // one 256-byte row at DH = 128).  Lane (r = lane / PR, c = lane % PR) accumulates channels 2c, 2c + 1 over the keys j = r (mod KPL);
      // the KPL partial sums are folded once per chunk and handed back to the channel-per-lane layout of `o` through LDS.
      constexpr int PR = DH / 2, KPL = 64 / PR;
      const int r = lane / PR, c = lane - r * PR;
      float o2[2] = {0.f, 0.f};
      for (int jj = 0; jj < n; jj += 8 * KPL) {
        uint32_t vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          int j = jj + u * KPL + r;
          j = j < n ? j : n - 1;
          vv[u] = *(const uint32_t*)(vbase + (int64_t)(kb + j) * a.ldv + 2 * c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = jj + u * KPL + r;
          const float pj = j < n ? ps[wave][j] : 0.f;
          o2[0] = fmaf(pj, kv_f32<KVT>((uint16_t)(vv[u] & 0xffffu)), o2[0]);
          o2[1] = fmaf(pj, kv_f32<KVT>((uint16_t)(vv[u] >> 16)), o2[1]);
        }
      }
      if constexpr (KPL == 2) {
        o2[0] += __shfl_xor(o2[0], 32, 64);
        o2[1] += __shfl_xor(o2[1], 32, 64);
      }
      // the probabilities in ps[wave] have been consumed: reuse the row to go from pair-per-lane back to channel-per-lane (64 channels a round)
#pragma unroll
      for (int h2 = 0; h2 < ND; ++h2) {
        wave_lds_sync2();
        if (r == 0 && c >= 32 * h2 && c < 32 * h2 + 32) { ps[wave][2 * (c - 32 * h2)] = o2[0]; ps[wave][2 * (c - 32 * h2) + 1] = o2[1]; }
        wave_lds_sync2();
        o[h2] += ps[wave][lane];
      }
    }
    wave_lds_sync2();
  }
  if (fused && wave == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) t = fmaf(qs[i * 64 + lane], ks_new[i * 64 + lane], t);
    const float s_new = wave_sum_fast(t);
    const float m_new = fmaxf(m, s_new);
    const float alpha = exp2f(m - m_new);   // m = -inf (no cached key visible) -> 0
    const float p = exp2f(s_new - m_new);
    l = l * alpha + p;
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = o[i] * alpha + p * vs_new[i * 64 + lane];
    m = m_new;
  }
  if (fused && wave == (NW > 1 ? 1 : 0) && h % (a.heads / a.kv_heads) == 0) {   // one writer per (item, kv head): the processed k and the raw v go into the cache
    using kvw = typename kv_t<KVT>::type;
    kvw* kdst = (kvw*)a.k + (int64_t)b * a.k_bstride + (a.k_hstride ? (int64_t)g * a.k_hstride : (int64_t)g * DH) + (int64_t)(len_k - 1) * a.ldk;
    kvw* vdst = (kvw*)a.v + (int64_t)b * a.v_bstride + (a.v_hstride ? (int64_t)g * a.v_hstride : (int64_t)g * DH) + (int64_t)(len_k - 1) * a.ldv;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int d = i * 64 + lane;
      if constexpr (KVT == 0) { kdst[d] = ks_new[d]; vdst[d] = vs_new[d]; }
      else if constexpr (KVT == 1) { kdst[d] = (uint16_t)(pack_bf16x2(ks_new[d], 0.f) & 0xffffu); vdst[d] = (uint16_t)(pack_bf16x2(vs_new[d], 0.f) & 0xffffu); }
      else { kdst[d] = (uint16_t)(pack_f16x2(ks_new[d], 0.f) & 0xffffu); vdst[d] = (uint16_t)(pack_f16x2(vs_new[d], 0.f) & 0xffffu); }
    }
  }
  if (lane == 0) { red_m[wave] = m; red_l[wave] = l; }
#pragma unroll
  for (int i = 0; i < ND; ++i) red_o[wave][i * 64 + lane] = o[i];
  __syncthreads();
  if (wave == 0) {
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NW; ++i) M = fmaxf(M, red_m[i]);
    float L = 0.f;
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      w[i] = red_m[i] == -INFINITY ? 0.f : exp2f(red_m[i] - M);
      L += red_l[i] * w[i];
    }
    float* orow = a.out + (int64_t)b * a.out_bstride + (int64_t)qi * a.ldo + h * DH;
    float t[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int d = i * 64 + lane;
      t[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NW; ++j) t[i] += red_o[j][d] * w[j];
    }
    if (nsplit > 1) {
      // publish this workgroup's partial (M, L, O), then take a ticket: the workgroup that draws the last ticket merges all of them.
      // No workgroup ever waits for another one (no spinning), so there is nothing to deadlock on.
      const int64_t idx = ((int64_t)b * a.heads + h) * a.Tq + qi;
      float* rec = a.split_ws + (idx * nsplit + sp) * (DH + 2);
      if (lane == 0) { rec[0] = M; rec[1] = L; }
#pragma unroll
      for (int i = 0; i < ND; ++i) rec[2 + i * 64 + lane] = t[i];
      __threadfence();
      int ticket = 0;
      if (lane == 0) ticket = atomicAdd(a.split_cnt + idx, 1);
      ticket = __shfl(ticket, 0, 64);
      if (ticket != nsplit - 1) return;
      __threadfence();
      const float* recs = a.split_ws + idx * nsplit * (DH + 2);
      M = -INFINITY;
      for (int s2 = 0; s2 < nsplit; ++s2) M = fmaxf(M, __hip_atomic_load(recs + s2 * (DH + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      L = 0.f;
#pragma unroll
      for (int i = 0; i < ND; ++i) t[i] = 0.f;
      for (int s2 = 0; s2 < nsplit; ++s2) {
        const float* r2 = recs + s2 * (DH + 2);
        const float ms = __hip_atomic_load(r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float ws2 = ms == -INFINITY ? 0.f : exp2f(ms - M);
        L += __hip_atomic_load(r2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * ws2;
#pragma unroll
        for (int i = 0; i < ND; ++i) t[i] += __hip_atomic_load(r2 + 2 + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * ws2;
      }
      if (lane == 0) a.split_cnt[idx] = 0;  // leave the counters zeroed for the next launch
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    if (a.out) {
#pragma unroll
      for (int i = 0; i < ND; ++i) orow[i * 64 + lane] = t[i] * inv;
    }
    if (a.out_planes) {   // the row also leaves as hi + lo planes (fragment order of mi355_rows_gemm): lane c < DH / 8 builds the piece of channels 8 c .. 8 c + 7
#pragma unroll
      for (int i = 0; i < ND; ++i) red_o[0][i * 64 + lane] = t[i] * inv;
      wave_lds_sync2();
      if (lane < DH / 8) {
        float e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = red_o[0][8 * lane + u];
        uint4 hi, lo;
        if (a.planes_dtype == MI355_W_F16) {
          auto sp2 = [](float x, float y, uint32_t& hh, uint32_t& ll) {
            hh = pack_f16x2(x, y);
            const float hx = (float)__builtin_bit_cast(_Float16, (uint16_t)(hh & 0xffffu)), hy = (float)__builtin_bit_cast(_Float16, (uint16_t)(hh >> 16));
            ll = pack_f16x2(x - hx, y - hy);
          };
          sp2(e[0], e[1], hi.x, lo.x); sp2(e[2], e[3], hi.y, lo.y); sp2(e[4], e[5], hi.z, lo.z); sp2(e[6], e[7], hi.w, lo.w);
        } else {
          auto sp2 = [](float x, float y, uint32_t& hh, uint32_t& ll) {
            hh = pack_bf16x2(x, y);
            const float hx = __builtin_bit_cast(float, hh << 16), hy = __builtin_bit_cast(float, hh & 0xffff0000u);
            ll = pack_bf16x2(x - hx, y - hy);
          };
          sp2(e[0], e[1], hi.x, lo.x); sp2(e[2], e[3], hi.y, lo.y); sp2(e[4], e[5], hi.z, lo.z); sp2(e[6], e[7], hi.w, lo.w);
        }
        const int k = h * DH + 8 * lane;   // column of the attention output row
        const int s = k >> 6, gq = (k & 63) >> 4, hh = (k >> 3) & 1;
        uint4* const pl = (uint4*)a.out_planes;
        pl[(((s * 2 + 0) * 2 + hh) * 4 + gq) * a.planes_R + b] = hi;
        pl[(((s * 2 + 1) * 2 + hh) * 4 + gq) * a.planes_R + b] = lo;
      }
    }
  }
}

}  // namespace

extern "C" int mi355_flash_attention(const mi355_flash_attn_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->q && ap->k && ap->v && (ap->out || ap->out_planes), "flash_attention: null tensor");
  mi355_flash_attn_args a = *ap;
  MI355_REQUIRE(a.dh == 64 || a.dh == 128, "flash_attention: head dim must be 64 or 128 (got %d)", a.dh);
  MI355_REQUIRE(a.heads > 0 && a.kv_heads > 0 && a.heads % a.kv_heads == 0, "flash_attention: heads must be a multiple of kv_heads");
  MI355_REQUIRE(a.B > 0 && a.Tq > 0 && a.Tk > 0, "flash_attention: bad shape");
  MI355_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && a.q_bstride % 4 == 0 && a.k_bstride % 4 == 0 &&
                    a.v_bstride % 4 == 0 && a.out_bstride % 4 == 0,
                "flash_attention: strides must be multiples of 4 floats");
  MI355_REQUIRE(((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.out | (uintptr_t)a.out_planes) % 16 == 0, "flash_attention: tensors must be 16-byte aligned");
  MI355_REQUIRE(a.in_kgroups <= 1 || (a.new_k && a.in_kg_stride > 0), "flash_attention: slab inputs exist for the fused decode step only");
  MI355_REQUIRE(!a.out_planes || (a.Tq == 1 && a.mode != 1 && a.nsplit <= 1 && (a.planes_R == 16 || a.planes_R == 32 || a.planes_R == 64) && a.B <= a.planes_R &&
                                  (a.planes_dtype == MI355_W_BF16 || a.planes_dtype == MI355_W_F16)),
                "flash_attention: a planes output needs a single-query decode step of at most planes_R (16 / 32 / 64) items");
  MI355_REQUIRE(a.out || a.out_planes, "flash_attention: no destination");
  MI355_REQUIRE(a.window >= 0, "flash_attention: window must be >= 0");
  MI355_REQUIRE(a.kv_dtype >= MI355_KV_F32 && a.kv_dtype <= MI355_KV_F16, "flash_attention: kv_dtype must be MI355_KV_F32, MI355_KV_BF16 or MI355_KV_F16");
  const int kvt = a.kv_dtype;
  MI355_REQUIRE(kvt == 0 || (((uintptr_t)a.k | (uintptr_t)a.v) % 8 == 0), "flash_attention: 16-bit K / V must be 8-byte aligned");
  MI355_REQUIRE(!a.new_k || (a.new_v && a.Tq == 1 && !a.lens_q && a.mode != 1 && a.nsplit <= 1 && a.causal),
                "flash_attention: the fused norm / rope step is a causal single-query decode step without ragged query lengths or key split");
  MI355_REQUIRE(!a.new_k || !a.rope_cos || (a.rope_sin && a.rope_rows > 0), "flash_attention: rope tables incomplete");
  hipStream_t st = (hipStream_t)stream;
  MI355_CLEAR_ERROR();
  if (a.new_k) { a.nsplit = 1; a.split_ws = nullptr; }
  const bool decode = a.mode == 2 || (a.mode == 0 && a.Tq <= 8);
  if (decode) {
    // long key ranges with few (query, head, item) triples leave most CUs idle and each workgroup pulls ~20 GB/s: split the keys
    // (flash-decoding) when the caller provided the partial-result workspace
    int nsplit = 1;
    if (a.split_ws && a.split_cnt && a.nsplit != 1) {
      const int blocks = a.Tq * a.heads * a.B;
      // Measured on Whisper's 1500-key cross-attention (96 workgroups): 38.0 us unsplit vs 53.6 us with 3 splits -- the agent-scope fences
      // around the ticket write back the L2, which costs more than the extra CUs bring.  So the split is opt-in (nsplit >= 2), never automatic.
      (void)blocks;
      if (a.nsplit > 1) nsplit = a.nsplit;
      if (nsplit > 8) nsplit = 8;
      if (nsplit < 1) nsplit = 1;
    }
    a.nsplit = nsplit;
    dim3 grid(a.Tq * nsplit, a.heads, a.B);
    // long key ranges (Whisper cross-attention: 1500 keys) get 16 waves per (query, head): the per-wave key loop is a dependent
    // chain of global loads, so more waves in flight is what shortens it; short ranges keep 4 waves
    const bool wide = a.Tk > 256;
    // ... and ranges of at most one 64-key chunk (the code predictor / depth decoder steps: <= 32 positions) ONE wave: of four waves three would
    // only hold registers and meet barriers, and 64 items x 16 heads of them do not fit the chip in one round
    static const bool one_off = getenv("MI355_ATTN_ONE_WAVE") != nullptr && getenv("MI355_ATTN_ONE_WAVE")[0] == '0';   // A/B knob
    const bool one = a.Tk <= 64 && nsplit == 1 && !one_off;
#define MI355_DECODE_CASE(KVT)                                                                                   \
    if (a.dh == 64) {                                                                                            \
      if (wide) hipLaunchKernelGGL((attn_decode_kernel<64, 16, KVT>), grid, dim3(1024), 0, st, a);               \
      else if (one) hipLaunchKernelGGL((attn_decode_kernel<64, 1, KVT>), grid, dim3(64), 0, st, a);              \
      else hipLaunchKernelGGL((attn_decode_kernel<64, 4, KVT>), grid, dim3(256), 0, st, a);                      \
    } else {                                                                                                     \
      if (wide) hipLaunchKernelGGL((attn_decode_kernel<128, 16, KVT>), grid, dim3(1024), 0, st, a);              \
      else if (one) hipLaunchKernelGGL((attn_decode_kernel<128, 1, KVT>), grid, dim3(64), 0, st, a);             \
      else hipLaunchKernelGGL((attn_decode_kernel<128, 4, KVT>), grid, dim3(256), 0, st, a);                     \
    }
    if (kvt == 0) { MI355_DECODE_CASE(0) } else if (kvt == 1) { MI355_DECODE_CASE(1) } else { MI355_DECODE_CASE(2) }
#undef MI355_DECODE_CASE
  } else {
    dim3 grid((a.Tq + 127) / 128, a.heads, a.B);
    if (kvt == 0) {
      if (a.dh == 64) hipLaunchKernelGGL((flash_attn_kernel<64, 0>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((flash_attn_kernel<128, 0>), grid, dim3(256), 0, st, a);
    } else {
      MI355_REQUIRE(a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.k_bstride % 8 == 0 && a.v_bstride % 8 == 0 && a.k_hstride % 8 == 0 && a.v_hstride % 8 == 0 &&
                        (((uintptr_t)a.k | (uintptr_t)a.v) % 16 == 0),
                    "flash_attention: 16-bit K / V rows must be 16-byte aligned for the prefill kernel");
#define MI355_FLASH16_CASE(KVT)                                                                    \
      if (a.dh == 64) hipLaunchKernelGGL((flash_attn16_kernel<64, KVT>), grid, dim3(256), 0, st, a); \
      else hipLaunchKernelGGL((flash_attn16_kernel<128, KVT>), grid, dim3(256), 0, st, a);
      if (kvt == 1) { MI355_FLASH16_CASE(1) } else { MI355_FLASH16_CASE(2) }
#undef MI355_FLASH16_CASE
    }
#undef MI355_FLASH_CASE
  }
  MI355_LAUNCH_CHECK("flash_attention");
  return MI355_OK;
}
