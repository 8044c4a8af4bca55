// One-launch decode step of a transformer stack: a persistent "phase program" kernel (gfx950).
//
// mi355_stack_decode_step (stack_step.cpp) replaced the reference's per-op Python schedule (tts/models/qwen3_tts/talker.py:385-500,
// lm/models/llama.py:160-198, stt/models/whisper/whisper.py:405-416, 476-498) by 6-10 kernel launches per layer.  At 1-8 rows per step each of
// those launches is a few microseconds of work behind ~12 us of launch + ramp + dependent-latency chain, so a generated frame (100-800 launches)
// ran at 6-16 % of the HBM rate its weight stream allows.  Here the whole step is ONE launch: one workgroup per CU stays resident and walks a
// list of phases (GEMV with fused norm prologue / SwiGLU / residual epilogues, per-head norm + RoPE, KV-streaming attention, final norm),
// separated by grid barriers.  Design points, all from /opt/skills/guides (MI355X_MICROARCH "inter-workgroup visibility"):
//   * per-XCD L2s are not coherent and a CU's L1 is never refreshed: every activation that is written and read inside the launch moves with
//     system-scope (sc0 sc1) stores and loads on BOTH sides -- write-through, L1-bypassing -- so no wbl2 / inv fences are needed (3.5 us each);
//     weights, tables and the older KV rows are read-only inside the launch and use plain cached loads;
//   * the barrier is one relaxed agent-scope atomic add per workgroup + one polling lane (with s_sleep), after every wave drained its stores
//     (s_waitcnt vmcnt(0)); the counter is monotonic across launches (wrap-safe compare), so there is no memset node per step;
//   * the spin is BOUNDED: a workgroup that waits too long raises an error flag and leaves, the launch always terminates;
//   * the weight stream of a phase does not depend on the previous phase's results: each wave issues its first weight slices BEFORE it
//     stages x, so HBM latency overlaps the barrier hop and the staging;
//   * the grid is exactly the number of co-resident workgroups (occupancy query x CU count), a precondition of any hand-rolled grid barrier.
// The phase list is built on the host from the same mi355_stack_desc as the multi-launch runner, uploaded once per (stack, buffers, B) and
// cached; the step index enters as a kernel argument (KV slot = base + offset * row, keys = offset + 1, RoPE position = offset).
//
// STATUS (round 1, GPU call 21, profiles/r1_decode_runner_ab_call21.txt): parity-green against the multi-launch runner and the oracle, but 1.7-2.0x
// SLOWER (CSM-1B 14.8 vs 7.4 ms per frame, Qwen3-TTS-1.7B 17.9 vs 10.8, Whisper-small 3.17 vs 1.68 ms per token step): ~17 us per phase (counter
// barrier at 256 workgroups + 4 waves per CU at 384 registers + scalar write-through stores) against ~7.6 us per launch.  It is therefore OPT-IN
// (MI355_STEP_FUSED=1 / mi355_stack_fused_set(1)) and kept as the A/B harness for the next attempt (<= 128 registers for 2 workgroups per CU,
// the XCD-hierarchical barrier, 16-byte write-through stores).
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace {

enum { PH_GEMV = 0, PH_ROPE = 1, PH_ATTN = 2, PH_NORM = 3 };

struct Phase {
  int32_t kind;
  // ---- gemv: y[m, n] = epilogue(sum_k norm(x)[m, k] W[n, k])
  int32_t x_id; int32_t ldx; int32_t M; int32_t K;   // activation operands are BUFFER IDS (Bufs below): the pointers change per call
  const uint16_t* w; int32_t N; int32_t wt;          // wt: MI355_W_BF16 / MI355_W_F16 / MI355_W_FP8 (bytes + wscale)
  const float* wscale;
  const float* bias; int32_t act; const float* colscale;
  int32_t res_id; int32_t ldr; int32_t glu;          // res_id < 0: no residual
  int32_t y_id; int32_t ldy;
  int32_t norm; const float* nw; const float* nb; float eps;
  float* y2; int32_t ldy2; int32_t split; int64_t y2_step;  // columns >= split go to y2 + offset * y2_step (the KV-cache slot)
  // ---- rope: q = buffer x_id [B, heads * dh] in place; k rows at k2 + offset * k2_step + b * k2_bstride, in place
  int32_t heads; int32_t heads2; int32_t dh; int32_t rope_mode; int32_t B;
  const float* qnw; const float* knw; const float* cos_t; const float* sin_t;
  float* k2; int64_t k2_bstride; int64_t k2_step;
  // ---- attn: one query per (item, head) over Tk_base (+ offset) keys
  int32_t q_id; int32_t ldq; const float* kc; const float* vc; int64_t kv_bstride; int32_t ldkv; int64_t hstride;
  int32_t kv_heads; int32_t Tk_base; int32_t tk_add_offset; int32_t window; float scale; int32_t out_id; int32_t ldo;
};

enum { BUF_X = 0, BUF_Q = 1, BUF_ATT = 2, BUF_MID = 3, BUF_OUT = 4 };
struct Bufs { float* p[5]; };  // x [B, d_model] (residual stream), q / att [B, heads dh], mid [B, d_ff], out [B, d_model] (nullable)

constexpr int kXsCap = 16384;          // floats of LDS for the staged input rows (64 KB)
constexpr int kStageSlots = 32;        // kXsCap / 512: 8-byte loads per thread when the chunk is full
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr float kLog2eM = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------- coherent activation accessors
typedef __attribute__((address_space(1))) uint32_t gu32;   // global address space: keeps these accesses global_* instead of flat_*
typedef __attribute__((address_space(1))) uint64_t gu64;
__device__ __forceinline__ float ld_act(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load((gu32*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ float2 ld_act2(const float* p) {
  const uint64_t u = __hip_atomic_load((gu64*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_float2(__builtin_bit_cast(float, (uint32_t)u), __builtin_bit_cast(float, (uint32_t)(u >> 32)));
}
__device__ __forceinline__ void st_act(float* p, float v) {
  __hip_atomic_store((gu32*)(uintptr_t)p, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Grid barrier: counter value `target` = every workgroup has arrived at this phase boundary.  Returns false when the wait was abandoned.
__device__ __forceinline__ bool grid_barrier(uint32_t* cnt, const uint32_t target, int32_t* err) {
  __shared__ int ok_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    int ok = 1;
    while ((int32_t)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) { ok = 0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

__device__ __forceinline__ float act_m(float v, int act) {
  switch (act) {
    case MI355_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case MI355_ACT_SILU: return v / (1.0f + expf(-v));
    case MI355_ACT_GELU_TANH: return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    case MI355_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case MI355_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// ---------------------------------------------------------------------------------------------- GEMV phase
// Column pairs are dealt round-robin to the 4 * gridDim.x waves of the grid; the input rows are staged (normalised on the way) into LDS once
// per workgroup when they fit one chunk (the common case: M * K <= 16384), else chunk by chunk per column-pair iteration.
template <int MT, int WT>
__device__ void phase_gemv(const Phase& p, const Bufs& bf, float* xs, float* st, const int offset) {
  constexpr int NC = 2, D = 4;
  constexpr int EPL = mi355_wt<WT>::EPL, ESZ = 16 / EPL, SL = 64 * EPL;  // elements per lane piece / bytes per element / elements per k-slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  const int K = p.K, N = p.N, M = p.M;
  const int ngroups = (N + NC - 1) / NC;
  const int iters = (ngroups + W - 1) / W;
  const int Kp = (K + SL - 1) / SL * SL;
  int KC = (kXsCap / MT) & ~(SL - 1);
  if (KC > Kp) KC = Kp;
  const int nchunks = (K + KC - 1) / KC;
  const int IPC = KC / SL;
  const int n_it = Kp / SL;
  float* y2 = p.y2 ? p.y2 + (int64_t)offset * p.y2_step : nullptr;
  const float* px = bf.p[p.x_id];
  const float* pres = p.res_id >= 0 ? bf.p[p.res_id] : nullptr;
  float* py = bf.p[p.y_id];

  auto stage = [&](const int c) {
    const int k0 = c * KC;
    const int kc = K - k0 < KC ? K - k0 : KC;
    const int q = ((kc + 511) & ~511) >> 9;  // 512-float slices per row in this chunk (scalar)
    __syncthreads();                         // the previous readers of xs are done
    float2 v[kStageSlots];
#pragma unroll
    for (int i = 0; i < kStageSlots; ++i) {  // every load of this thread goes out before the first store: ONE memory round trip
      const int m = i / q, kk = tid * 2 + (i - m * q) * 512;
      v[i] = make_float2(0.f, 0.f);
      if (m < M && m < MT && kk < kc) v[i] = ld_act2(px + (int64_t)m * p.ldx + k0 + kk);
    }
#pragma unroll
    for (int i = 0; i < kStageSlots; ++i) {
      const int m = i / q, kk = tid * 2 + (i - m * q) * 512;
      if (m < MT && kk < KC) *(float2*)(xs + m * KC + kk) = v[i];  // zero fill past kc and past M
    }
    if (p.norm) {  // only reached with nchunks == 1 (host-side eligibility): the whole row is in LDS
      __syncthreads();
      for (int m = wave; m < M; m += 4) {
        const float* xr = xs + m * KC;
        float s = 0.f;
        for (int k = lane * 4; k < K; k += 256) { const float4 t = *(const float4*)(xr + k); s += (t.x + t.y) + (t.z + t.w); }
        const float mean = p.norm == 1 ? wave_sum(s) / (float)K : 0.f;
        float qq = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
          const float4 t = *(const float4*)(xr + k);
          const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
          qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float var = wave_sum(qq) / (float)K;
        if (lane == 0) { st[m] = mean; st[8 + m] = p.norm == 1 ? 1.0f / sqrtf(var + p.eps) : rsqrtf(var + p.eps); }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kStageSlots; ++i) {
        const int m = i / q, kk = tid * 2 + (i - m * q) * 512;
        if (m < M && m < MT && kk < kc) {
          const float mu = st[m], rs = st[8 + m];
          float2 w2 = make_float2(1.f, 1.f), b2 = make_float2(0.f, 0.f);
          if (p.nw) w2 = *(const float2*)(p.nw + kk);
          if (p.nb) b2 = *(const float2*)(p.nb + kk);
          float2* d = (float2*)(xs + m * KC + kk);
          const float2 t = *d;
          *d = make_float2((t.x - mu) * rs * w2.x + b2.x, (t.y - mu) * rs * w2.y + b2.y);
        }
      }
    }
    __syncthreads();
  };

  for (int gi = 0; gi < iters; ++gi) {
    const int g = gw + gi * W;
    const bool valid = g < ngroups;  // wave-uniform
    const int n0 = g * NC;
    float acc[NC][MT];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
    const uint8_t* wrow[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = n0 + c < N ? n0 + c : N - 1;
      wrow[c] = (const uint8_t*)p.w + (int64_t)(valid ? n : 0) * K * ESZ;
    }
    uint4 ring[D][NC];
    auto issue = [&](int it, uint4 (&dst)[NC]) {
      const int k = it * SL + lane * EPL;
#pragma unroll
      for (int c = 0; c < NC; ++c) dst[c] = (valid && k < K) ? *(const uint4*)(wrow[c] + (int64_t)k * ESZ) : make_uint4(0u, 0u, 0u, 0u);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < n_it) issue(d, ring[d]);
    for (int base = 0; base < n_it; base += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int it = base + d;
        if (it >= n_it) break;
        if (it % IPC == 0 && (gi == 0 || nchunks > 1)) stage(it / IPC);  // block-uniform
        const int kl = (it % IPC) * SL + lane * EPL;
        if (it * SL + lane * EPL < K) {
          float wf[NC][EPL];
#pragma unroll
          for (int c = 0; c < NC; ++c) cvt_w16<WT>(ring[d][c], wf[c]);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float xv[EPL];
#pragma unroll
            for (int j4 = 0; j4 < EPL / 4; ++j4) {
              const float4 t = *(const float4*)(xs + m * KC + kl + 4 * j4);
              xv[4 * j4] = t.x; xv[4 * j4 + 1] = t.y; xv[4 * j4 + 2] = t.z; xv[4 * j4 + 3] = t.w;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int j = 0; j < EPL; ++j) acc[c][m] = fmaf(xv[j], wf[c][j], acc[c][m]);
          }
        }
        if (it + D < n_it) issue(it + D, ring[d]);
      }
    }
    // wave-level reduction + epilogue
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[c][m] = wave_sum(acc[c][m]);
    if (!valid || lane != 0) continue;
    if (p.glu) {
      if (n0 + 1 < N) {
        const float bg = p.bias ? p.bias[n0] : 0.f, bu = p.bias ? p.bias[n0 + 1] : 0.f;
        const float wg = p.wscale ? p.wscale[n0] * kFp8Unbias : 1.f, wu = p.wscale ? p.wscale[n0 + 1] * kFp8Unbias : 1.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (m >= M) break;
          const float gg = acc[0][m] * wg + bg, u = acc[1][m] * wu + bu;
          st_act(py + (int64_t)m * p.ldy + (n0 >> 1), (gg / (1.0f + expf(-gg))) * u);
        }
      }
      continue;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = n0 + c;
      if (n >= N) break;
      const float bias = p.bias ? p.bias[n] : 0.f;
      const float cs = p.colscale ? p.colscale[n] : 1.f;
      const float ws = p.wscale ? p.wscale[n] * kFp8Unbias : 1.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= M) break;
        float v = act_m(acc[c][m] * ws + bias, p.act) * cs;
        if (pres) v += ld_act(pres + (int64_t)m * p.ldr + n);
        if (y2 && n >= p.split) st_act(y2 + (int64_t)m * p.ldy2 + (n - p.split), v);
        else st_act(py + (int64_t)m * p.ldy + n, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- per-head RMSNorm + RoPE phase (q and the new k row)
__device__ void phase_rope(const Phase& p, const Bufs& bf, const int offset) {
  const int lane = threadIdx.x & 63;
  const int W = gridDim.x * 4;
  const int ht = p.heads + p.heads2;
  const int total = p.B * ht;
  const int half = p.dh >> 1;
  for (int wid = blockIdx.x * 4 + (threadIdx.x >> 6); wid < total; wid += W) {
    int h = wid % ht;
    const int b = wid / ht;
    const bool second = h >= p.heads;
    if (second) h -= p.heads;
    float* xr = second ? p.k2 + (int64_t)offset * p.k2_step + (int64_t)b * p.k2_bstride + h * p.dh : bf.p[p.x_id] + (int64_t)b * p.ldx + h * p.dh;
    const float* nw = second ? p.knw : p.qnw;
    const bool act = lane < half;
    int i0, i1;
    if (p.rope_mode == 1) { i0 = 2 * lane; i1 = 2 * lane + 1; }
    else { i0 = lane; i1 = lane + half; }
    float x0 = 0.f, x1 = 0.f;
    if (act) { x0 = ld_act(xr + i0); x1 = ld_act(xr + i1); }
    if (nw) {
      const float ss = wave_sum(x0 * x0 + x1 * x1);
      const float r = rsqrtf(ss / (float)p.dh + p.eps);
      if (act) { x0 = x0 * r * nw[i0]; x1 = x1 * r * nw[i1]; }
    }
    if (p.cos_t && act) {
      const float c = p.cos_t[(int64_t)offset * half + lane], s = p.sin_t[(int64_t)offset * half + lane];
      const float y0 = x0 * c - x1 * s;
      const float y1 = x1 * c + x0 * s;
      x0 = y0; x1 = y1;
    }
    if (act) { st_act(xr + i0, x0); st_act(xr + i1, x1); }
  }
}

// ---------------------------------------------------------------------------------------------- attention phase (one query per item and head)
// Same algorithm as attn_decode_kernel<DH, 4> (flash_attn.hip): four waves take interleaved 64-key chunks, four lanes per key for q.k, eight
// value rows in flight for p.V, online softmax per wave, merged through LDS.  q and the output are coherent accesses; K / V rows are plain
// loads: rows older than this step were written by earlier launches, and the newest row was written through to memory in this launch and has
// never been cached by any reader.
template <int DH>
__device__ void phase_attn(const Phase& p, const Bufs& bf, const int offset) {
  constexpr int NW = 4, ND = DH / 64;
  __shared__ float qs[DH];
  __shared__ float ps[NW][64];
  __shared__ float red_m[NW], red_l[NW];
  __shared__ float red_o[NW][DH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int units = p.B * p.heads;
  const int Tk = p.Tk_base + (p.tk_add_offset ? offset : 0);
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int b = u / p.heads, h = u - b * p.heads;
    const int g = h / (p.heads / p.kv_heads);
    __syncthreads();  // the previous unit's readers of the shared arrays are done
    for (int t = tid; t < DH; t += NW * 64) qs[t] = ld_act(bf.p[p.q_id] + (int64_t)b * p.ldq + h * DH + t) * (p.scale * kLog2eM);
    __syncthreads();
    const int kend = Tk;
    int kbeg = 0;
    if (p.window > 0) { kbeg = Tk - p.window; if (kbeg < 0) kbeg = 0; }
    const float* kbase = p.kc + (int64_t)b * p.kv_bstride + (p.hstride ? (int64_t)g * p.hstride : (int64_t)g * DH);
    const float* vbase = p.vc + (int64_t)b * p.kv_bstride + (p.hstride ? (int64_t)g * p.hstride : (int64_t)g * DH);
    float m = -INFINITY, l = 0.f, o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = 0.f;
    for (int kb = kbeg + wave * 64; kb < kend; kb += NW * 64) {
      {
        const int sub = lane & 3, grp = lane >> 2;
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
          const int key = kb + p4 * 16 + grp;
          const bool valid = key < kend;
          const float* krow = kbase + (int64_t)(valid ? key : kend - 1) * p.ldkv + sub * 4;
          float4 kv[DH / 16];
#pragma unroll
          for (int i = 0; i < DH / 16; ++i) kv[i] = *(const float4*)(krow + i * 16);
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int i = 0; i < DH / 16; ++i) {
            const int d = i * 16 + sub * 4;
            t0 = fmaf(qs[d], kv[i].x, t0);
            t1 = fmaf(qs[d + 1], kv[i].y, t1);
            t0 = fmaf(qs[d + 2], kv[i].z, t0);
            t1 = fmaf(qs[d + 3], kv[i].w, t1);
          }
          float t = t0 + t1;
          t += __shfl_xor(t, 1, 64);
          t += __shfl_xor(t, 2, 64);
          if (sub == 0) ps[wave][p4 * 16 + grp] = valid ? t : -INFINITY;
        }
      }
      wave_sync_lds();
      const float s = ps[wave][lane];
      wave_sync_lds();
      const float m_new = fmaxf(m, wave_max(s));
      const float alpha = exp2f(m - m_new);
      const float pr = exp2f(s - m_new);
      l = l * alpha + wave_sum(pr);
      m = m_new;
      ps[wave][lane] = pr;
      wave_sync_lds();
      const int n = kend - kb < 64 ? kend - kb : 64;
#pragma unroll
      for (int i = 0; i < ND; ++i) o[i] *= alpha;
      for (int jj = 0; jj < n; jj += 8) {
        float vv[8][ND];
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          const int j = jj + q8 < n ? jj + q8 : n - 1;
          const float* vrow = vbase + (int64_t)(kb + j) * p.ldkv;
#pragma unroll
          for (int i = 0; i < ND; ++i) vv[q8][i] = vrow[i * 64 + lane];
        }
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) {
          const float pj = jj + q8 < n ? ps[wave][jj + q8] : 0.f;
#pragma unroll
          for (int i = 0; i < ND; ++i) o[i] = fmaf(pj, vv[q8][i], o[i]);
        }
      }
      wave_sync_lds();
    }
    if (lane == 0) { red_m[wave] = m; red_l[wave] = l; }
#pragma unroll
    for (int i = 0; i < ND; ++i) red_o[wave][i * 64 + lane] = o[i];
    __syncthreads();
    if (wave == 0) {
      float Mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NW; ++i) Mx = fmaxf(Mx, red_m[i]);
      float L = 0.f;
      float w[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        w[i] = red_m[i] == -INFINITY ? 0.f : exp2f(red_m[i] - Mx);
        L += red_l[i] * w[i];
      }
      float* orow = bf.p[p.out_id] + (int64_t)b * p.ldo + h * DH;
      const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int d = i * 64 + lane;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NW; ++j) t += red_o[j][d] * w[j];
        st_act(orow + d, t * inv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- final norm phase: one workgroup per row
__device__ void phase_norm(const Phase& p, const Bufs& bf, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K;
  for (int m = blockIdx.x; m < p.M; m += gridDim.x) {
    const float* xr = bf.p[p.x_id] + (int64_t)m * p.ldx;
    __syncthreads();
    float s = 0.f;
    for (int k = tid; k < K; k += 256) s += ld_act(xr + k);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = p.norm == 1 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)K : 0.f;
    __syncthreads();
    float q = 0.f;
    for (int k = tid; k < K; k += 256) { const float d = ld_act(xr + k) - mean; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float var = ((red[0] + red[1]) + (red[2] + red[3])) / (float)K;
    const float rs = p.norm == 1 ? 1.0f / sqrtf(var + p.eps) : rsqrtf(var + p.eps);
    for (int k = tid; k < K; k += 256) {
      const float v = (ld_act(xr + k) - mean) * rs * (p.nw ? p.nw[k] : 1.f) + (p.nb ? p.nb[k] : 0.f);
      bf.p[p.y_id][(int64_t)m * p.ldy + k] = v;  // the launch's output: plain store, visible at kernel end
    }
  }
}

__global__ __launch_bounds__(256) void step_program_kernel(const Phase* __restrict__ prog, const int nphases, const int offset, uint32_t* cnt,
                                                           const uint32_t base, int32_t* err, const Bufs bf) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // kXsCap floats
  __shared__ float st[16];
  __shared__ Phase ph;
  __shared__ int dead;
  // an earlier launch of this program abandoned a barrier: its counter is inconsistent until mi355_stack_fused_check resets it -- leave at once
  // instead of timing out barrier after barrier (the flag is only ever raised, so every workgroup of this launch reads the same value)
  if (threadIdx.x == 0) dead = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (dead) return;
  for (int i = 0; i < nphases; ++i) {
    __syncthreads();
    {  // the phase record: read-only for the whole launch, plain loads
      const uint32_t* src = (const uint32_t*)(prog + i);
      uint32_t* dst = (uint32_t*)&ph;
      for (int t = threadIdx.x; t < (int)(sizeof(Phase) / 4); t += 256) dst[t] = src[t];
    }
    __syncthreads();
    switch (ph.kind) {
      case PH_GEMV:
        if (ph.wt == MI355_W_FP8) {
          if (ph.M == 1) phase_gemv<1, MI355_W_FP8>(ph, bf, xs, st, offset);
          else if (ph.M == 2) phase_gemv<2, MI355_W_FP8>(ph, bf, xs, st, offset);
          else if (ph.M <= 4) phase_gemv<4, MI355_W_FP8>(ph, bf, xs, st, offset);
          else phase_gemv<8, MI355_W_FP8>(ph, bf, xs, st, offset);
        } else if (ph.wt == MI355_W_F16) {
          if (ph.M == 1) phase_gemv<1, MI355_W_F16>(ph, bf, xs, st, offset);
          else if (ph.M == 2) phase_gemv<2, MI355_W_F16>(ph, bf, xs, st, offset);
          else if (ph.M <= 4) phase_gemv<4, MI355_W_F16>(ph, bf, xs, st, offset);
          else phase_gemv<8, MI355_W_F16>(ph, bf, xs, st, offset);
        } else {
          if (ph.M == 1) phase_gemv<1, MI355_W_BF16>(ph, bf, xs, st, offset);
          else if (ph.M == 2) phase_gemv<2, MI355_W_BF16>(ph, bf, xs, st, offset);
          else if (ph.M <= 4) phase_gemv<4, MI355_W_BF16>(ph, bf, xs, st, offset);
          else phase_gemv<8, MI355_W_BF16>(ph, bf, xs, st, offset);
        }
        break;
      case PH_ROPE: phase_rope(ph, bf, offset); break;
      case PH_ATTN:
        if (ph.dh == 64) phase_attn<64>(ph, bf, offset);
        else phase_attn<128>(ph, bf, offset);
        break;
      default: phase_norm(ph, bf, st); break;
    }
    if (i + 1 < nphases) {
      if (!grid_barrier(cnt, base + (uint32_t)(i + 1) * gridDim.x, err)) return;
    }
  }
}

// ================================================================================================ host side
struct Program {
  uint64_t key = 0;
  Phase* dev = nullptr;
  int nphases = 0;
  int capacity = 0;
  uint32_t* cnt = nullptr;
  int32_t* err = nullptr;
  uint32_t base = 0;
  uint64_t last_use = 0;
};

std::mutex g_mu;
std::vector<Program> g_programs;
uint64_t g_clock = 0;
int g_grid = 0;
int g_enabled = -1;

uint64_t fnv(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

Phase gemv_phase(int x, int ldx, int M, int K, const uint16_t* w, int N, int wdtype, const float* bias, int act, const float* colscale,
                 int res, int ldr, int glu, int y, int ldy, int norm, const float* nw, const float* nb, float eps, const float* wscale) {
  Phase p;
  memset(&p, 0, sizeof(p));
  p.kind = PH_GEMV; p.x_id = x; p.ldx = ldx; p.M = M; p.K = K; p.w = w; p.N = N; p.wt = wdtype; p.wscale = wdtype == MI355_W_FP8 ? wscale : nullptr; p.bias = bias; p.act = act;
  p.colscale = colscale; p.res_id = res; p.ldr = ldr; p.glu = glu; p.y_id = y; p.ldy = ldy; p.norm = norm; p.nw = nw; p.nb = nb; p.eps = eps;
  return p;
}

Phase attn_phase(int q, int ldq, const float* k, const float* v, int64_t kv_bstride, int ldkv, int64_t hstride, int heads, int kv_heads,
                 int dh, int Tk_base, int add_off, int window, float scale, int B, int out, int ldo) {
  Phase p;
  memset(&p, 0, sizeof(p));
  p.kind = PH_ATTN; p.q_id = q; p.ldq = ldq; p.kc = k; p.vc = v; p.kv_bstride = kv_bstride; p.ldkv = ldkv; p.hstride = hstride; p.heads = heads;
  p.kv_heads = kv_heads; p.dh = dh; p.Tk_base = Tk_base; p.tk_add_offset = add_off; p.window = window; p.scale = scale; p.B = B; p.out_id = out;
  p.ldo = ldo;
  return p;
}

}  // namespace

// 1 = the stack qualifies for the one-launch runner (everything else keeps the multi-launch schedule of stack_step.cpp)
extern "C" int mi355_stack_fused_eligible(const mi355_stack_desc* dp, int32_t B) {
  if (!dp || !dp->layers || B < 1 || B > 8) return 0;
  const mi355_stack_desc& d = *dp;
  if (d.n_layers <= 0 || d.d_model % 8 || d.d_ff % 8 || (d.dh != 64 && d.dh != 128)) return 0;
  if (d.norm != 1 && d.norm != 2) return 0;
  if (d.wdtype != MI355_W_BF16 && d.wdtype != MI355_W_F16 && d.wdtype != MI355_W_FP8) return 0;
  if (d.wdtype == MI355_W_FP8 && (d.d_model % 16 || d.d_ff % 16)) return 0;
  const int MT = B == 1 ? 1 : (B == 2 ? 2 : (B <= 4 ? 4 : 8));
  if (MT * ((d.d_model + 511) & ~511) > kXsCap) return 0;  // the fused pre-norm needs whole rows in one LDS chunk
  if (d.heads % d.kv_heads) return 0;
  if (!d.causal && d.window > 0) return 0;
  return 1;
}

extern "C" int mi355_stack_fused_set(int32_t enabled) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int prev = g_enabled;
  g_enabled = enabled ? 1 : 0;
  return prev;
}

extern "C" int mi355_stack_fused_enabled(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_enabled < 0) {
    // Opt-in: measured on MI355X (profiles/r1_decode_runner_ab_call21.txt) the phase program is 1.7-2.0x SLOWER than the multi-launch runner
    // (CSM-1B 14.8 vs 7.4 ms per frame, Qwen3-TTS-1.7B 17.9 vs 10.8, Whisper-small 3.17 vs 1.68 ms per token step): ~17 us per phase
    // (grid barrier + 4 waves per CU at 384 registers + scalar write-through stores) against ~7.6 us per launch.
    const char* e = getenv("MI355_STEP_FUSED");
    g_enabled = (e && e[0] == '1') ? 1 : 0;
  }
  return g_enabled;
}

// Synchronises the stream and reports (then clears) the error flags of every cached program: non-zero = a grid barrier was abandoned.
extern "C" int mi355_stack_fused_check(void* stream) {
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  MI355_REQUIRE(e == hipSuccess, "stack_fused_check: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(g_mu);
  int bad = 0;
  for (auto& pr : g_programs) {
    int32_t v = 0;
    e = hipMemcpy(&v, pr.err, sizeof(v), hipMemcpyDeviceToHost);
    MI355_REQUIRE(e == hipSuccess, "stack_fused_check: %s", hipGetErrorString(e));
    if (v) {  // the abandoned barrier left the counter short of its target: start this program's count over
      bad = 1;
      (void)hipMemset(pr.cnt, 0, 256);
      pr.base = 0;
    }
  }
  MI355_REQUIRE(!bad, "stack_decode_step(fused): a grid barrier timed out (workgroups of the step kernel were not co-resident)");
  return MI355_OK;
}

extern "C" int mi355_stack_decode_step_fused(const mi355_stack_desc* dp, float* x, int32_t B, int32_t offset, float* ws, float* out, void* stream) {
  MI355_REQUIRE(dp && x && ws && dp->layers, "stack_decode_step(fused): null argument");
  MI355_REQUIRE(mi355_stack_fused_eligible(dp, B), "stack_decode_step(fused): this stack / batch does not qualify");
  MI355_REQUIRE(offset >= 0, "stack_decode_step(fused): negative offset");
  const mi355_stack_desc d = *dp;
  const int D = d.d_model, H = d.heads, G = d.kv_heads, dh = d.dh;
  const int nq = H * dh, nkv = 2 * G * dh;
  for (int i = 0; i < d.n_layers; ++i) {
    const mi355_layer_desc& L = d.layers[i];
    MI355_REQUIRE(L.wqkv && L.wo && L.w_in && L.w_out && L.kv, "stack_decode_step(fused): layer %d is missing a tensor", i);
    MI355_REQUIRE(offset < L.kv_capacity, "stack_decode_step(fused): KV cache of layer %d is full (offset %d, capacity %d)", i, offset, L.kv_capacity);
  }
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_grid == 0) {
    int dev = 0, cus = 0, occ = 0;
    hipError_t e = hipGetDevice(&dev);
    MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): %s", hipGetErrorString(e));
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    MI355_REQUIRE(e == hipSuccess && cus > 0, "stack_decode_step(fused): cannot read the CU count");
    e = hipFuncSetAttribute((const void*)step_program_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kXsCap * 4);
    MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): cannot reserve LDS: %s", hipGetErrorString(e));
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)step_program_kernel, 256, kXsCap * 4);
    MI355_REQUIRE(e == hipSuccess && occ >= 1, "stack_decode_step(fused): occupancy query failed");
    g_grid = cus * (occ >= 2 ? 2 : 1);
    const char* ge = getenv("MI355_STEP_FUSED_WGS_PER_CU");
    if (ge && atoi(ge) == 1) g_grid = cus;
  }
  // ---- program lookup (content hash of everything the phase list depends on)
  uint64_t key = fnv(&d, sizeof(d) - 0, 1469598103934665603ull);
  key = fnv(d.layers, sizeof(mi355_layer_desc) * (size_t)d.n_layers, key);
  const int32_t kb[2] = {B, (out && d.final_norm_w) ? 1 : 0};
  key = fnv(kb, sizeof(kb), key);
  Program* prog = nullptr;
  for (auto& pr : g_programs)
    if (pr.key == key) { prog = &pr; break; }
  if (!prog) {
    std::vector<Phase> ph;
    const int x_ = BUF_X, q = BUF_Q, att = BUF_ATT, mid = BUF_MID, none = -1;
    const float scale = d.attn_scale > 0.f ? d.attn_scale : 1.0f / sqrtf((float)dh);
    for (int i = 0; i < d.n_layers; ++i) {
      const mi355_layer_desc& L = d.layers[i];
      Phase g1 = gemv_phase(x_, D, B, D, L.wqkv, nq + nkv, d.wdtype, L.bqkv, MI355_ACT_NONE, nullptr, none, 0, 0, q, nq, d.norm, L.attn_norm_w,
                            L.attn_norm_b, d.eps, L.s_qkv);
      g1.y2 = L.kv; g1.ldy2 = (int)L.kv_bstride; g1.split = nq; g1.y2_step = nkv;
      ph.push_back(g1);
      if (L.q_norm || d.cos) {
        Phase r;
        memset(&r, 0, sizeof(r));
        r.kind = PH_ROPE; r.x_id = q; r.ldx = nq; r.heads = H; r.heads2 = G; r.dh = dh; r.rope_mode = d.rope_mode; r.B = B; r.qnw = L.q_norm;
        r.knw = L.k_norm; r.cos_t = d.cos; r.sin_t = d.sin; r.eps = d.eps; r.k2 = L.kv; r.k2_bstride = L.kv_bstride; r.k2_step = nkv;
        ph.push_back(r);
      }
      ph.push_back(attn_phase(q, nq, L.kv, L.kv + G * dh, L.kv_bstride, nkv, 0, H, G, dh, 1, 1, d.window, scale, B, att, nq));
      ph.push_back(gemv_phase(att, nq, B, nq, L.wo, D, d.wdtype, L.bo, MI355_ACT_NONE, L.ls1, x_, D, 0, x_, D, 0, nullptr, nullptr, 0.f, L.s_o));
      if (L.cross_k) {
        MI355_REQUIRE(L.wcq && L.wco && L.cross_v && L.cross_len > 0, "stack_decode_step(fused): layer %d has cross K but no cross projections / V", i);
        ph.push_back(gemv_phase(x_, D, B, D, L.wcq, nq, d.wdtype, L.bcq, MI355_ACT_NONE, nullptr, none, 0, 0, q, nq, d.norm, L.cross_norm_w,
                                L.cross_norm_b, d.eps, L.s_cq));
        ph.push_back(attn_phase(q, nq, L.cross_k, L.cross_v, L.cross_bstride, L.cross_ld, L.cross_hstride, H, G, dh, L.cross_len, 0, 0, scale, B, att,
                                nq));
        ph.push_back(gemv_phase(att, nq, B, nq, L.wco, D, d.wdtype, L.bco, MI355_ACT_NONE, nullptr, x_, D, 0, x_, D, 0, nullptr, nullptr, 0.f, L.s_co));
      }
      ph.push_back(gemv_phase(x_, D, B, D, L.w_in, d.glu ? 2 * d.d_ff : d.d_ff, d.wdtype, L.b_in, d.glu ? MI355_ACT_NONE : d.act, nullptr, none, 0,
                              d.glu, mid, d.d_ff, d.norm, L.mlp_norm_w, L.mlp_norm_b, d.eps, L.s_in));
      ph.push_back(gemv_phase(mid, d.d_ff, B, d.d_ff, L.w_out, D, d.wdtype, L.b_out, MI355_ACT_NONE, L.ls2, x_, D, 0, x_, D, 0, nullptr, nullptr, 0.f, L.s_out));
    }
    if (out && d.final_norm_w) {
      Phase n;
      memset(&n, 0, sizeof(n));
      n.kind = PH_NORM; n.x_id = BUF_X; n.ldx = D; n.M = B; n.K = D; n.norm = d.norm; n.nw = d.final_norm_w; n.nb = d.norm == 1 ? d.final_norm_b : nullptr;
      n.eps = d.eps; n.y_id = BUF_OUT; n.ldy = D;
      ph.push_back(n);
    }
    for (const Phase& p : ph) {
      if (p.kind != PH_GEMV) continue;
      MI355_REQUIRE(p.K % 8 == 0 && ((uintptr_t)p.w) % 16 == 0 && p.ldx % 2 == 0, "stack_decode_step(fused): misaligned GEMV operand");
      MI355_REQUIRE(!p.glu || p.N % 2 == 0, "stack_decode_step(fused): SwiGLU needs an even N");
      MI355_REQUIRE(p.wt != MI355_W_FP8 || (p.wscale && p.K % 16 == 0), "stack_decode_step(fused): fp8 image without scales / K not a multiple of 16");
    }
    // slot: reuse the least recently used entry once 32 programs are cached
    if (g_programs.size() < 32) {
      g_programs.emplace_back();
      prog = &g_programs.back();
    } else {
      prog = &g_programs[0];
      for (auto& pr : g_programs)
        if (pr.last_use < prog->last_use) prog = &pr;
      hipError_t es = hipStreamSynchronize(st);  // the evicted program may still be in flight
      MI355_REQUIRE(es == hipSuccess, "stack_decode_step(fused): %s", hipGetErrorString(es));
    }
    if (prog->capacity < (int)ph.size()) {
      if (prog->dev) (void)hipFree(prog->dev);
      hipError_t e = hipMalloc((void**)&prog->dev, sizeof(Phase) * ph.size());
      MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): cannot allocate the phase list: %s", hipGetErrorString(e));
      prog->capacity = (int)ph.size();
    }
    if (!prog->cnt) {
      hipError_t e = hipMalloc((void**)&prog->cnt, 256);
      MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): cannot allocate the barrier counter: %s", hipGetErrorString(e));
      e = hipMemset(prog->cnt, 0, 256);
      MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): %s", hipGetErrorString(e));
      prog->err = (int32_t*)(prog->cnt + 32);
      prog->base = 0;
    }
    hipError_t e = hipMemcpy(prog->dev, ph.data(), sizeof(Phase) * ph.size(), hipMemcpyHostToDevice);  // blocking: once per (stack, buffers, B)
    MI355_REQUIRE(e == hipSuccess, "stack_decode_step(fused): cannot upload the phase list: %s", hipGetErrorString(e));
    prog->nphases = (int)ph.size();
    prog->key = key;
  }
  prog->last_use = ++g_clock;
  MI355_CLEAR_ERROR();
  Bufs bf;
  bf.p[BUF_X] = x; bf.p[BUF_Q] = ws; bf.p[BUF_ATT] = ws + (size_t)B * nq; bf.p[BUF_MID] = ws + (size_t)2 * B * nq; bf.p[BUF_OUT] = out;
  MI355_REQUIRE(((uintptr_t)x) % 8 == 0 && ((uintptr_t)ws) % 8 == 0, "stack_decode_step(fused): x / ws must be 8-byte aligned");
  hipLaunchKernelGGL(step_program_kernel, dim3(g_grid), dim3(256), kXsCap * 4, st, prog->dev, prog->nphases, (int)offset, prog->cnt, prog->base,
                     prog->err, bf);
  MI355_LAUNCH_CHECK("stack_decode_step(fused)");
  prog->base += (uint32_t)(prog->nphases - 1) * (uint32_t)g_grid;
  return MI355_OK;
}
