// Multi-head self-attention for short sequences (PL-BERT inside Kokoro, T <= 512), fp32 (gfx950).
// Replaces the explicit QK^T / softmax / PV of AlbertSelfAttention
// (tts/models/kokoro/modules.py:493-508).  One wavefront per (query, head, utterance): lanes own
// keys for the score / softmax phase (wave-level max and sum reductions) and own head channels for
// the PV phase; probabilities are handed over through LDS.
#include "common.h"

namespace {

constexpr int kMaxT = 512;

// LDS hand-over between lanes of ONE wavefront (waves of a block may exit early, so no s_barrier)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void attention_kernel(const mi355_attention_args a) {
  __shared__ float qs[4][64];
  __shared__ float ps[4][kMaxT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  const int len = a.lens ? a.lens[b] : a.T;
  if (q >= len) return;  // whole wave exits together; no block-level sync below
  const int D = a.heads * a.dh;
  const float* base = a.qkv + (int64_t)b * a.bstride;
  const float* qrow = base + (int64_t)q * a.ld + h * a.dh;
  if (lane < a.dh) qs[w][lane] = qrow[lane];
  wave_lds_sync();
  const float inv = sqrtf((float)a.dh);
  float sc[kMaxT / 64];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kMaxT / 64; ++i) {
    const int j = i * 64 + lane;
    sc[i] = -INFINITY;
    if (j < len) {
      const float* krow = base + (int64_t)j * a.ld + D + h * a.dh;
      float s = 0.f;
      for (int d = 0; d < a.dh; d += 4) {
        const float4 kv = *(const float4*)(krow + d);
        s = fmaf(qs[w][d], kv.x, s);
        s = fmaf(qs[w][d + 1], kv.y, s);
        s = fmaf(qs[w][d + 2], kv.z, s);
        s = fmaf(qs[w][d + 3], kv.w, s);
      }
      sc[i] = s / inv;
      mx = fmaxf(mx, sc[i]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxT / 64; ++i) {
    const int j = i * 64 + lane;
    if (j < len) {
      const float p = expf(sc[i] - mx);
      sum += p;
      ps[w][j] = p;
    }
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < kMaxT / 64; ++i) {
    const int j = i * 64 + lane;
    if (j < len) ps[w][j] = ps[w][j] / sum;  // softmax, then probs @ V like the reference
  }
  wave_lds_sync();
  if (lane < a.dh) {
    const float* vcol = base + 2 * D + h * a.dh + lane;
    float o = 0.f;
    for (int j = 0; j < len; ++j) o = fmaf(ps[w][j], vcol[(int64_t)j * a.ld], o);
    a.out[(int64_t)b * a.out_bstride + (int64_t)q * a.ldo + h * a.dh + lane] = o;
  }
}

}  // namespace

extern "C" int mi355_attention(const mi355_attention_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->qkv && ap->out, "attention: null tensor");
  const mi355_attention_args a = *ap;
  MI355_REQUIRE(a.T > 0 && a.T <= kMaxT, "attention: T must be in [1, %d] (got %d)", kMaxT, a.T);
  MI355_REQUIRE(a.dh > 0 && a.dh <= 64 && a.dh % 4 == 0, "attention: head dim must be a multiple of 4 and <= 64");
  MI355_REQUIRE(a.ld % 4 == 0 && a.bstride % 4 == 0, "attention: strides must be multiples of 4");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(attention_kernel, dim3((a.T + 3) / 4, a.heads, a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("attention");
  return MI355_OK;
}
