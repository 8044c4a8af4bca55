// Token sampling on device for the codec-token language models (Qwen3-TTS talker / code predictor, CSM), gfx950.
//
// Replaces the per-frame chain of Model._sample_token / _sample_token_batch (tts/models/qwen3_tts/qwen3_tts.py:805-925):
// suppress list -> repetition penalty over the generated tokens -> / temperature -> top-k (apply_top_k,
// lm/sample_utils.py:130-152) -> top-p / min-p on the log-softmax (_apply_probability_filters, qwen3_tts.py:47-61;
// apply_top_p / apply_min_p, lm/sample_utils.py:155-241) -> categorical (lm/sample_utils.py:279-281), which the reference
// runs as ~15 MLX ops with a Python loop over sequences for the repetition penalty (qwen3_tts.py:896-914) and a host sync
// per frame.  One workgroup per sequence keeps the whole vocabulary row (V <= 8192: 3072 / 2048 / 2051 here) in LDS.
//
// Randomness is explicit: mx.random.categorical(logits) is the Gumbel-max trick, so the caller passes Gumbel(0,1) noise
// [B, V] (null => arg-max; temperature <= 0 => arg-max of the penalised logits like the reference).  Ties in top-k are
// resolved towards the lower index (mx.argpartition leaves them unspecified).
#include "common.h"

namespace {

constexpr int kT = 1024;
constexpr int kMaxV = 8192;

__device__ __forceinline__ float blk_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < kT / 64; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < kT / 64; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ int blk_sum_i(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = 0;
  for (int i = 0; i < kT / 64; ++i) r += red[i];
  return r;
}
// order-preserving map float -> uint32 (larger float => larger key); -inf maps below every finite value
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kT) void sample_kernel(const mi355_sample_args a) {
  __shared__ float lg[kMaxV];
  __shared__ float red[kT / 64];
  __shared__ int redi[kT / 64];
  __shared__ int best_i;
  const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
  const float NEG = -INFINITY;
  const float* src = a.logits + (int64_t)b * a.ld;
  // 1. suppress list
  for (int v = tid; v < V; v += kT) lg[v] = src[v] + (a.suppress_mask ? a.suppress_mask[v] : 0.f);
  __syncthreads();
  // 2. repetition penalty over the set of generated tokens (each distinct token once)
  if (a.history && a.repetition_penalty != 1.0f) {
    const int n = a.hist_len ? a.hist_len[b] : a.n_hist;
    const int32_t* h = a.history + (int64_t)b * a.hist_ld;
    for (int i = tid; i < n; i += kT) {
      const int t = h[i];
      bool first = t >= 0 && t < V;
      for (int j = 0; first && j < i; ++j) first = h[j] != t;  // n is at most a few thousand frames
      if (first) { const float x = lg[t]; lg[t] = x < 0.f ? x * a.repetition_penalty : x / a.repetition_penalty; }
    }
    __syncthreads();
  }
  if (a.temperature > 0.f) {
    if (a.temperature != 1.0f)
      for (int v = tid; v < V; v += kT) lg[v] = lg[v] / a.temperature;
    __syncthreads();
    // 3. top-k: radix-select the k-th largest key, keep keys above it plus the first ties in index order
    if (a.top_k > 0 && a.top_k < V) {
      uint32_t prefix = 0;
      int need = a.top_k;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        const uint32_t mask = ~((1u << bit) - 1u);
        int c = 0;
        for (int v = tid; v < V; v += kT) c += (fkey(lg[v]) & mask) == cand ? 1 : 0;  // prefix matches and this bit set
        c = blk_sum_i(c, redi);
        if (c >= need) prefix = cand; else need -= c;
      }
      // prefix == key of the k-th largest value; `need` of the entries equal to it survive
      int ties = 0;
      for (int v = tid; v < V; v += kT) ties += fkey(lg[v]) == prefix ? 1 : 0;
      ties = blk_sum_i(ties, redi);
      if (ties > need) {
        __syncthreads();
        if (tid == 0) {
          int kept = 0;
          for (int v = 0; v < V; ++v)
            if (fkey(lg[v]) == prefix) { if (kept >= need) lg[v] = NEG; ++kept; }
        }
        __syncthreads();
      }
      for (int v = tid; v < V; v += kT) if (fkey(lg[v]) < prefix) lg[v] = NEG;
      __syncthreads();
    }
    // 4. top-p / min-p on log-softmax
    const bool use_p = a.top_p > 0.f && a.top_p < 1.f;
    if (use_p || a.min_p > 0.f) {
      float mx = NEG;
      for (int v = tid; v < V; v += kT) mx = fmaxf(mx, lg[v]);
      mx = blk_max(mx, red);
      float s = 0.f;
      for (int v = tid; v < V; v += kT) s += expf(lg[v] - mx);
      s = blk_sum(s, red);
      const float lse = mx + logf(s);
      bool kill[kMaxV / kT];
#pragma unroll
      for (int i = 0; i < kMaxV / kT; ++i) kill[i] = false;
      if (use_p) {
        // ascending cumulative probability of every entry (stable by index), keep where cum > 1 - top_p
        for (int i = 0, v = tid; v < V; v += kT, ++i) {
          const float lv = lg[v];
          if (lv == NEG) continue;
          float cum = 0.f;
          for (int u = 0; u < V; ++u) {
            const float lu = lg[u];
            if (lu < lv || (lu == lv && u <= v)) cum += expf(lu - lse);
          }
          kill[i] = !(cum > 1.0f - a.top_p);
        }
      }
      if (a.min_p > 0.f) {
        const float thr = (mx - lse) + logf(a.min_p);
        for (int i = 0, v = tid; v < V; v += kT, ++i)
          if ((lg[v] - lse) < thr) kill[i] = true;
      }
      __syncthreads();
      for (int i = 0, v = tid; v < V; v += kT, ++i) if (kill[i]) lg[v] = NEG;
      __syncthreads();
    }
  }
  if (a.filtered)
    for (int v = tid; v < V; v += kT) a.filtered[(int64_t)b * a.ld + v] = lg[v];
  // 5. categorical via Gumbel-max (or arg-max)
  const bool noisy = a.gumbel && a.temperature > 0.f;
  float bv = NEG;
  int bi = 0x7fffffff;
  for (int v = tid; v < V; v += kT) {
    const float x = noisy ? lg[v] + a.gumbel[(int64_t)b * a.ld + v] : lg[v];
    if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = bv; redi[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < kT / 64; ++i)
      if (red[i] > bv || (red[i] == bv && redi[i] < bi)) { bv = red[i]; bi = redi[i]; }
    int tokv = bi;
    if (a.done && a.done[b]) tokv = a.done_token;  // finished rows keep emitting the EOS (qwen3_tts.py:1881-1887)
    a.out[(int64_t)b * a.out_ld] = tokv;
    best_i = tokv;
  }
  (void)best_i;
}

}  // namespace

extern "C" int mi355_sample(const mi355_sample_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->logits && ap->out, "sample: null tensor");
  const mi355_sample_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.V > 0 && a.V <= kMaxV && a.ld >= a.V, "sample: vocabulary must be in [1, %d]", kMaxV);
  MI355_REQUIRE(a.top_p >= 0.f && a.top_p <= 1.f && a.min_p >= 0.f && a.min_p <= 1.f, "sample: top_p / min_p must be in [0, 1]");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(kT), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("sample");
  return MI355_OK;
}
