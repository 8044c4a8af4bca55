// Whisper greedy decode step on device: logit filters + token selection + log-prob bookkeeping (gfx950).
//
// Replaces, per generated token, the chain SuppressBlank.apply -> SuppressTokens.apply -> ApplyTimestampRules.apply ->
// GreedyDecoder.update (stt/models/whisper/decoding.py:333-443, 302-330) that the reference evaluates as ~20 separate MLX
// ops plus a `tokens.tolist()` device->host round trip per step (decoding.py:390).  One workgroup per sequence walks the
// vocabulary row (V = 51865) five times out of L2; nothing returns to the host, so the decode loop can be enqueued for a
// fixed number of steps without synchronising.
//
// Reference quirk kept on purpose: ApplyTimestampRules builds `timestamps` as the *indices* i of sampled tokens above
// timestamp_begin and then masks `timestamp_begin : last_timestamp` with that small index (decoding.py:410-419), which is
// an empty slice -- the "timestamps must not decrease" rule of the original OpenAI implementation is a no-op in the
// reference, and therefore here.
#include "common.h"

namespace {

constexpr int kT = 1024;

struct ArgMax { float v; int i; };

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < kT / 64; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < kT / 64; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
  // larger value wins; on ties the smaller index (mx.argmax returns the first maximum)
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax x, float* redv, int* redi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    x = better(x, y);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { redv[w] = x.v; redi[w] = x.i; }
  __syncthreads();
  ArgMax r;
  r.v = redv[0]; r.i = redi[0];
  for (int i = 1; i < kT / 64; ++i) { ArgMax y; y.v = redv[i]; y.i = redi[i]; r = better(r, y); }
  return r;
}

__global__ __launch_bounds__(kT) void whisper_greedy_step_kernel(const mi355_whisper_step_args a) {
  __shared__ float red[kT / 64];
  __shared__ int redi[kT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = a.logits + (int64_t)b * a.ld;
  int32_t* tk = a.tokens + (int64_t)b * a.tokens_ld;
  const int n = a.n, nseq = n - a.sample_begin;
  const int last = n >= 1 ? tk[n - 1] : -1;
  const bool first = n == a.sample_begin;
  const bool last_ts = nseq >= 1 && last >= a.timestamp_begin;
  const bool pen_ts = nseq < 2 || tk[n - 2] >= a.timestamp_begin;
  const float NEG = -INFINITY;

  auto l1 = [&](int v) -> float {  // logits after SuppressBlank and SuppressTokens
    float x = lg[v];
    if (first && a.blank_ids)
      for (int i = 0; i < a.n_blank; ++i) if (a.blank_ids[i] == v) x = NEG;
    if (a.suppress_mask) x += a.suppress_mask[v];
    return x;
  };
  auto mask2 = [&](int v) -> float {  // ApplyTimestampRules, before the timestamp-dominance rule
    if (!a.timestamp_rules) return 0.f;
    if (v == a.no_timestamps) return NEG;
    if (last_ts) {
      if (pen_ts) { if (v >= a.timestamp_begin) return NEG; }
      else if (v < a.eot) return NEG;
    }
    if (first) {
      if (v < a.timestamp_begin) return NEG;
      if (a.max_initial_timestamp_index >= 0 && v > a.timestamp_begin + a.max_initial_timestamp_index) return NEG;
    }
    return 0.f;
  };

  bool text_killed = false;
  if (a.timestamp_rules) {
    // logprobs = l1 - logsumexp(l1); compare logsumexp(logprobs[ts:]) with max(logprobs[:ts])
    float mx = NEG;
    for (int v = tid; v < a.V; v += kT) mx = fmaxf(mx, l1(v));
    mx = block_max(mx, red);
    float s = 0.f;
    for (int v = tid; v < a.V; v += kT) s += expf(l1(v) - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    float mts = NEG, mtext = NEG;
    for (int v = tid; v < a.V; v += kT) {
      const float lp = l1(v) - lse;
      if (v >= a.timestamp_begin) mts = fmaxf(mts, lp); else mtext = fmaxf(mtext, lp);
    }
    mts = block_max(mts, red);
    mtext = block_max(mtext, red);
    float sts = 0.f;
    for (int v = a.timestamp_begin + tid; v < a.V; v += kT) sts += expf((l1(v) - lse) - mts);
    sts = block_sum(sts, red);
    const float ts_lp = mts + logf(sts);
    text_killed = ts_lp > mtext;
  }

  auto fin = [&](int v) -> float {
    float x = l1(v) + mask2(v);
    if (text_killed && v < a.timestamp_begin) x = NEG;
    return x;
  };

  ArgMax best; best.v = NEG; best.i = 0x7fffffff;
  ArgMax bsel; bsel.v = NEG; bsel.i = 0x7fffffff;
  float fmx = NEG;
  for (int v = tid; v < a.V; v += kT) {
    const float x = fin(v);
    if (a.filtered) a.filtered[(int64_t)b * a.ld + v] = x;
    fmx = fmaxf(fmx, x);
    ArgMax c; c.v = x; c.i = v;
    best = better(best, c);
    if (a.gumbel) { ArgMax d; d.v = x / a.temperature + a.gumbel[(int64_t)b * a.ld + v]; d.i = v; bsel = better(bsel, d); }
  }
  best = block_argmax(best, red, redi);
  if (a.gumbel) best = block_argmax(bsel, red, redi);
  fmx = block_max(fmx, red);
  float fs = 0.f;
  for (int v = tid; v < a.V; v += kT) fs += expf(fin(v) - fmx);
  fs = block_sum(fs, red);
  if (tid == 0) {
    const int chosen = a.forced_next ? a.forced_next[b] : best.i;
    const float lp = fin(chosen) - (fmx + logf(fs));
    const bool done = last == a.eot;
    if (!done) a.sum_logprobs[b] += lp;
    tk[n] = done ? a.eot : chosen;
  }
}

// Same step with the row held in registers: the generic kernel above walks the V logits seven times out of L2 (max, sum, two maxima, timestamp
// sum, arg-max, final sum), re-applying the suppress lists on every pass -- 80 us per step for V = 51 865 with only B workgroups on the chip.
// Here a thread loads its NV = ceil(V / 1024) logits once (all loads in flight together), applies SuppressBlank / SuppressTokens once, and every
// later pass is register arithmetic plus a block reduction.  Same formulas and reduction helpers; only the per-thread grouping of the
// timestamp-mass sum differs (elements tid + j * 1024 instead of a walk that starts at timestamp_begin).
template <int NV>
__global__ __launch_bounds__(kT) void whisper_greedy_step_reg_kernel(const mi355_whisper_step_args a) {
  __shared__ float red[kT / 64];
  __shared__ int redi[kT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = a.logits + (int64_t)b * a.ld;
  int32_t* tk = a.tokens + (int64_t)b * a.tokens_ld;
  const int n = a.n, nseq = n - a.sample_begin;
  const int last = n >= 1 ? tk[n - 1] : -1;
  const bool first = n == a.sample_begin;
  const bool last_ts = nseq >= 1 && last >= a.timestamp_begin;
  const bool pen_ts = nseq < 2 || tk[n - 2] >= a.timestamp_begin;
  const float NEG = -INFINITY;
  float x1[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * kT;
    x1[j] = v < a.V ? lg[v] : NEG;
  }
  if (a.suppress_mask) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = tid + j * kT;
      if (v < a.V) x1[j] += a.suppress_mask[v];
    }
  }
  if (first && a.blank_ids) {
    for (int i = 0; i < a.n_blank; ++i) {
      const int bv = a.blank_ids[i];
      if (bv >= 0 && bv < a.V && (bv & (kT - 1)) == tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) if (j == (bv >> 10)) x1[j] = NEG;   // (the mask add above keeps -inf at -inf)
      }
    }
  }
  auto mask2 = [&](int v) -> float {  // ApplyTimestampRules, before the timestamp-dominance rule
    if (!a.timestamp_rules) return 0.f;
    if (v == a.no_timestamps) return NEG;
    if (last_ts) {
      if (pen_ts) { if (v >= a.timestamp_begin) return NEG; }
      else if (v < a.eot) return NEG;
    }
    if (first) {
      if (v < a.timestamp_begin) return NEG;
      if (a.max_initial_timestamp_index >= 0 && v > a.timestamp_begin + a.max_initial_timestamp_index) return NEG;
    }
    return 0.f;
  };
  bool text_killed = false;
  if (a.timestamp_rules) {
    float mx = NEG;
#pragma unroll
    for (int j = 0; j < NV; ++j) mx = fmaxf(mx, x1[j]);
    mx = block_max(mx, red);
    // ONE exponential pass feeds both sums: the total (-> logsumexp) and the timestamp mass.  logsumexp(logprobs[ts:]) =
    // log(sum_ts exp(x - mx)) + mx - lse, the same quantity the generic kernel forms around the timestamp maximum (decoding.py:428-436).
    float s = 0.f, s_ts = 0.f, mtext = NEG;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = tid + j * kT;
      if (v < a.V) {
        const float e = expf(x1[j] - mx);
        s += e;
        if (v >= a.timestamp_begin) s_ts += e; else mtext = fmaxf(mtext, x1[j]);
      }
    }
    s = block_sum(s, red);
    s_ts = block_sum(s_ts, red);
    mtext = block_max(mtext, red);
    const float lse = mx + logf(s);
    const float mts = mx - lse, sts = s_ts;   // (names of the generic kernel: ts_lp = mts + log(sts))
    mtext -= lse;
    text_killed = mts + logf(sts) > mtext;
  }
  ArgMax best; best.v = NEG; best.i = 0x7fffffff;
  ArgMax bsel; bsel.v = NEG; bsel.i = 0x7fffffff;
  float fmx = NEG;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * kT;
    if (v < a.V) {
      float x = x1[j] + mask2(v);
      if (text_killed && v < a.timestamp_begin) x = NEG;
      x1[j] = x;
      if (a.filtered) a.filtered[(int64_t)b * a.ld + v] = x;
      fmx = fmaxf(fmx, x);
      ArgMax c; c.v = x; c.i = v;
      best = better(best, c);
      if (a.gumbel) { ArgMax d; d.v = x / a.temperature + a.gumbel[(int64_t)b * a.ld + v]; d.i = v; bsel = better(bsel, d); }
    }
  }
  best = block_argmax(best, red, redi);
  if (a.gumbel) best = block_argmax(bsel, red, redi);
  fmx = block_max(fmx, red);
  float fs = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) if (tid + j * kT < a.V) fs += expf(x1[j] - fmx);
  fs = block_sum(fs, red);
  // the chosen token's filtered logit lives in one thread's registers: hand it over through LDS
  __shared__ float chosen_x;
  const int chosen = a.forced_next ? a.forced_next[b] : best.i;
  if (tid == 0) chosen_x = -INFINITY;   // an out-of-range forced id contributes log p = -inf instead of uninitialised shared memory
  __syncthreads();
  if (chosen >= 0 && chosen < a.V && (chosen & (kT - 1)) == tid) {
#pragma unroll
    for (int j = 0; j < NV; ++j) if (j == (chosen >> 10)) chosen_x = x1[j];
  }
  __syncthreads();
  if (tid == 0) {
    const float lp = chosen_x - (fmx + logf(fs));
    const bool done = last == a.eot;
    if (!done) a.sum_logprobs[b] += lp;
    tk[n] = done ? a.eot : chosen;
  }
}

// ---------------------------------------------------------------------------------------------------- a row spread over kS workgroups
// One workgroup per row leaves the step on B CUs for ~57 us (three transcendental passes over 51 865 logits on one CU).  Here a row is cut into kS
// slices: phase A (B x kS workgroups) reduces each slice to (max, sum exp, timestamp sum exp, text max); phase B recombines the kS records of its
// row (every workgroup redundantly: 64 floats), applies the rules to its slice, reduces it to (arg-max, max, sum exp, Gumbel arg-max) and takes a
// ticket; the workgroup that draws the last ticket of a row merges the kS records in slice order (deterministic) and writes the token and the
// log-probability.  Nobody waits for anybody (no spinning): nothing to deadlock on.
constexpr int kS = 16, kTS = 256;

struct StepCtx {
  const float* lg; const int32_t* tk; int n, nseq, last; bool first, last_ts, pen_ts;
};

__device__ __forceinline__ StepCtx step_ctx(const mi355_whisper_step_args& a, int b) {
  StepCtx c;
  c.lg = a.logits + (int64_t)b * a.ld;
  c.tk = a.tokens + (int64_t)b * a.tokens_ld;
  c.n = a.n; c.nseq = a.n - a.sample_begin;
  c.last = a.n >= 1 ? c.tk[a.n - 1] : -1;
  c.first = a.n == a.sample_begin;
  c.last_ts = c.nseq >= 1 && c.last >= a.timestamp_begin;
  c.pen_ts = c.nseq < 2 || c.tk[a.n - 2] >= a.timestamp_begin;
  return c;
}

__device__ __forceinline__ float step_l1(const mi355_whisper_step_args& a, const StepCtx& c, int v) {  // after SuppressBlank and SuppressTokens
  float x = c.lg[v];
  if (c.first && a.blank_ids)
    for (int i = 0; i < a.n_blank; ++i) if (a.blank_ids[i] == v) x = -INFINITY;
  if (a.suppress_mask) x += a.suppress_mask[v];
  return x;
}

__device__ __forceinline__ float step_mask2(const mi355_whisper_step_args& a, const StepCtx& c, int v) {  // ApplyTimestampRules, before the dominance rule
  if (!a.timestamp_rules) return 0.f;
  if (v == a.no_timestamps) return -INFINITY;
  if (c.last_ts) {
    if (c.pen_ts) { if (v >= a.timestamp_begin) return -INFINITY; }
    else if (v < a.eot) return -INFINITY;
  }
  if (c.first) {
    if (v < a.timestamp_begin) return -INFINITY;
    if (a.max_initial_timestamp_index >= 0 && v > a.timestamp_begin + a.max_initial_timestamp_index) return -INFINITY;
  }
  return 0.f;
}

__device__ __forceinline__ float blk_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kTS) void whisper_step_split_a_kernel(const mi355_whisper_step_args a) {
  __shared__ float red[4];
  const int b = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const StepCtx c = step_ctx(a, b);
  const int per = (a.V + kS - 1) / kS, v0 = sp * per, v1 = v0 + per < a.V ? v0 + per : a.V;
  float mx = -INFINITY, mtext = -INFINITY;
  for (int v = v0 + tid; v < v1; v += kTS) {
    const float x = step_l1(a, c, v);
    mx = fmaxf(mx, x);
    if (v < a.timestamp_begin) mtext = fmaxf(mtext, x);
  }
  mx = blk_max(mx, red);
  mtext = blk_max(mtext, red);
  float s = 0.f, s_ts = 0.f;
  if (mx > -INFINITY) {
    for (int v = v0 + tid; v < v1; v += kTS) {
      const float e = expf(step_l1(a, c, v) - mx);
      s += e;
      if (v >= a.timestamp_begin) s_ts += e;
    }
  }
  s = blk_sum(s, red);
  s_ts = blk_sum(s_ts, red);
  if (tid == 0) {
    float* rec = a.split_ws + ((int64_t)b * kS + sp) * 12;
    rec[0] = mx; rec[1] = s; rec[2] = s_ts; rec[3] = mtext;
  }
}

__global__ __launch_bounds__(kTS) void whisper_step_split_b_kernel(const mi355_whisper_step_args a) {
  __shared__ float red[4];
  __shared__ int redi[4];
  __shared__ float redv[4];
  const int b = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const StepCtx c = step_ctx(a, b);
  const float NEG = -INFINITY;
  // ---- the row's statistics from the kS phase-A records (same order in every workgroup)
  bool text_killed = false;
  if (a.timestamp_rules) {
    const float* recs = a.split_ws + (int64_t)b * kS * 12;
    float M = NEG, mtext = NEG;
    for (int i = 0; i < kS; ++i) { M = fmaxf(M, recs[i * 12]); mtext = fmaxf(mtext, recs[i * 12 + 3]); }
    float s = 0.f, s_ts = 0.f;
    for (int i = 0; i < kS; ++i) {
      const float w = recs[i * 12] == NEG ? 0.f : expf(recs[i * 12] - M);
      s += recs[i * 12 + 1] * w;
      s_ts += recs[i * 12 + 2] * w;
    }
    const float lse = M + logf(s);
    text_killed = (M - lse) + logf(s_ts) > mtext - lse;   // logsumexp(logprobs[ts:]) > max(logprobs[:ts])  (decoding.py:428-436)
  }
  // ---- the rules on this slice
  const int per = (a.V + kS - 1) / kS, v0 = sp * per, v1 = v0 + per < a.V ? v0 + per : a.V;
  ArgMax best; best.v = NEG; best.i = 0x7fffffff;
  ArgMax bsel; bsel.v = NEG; bsel.i = 0x7fffffff;
  float fmx = NEG;
  for (int v = v0 + tid; v < v1; v += kTS) {
    float x = step_l1(a, c, v) + step_mask2(a, c, v);
    if (text_killed && v < a.timestamp_begin) x = NEG;
    if (a.filtered) a.filtered[(int64_t)b * a.ld + v] = x;
    fmx = fmaxf(fmx, x);
    ArgMax cand; cand.v = x; cand.i = v;
    best = better(best, cand);
    if (a.gumbel) { ArgMax d; d.v = x / a.temperature + a.gumbel[(int64_t)b * a.ld + v]; d.i = v; bsel = better(bsel, d); }
  }
  auto blk_argmax = [&](ArgMax x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ArgMax y;
      y.v = __shfl_xor(x.v, o, 64);
      y.i = __shfl_xor(x.i, o, 64);
      x = better(x, y);
    }
    __syncthreads();
    if (lane == 0) { redv[wave] = x.v; redi[wave] = x.i; }
    __syncthreads();
    ArgMax r; r.v = redv[0]; r.i = redi[0];
    for (int i = 1; i < 4; ++i) { ArgMax y; y.v = redv[i]; y.i = redi[i]; r = better(r, y); }
    return r;
  };
  best = blk_argmax(best);
  if (a.gumbel) bsel = blk_argmax(bsel);
  fmx = blk_max(fmx, red);
  float fs = 0.f;
  if (fmx > NEG) {
    for (int v = v0 + tid; v < v1; v += kTS) {
      float x = step_l1(a, c, v) + step_mask2(a, c, v);
      if (text_killed && v < a.timestamp_begin) x = NEG;
      fs += expf(x - fmx);
    }
  }
  fs = blk_sum(fs, red);
  float* rec = a.split_ws + ((int64_t)b * kS + sp) * 12 + 4;
  if (tid == 0) {
    rec[0] = best.v; rec[1] = __int_as_float(best.i); rec[2] = fmx; rec[3] = fs; rec[4] = bsel.v; rec[5] = __int_as_float(bsel.i);
    __threadfence();
  }
  __syncthreads();
  __shared__ int ticket;
  if (tid == 0) ticket = atomicAdd(a.split_cnt + b, 1);
  __syncthreads();
  if (ticket != kS - 1) return;
  __threadfence();
  if (tid == 0) {
    const float* recs = a.split_ws + (int64_t)b * kS * 12 + 4;
    ArgMax B1; B1.v = NEG; B1.i = 0x7fffffff;
    ArgMax B2; B2.v = NEG; B2.i = 0x7fffffff;
    float FM = NEG;
    for (int i = 0; i < kS; ++i) {
      ArgMax y; y.v = __hip_atomic_load(recs + i * 12, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      y.i = __float_as_int(__hip_atomic_load(recs + i * 12 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      B1 = better(B1, y);
      ArgMax z; z.v = __hip_atomic_load(recs + i * 12 + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      z.i = __float_as_int(__hip_atomic_load(recs + i * 12 + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      B2 = better(B2, z);
      FM = fmaxf(FM, __hip_atomic_load(recs + i * 12 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    float FS = 0.f;
    for (int i = 0; i < kS; ++i) {
      const float m_i = __hip_atomic_load(recs + i * 12 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (m_i > NEG) FS += __hip_atomic_load(recs + i * 12 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * expf(m_i - FM);
    }
    const int chosen = a.forced_next ? a.forced_next[b] : (a.gumbel ? B2.i : B1.i);
    float xc = (chosen >= 0 && chosen < a.V) ? step_l1(a, c, chosen) + step_mask2(a, c, chosen) : -INFINITY;   // out-of-range forced id: no read
    if (text_killed && chosen < a.timestamp_begin) xc = NEG;
    const bool done = c.last == a.eot;
    if (!done) a.sum_logprobs[b] += xc - (FM + logf(FS));
    ((int32_t*)c.tk)[c.n] = done ? a.eot : chosen;
    a.split_cnt[b] = 0;   // leave the counter zeroed for the next launch
  }
}

__global__ __launch_bounds__(kT) void softmax_prob_at_kernel(const float* logits, int ld, int V, int token, float* out) {
  __shared__ float red[kT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = logits + (int64_t)b * ld;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += kT) mx = fmaxf(mx, lg[v]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int v = tid; v < V; v += kT) s += expf(lg[v] - mx);
  s = block_sum(s, red);
  if (tid == 0) out[b] = expf(lg[token] - mx) / s;
}

}  // namespace

extern "C" int mi355_whisper_greedy_step(const mi355_whisper_step_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->logits && ap->tokens && ap->sum_logprobs, "whisper_greedy_step: null tensor");
  const mi355_whisper_step_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.V > 0 && a.ld >= a.V, "whisper_greedy_step: bad shape");
  MI355_REQUIRE(a.n >= 1 && a.n < a.tokens_ld && a.sample_begin <= a.n, "whisper_greedy_step: token buffer too small or n < sample_begin");
  MI355_REQUIRE(!a.timestamp_rules || (a.timestamp_begin > 0 && a.timestamp_begin < a.V && a.eot >= 0), "whisper_greedy_step: bad timestamp ids");
  MI355_REQUIRE(!a.gumbel || a.temperature > 0.f, "whisper_greedy_step: sampling needs temperature > 0");
  MI355_CLEAR_ERROR();
  static_assert(kT == 1024, "the register kernel indexes with v & 1023 / v >> 10");
  if (a.split_ws && a.split_cnt && a.V >= 4096) {
    hipLaunchKernelGGL(whisper_step_split_a_kernel, dim3(kS, a.B), dim3(kTS), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(whisper_step_split_b_kernel, dim3(kS, a.B), dim3(kTS), 0, (hipStream_t)stream, a);
    MI355_LAUNCH_CHECK("whisper_greedy_step(split)");
    return MI355_OK;
  }
  if (a.V <= 52 * kT) hipLaunchKernelGGL(whisper_greedy_step_reg_kernel<52>, dim3(a.B), dim3(kT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(whisper_greedy_step_kernel, dim3(a.B), dim3(kT), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("whisper_greedy_step");
  return MI355_OK;
}

extern "C" int mi355_softmax_prob_at(const float* logits, int32_t ld, int32_t V, int32_t B, int32_t token, float* out, void* stream) {
  MI355_REQUIRE(logits && out && B > 0 && V > 0 && token >= 0 && token < V, "softmax_prob_at: bad arguments");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(softmax_prob_at_kernel, dim3(B), dim3(kT), 0, (hipStream_t)stream, logits, (int)ld, (int)V, (int)token, out);
  MI355_LAUNCH_CHECK("softmax_prob_at");
  return MI355_OK;
}
