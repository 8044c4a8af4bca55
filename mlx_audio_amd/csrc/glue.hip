// Small glue kernels of Kokoro's Model.__call__ (tts/models/kokoro/kokoro.py:111-177) that the
// reference expresses as gathers, one-hot matmuls and Python loops (gfx950).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const mi355_gather_rows_args a) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / a.L), l = (int)(row - (int64_t)b * a.L);
  const int len = a.lens ? a.lens[b] : a.L;
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)l * a.ldy;
  if (l >= len) {
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) yr[c] = 0.f;
    return;
  }
  const int src = a.idx[(int64_t)b * a.idx_ld + l];
  const float* tr = a.table + (int64_t)b * a.table_bstride + (int64_t)src * a.ld_table;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    float v = tr[c];
    if (a.pos_table) v += a.pos_table[(int64_t)l * a.ld_pos + c];
    if (a.add_row) v += a.add_row[c];
    yr[c] = v;
  }
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* v, int ldv, int C, float* y, int64_t ybs, int ldy,
                                                             int L, const int32_t* lens, int B) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / L), l = (int)(row - (int64_t)b * L);
  const int len = lens ? lens[b] : L;
  float* yr = y + (int64_t)b * ybs + (int64_t)l * ldy;
  for (int c = threadIdx.x; c < C; c += blockDim.x) yr[c] = (l < len) ? v[(int64_t)b * ldv + c] : 0.f;
}

// one block per utterance: duration head + exclusive scan + frame->token index
__global__ __launch_bounds__(512) void duration_align_kernel(const mi355_duration_args a) {
  __shared__ int sdur[512];
  __shared__ int sstart[513];
  const int b = blockIdx.x, t = threadIdx.x;
  const int len = a.lens ? a.lens[b] : a.T;
  int d = 0;
  if (t < len) {
    if (a.forced_dur) {
      d = a.forced_dur[(int64_t)b * a.T + t];
    } else {
      const float* lr = a.logits + (int64_t)b * a.bstride + (int64_t)t * a.ld;
      float s = 0.f;
      for (int j = 0; j < a.bins; ++j) s += 1.0f / (1.0f + expf(-lr[j]));
      float v = s / a.speed;
      if (a.dur_raw) a.dur_raw[(int64_t)b * a.T + t] = v;
      if (v != v) v = 1.0f;                       // nan -> 1
      else if (v == INFINITY) v = 100.0f;         // +inf -> max_frames_per_phoneme
      else if (v == -INFINITY) v = 1.0f;
      v = fmaxf(rintf(v), 1.0f);                  // mx.round = half-to-even
      if (a.max_frames >= 0) v = fminf(v, a.max_frames ? (float)a.max_frames : 100.0f);
      d = (int)v;
    }
    a.dur[(int64_t)b * a.T + t] = d;
  } else if (t < a.T) {
    a.dur[(int64_t)b * a.T + t] = 0;
  }
  sdur[t] = d;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < a.T && i < 512; ++i) { sstart[i] = acc; acc += sdur[i]; }
    sstart[512] = acc;
    a.frames[b] = acc;
  }
  __syncthreads();
  if (t < len) {
    const int s0 = sstart[t];
    for (int f = 0; f < d; ++f)
      if (s0 + f < a.idx_ld) a.idx[(int64_t)b * a.idx_ld + s0 + f] = t;
  }
}

__global__ __launch_bounds__(256) void pool_up2_kernel(const mi355_pool_up2_args a) {
  // y[b, n, c], n in [0, 2L): position p = n + 1 of the un-trimmed (2L+1)-long transposed conv
  const int64_t row = blockIdx.x;
  const int L2 = 2 * a.L;
  const int b = (int)(row / L2), n = (int)(row - (int64_t)b * L2);
  const int len = a.lens ? a.lens[b] : a.L;
  if (n >= 2 * len) return;
  const int p = n + 1;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)n * a.ldy;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const float sc = a.scale[(int64_t)b * a.pre_ld + c], sh = a.shift[(int64_t)b * a.pre_ld + c];
    auto act = [&](int t) {
      float v = xb[(int64_t)t * a.ldx + c] * sc + sh;
      return v > 0.f ? v : v * a.slope;
    };
    float o;
    if (p & 1) {
      o = act((p - 1) >> 1) * a.w[c * 3 + 1];
    } else {
      const int t = p >> 1;
      o = 0.f;
      if (t < len) o += act(t) * a.w[c * 3 + 0];
      if (t - 1 >= 0) o += act(t - 1) * a.w[c * 3 + 2];
    }
    yr[c] = o + a.bias[c];
  }
}

__global__ void conv1d_c1_k3s2_kernel(const float* x, int ldxb, int Lin, const int32_t* lens_in, float w0, float w1, float w2,
                                      float bias, float* y, int64_t ybs, int ldy, int col, int Lout, int B) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (l >= Lout) return;
  const int len = lens_in ? lens_in[b] : Lin;
  const int lout = (len + 2 - 3) / 2 + 1;
  if (l >= lout) return;
  const float* xb = x + (int64_t)b * ldxb;
  float s = 0.f;
  const int i0 = 2 * l - 1;
  if (i0 >= 0 && i0 < len) s += w0 * xb[i0];
  if (i0 + 1 < len) s += w1 * xb[i0 + 1];
  if (i0 + 2 < len) s += w2 * xb[i0 + 2];
  y[(int64_t)b * ybs + (int64_t)l * ldy + col] = s + bias;
}

// ---- BigVGAN anti-aliased activation: one workgroup = T outputs x CT channels (T * CT = 4096); the x window and the activated 2x signal live in
// LDS, so the double-rate tensor never reaches HBM.  CT is the widest power of two (<= 64) that divides C: the late stages are narrow (24 .. 96
// channels) and a fixed 64-lane channel tile would idle up to 62 % of the lanes there.
template <int CT>
__global__ __launch_bounds__(256) void aa_act_kernel(const mi355_aa_act_args a) {
  constexpr int T = 4096 / CT, RG = 256 / CT;
  __shared__ float xs[(T + 11) * CT];       // x rows t0 - 6 .. t0 + T + 4 (edge-clamped)
  __shared__ float as_[(2 * T + 10) * CT];  // a[m], m = 2 t0 - 5 .. 2 t0 + 2 T + 4 (edge-clamped)
  __shared__ float fu[12], fd[12];
  const int b = blockIdx.z, c0 = blockIdx.y * CT, t0 = blockIdx.x * T;
  const int len = a.lens ? a.lens[b] : a.L;
  if (t0 >= len) return;
  const int cl = threadIdx.x % CT, rg = threadIdx.x / CT, c = c0 + cl;
  const bool cok = c < a.C;
  if (threadIdx.x < 12) { fu[threadIdx.x] = a.up_filter[threadIdx.x]; fd[threadIdx.x] = a.down_filter[threadIdx.x]; }
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  const int nrows = min(T, len - t0);
  for (int r = rg; r < nrows + 11; r += RG) {
    const int t = min(max(t0 - 6 + r, 0), len - 1);
    xs[r * CT + cl] = cok ? xb[(int64_t)t * a.ldx + c] : 0.f;
  }
  __syncthreads();
  const float al = cok ? a.alpha[c] : 1.f, ib = cok ? a.inv_beta[c] : 0.f;
  for (int j = rg; j < 2 * nrows + 10; j += RG) {
    const int m = min(max(2 * t0 - 5 + j, 0), 2 * len - 1);  // edge padding of the activated signal = its first / last sample
    // u[m] = 2 * sum_k f[k] * xp[(m + 15 - k) / 2] over k of the parity of m + 15; xp[i] = x[clamp(i - 5)]
    const int p = (m + 15) & 1;
    float u = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int k = p + 2 * q;
      const int xi = min(max((m + 15 - k) / 2 - 5, 0), len - 1);
      u += fu[k] * xs[(xi - (t0 - 6)) * CT + cl];
    }
    u *= 2.0f;
    const float sn = __sinf(al * u);  // hardware sine (|al * u| stays far below its 256-revolution range); ~1e-6 absolute, as in the conv prologues
    as_[j * CT + cl] = u + ib * (sn * sn);
  }
  __syncthreads();
  float* yb = a.y + (int64_t)b * a.y_bstride;
  for (int r = rg; r < nrows; r += RG) {
    if (!cok) continue;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += fd[k] * as_[(2 * r + k) * CT + cl];
    yb[(int64_t)(t0 + r) * a.ldy + c] = acc;
  }
}

template <int CT>
void launch_aa(const mi355_aa_act_args& a, hipStream_t st) {
  constexpr int T = 4096 / CT;
  hipLaunchKernelGGL(aa_act_kernel<CT>, dim3((a.L + T - 1) / T, (a.C + CT - 1) / CT, a.B), dim3(256), 0, st, a);
}

}  // namespace

extern "C" int mi355_aa_activation(const mi355_aa_act_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y && ap->up_filter && ap->down_filter && ap->alpha && ap->inv_beta, "aa_activation: null tensor");
  const mi355_aa_act_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0 && a.ldx >= a.C && a.ldy >= a.C, "aa_activation: bad shape");
  MI355_REQUIRE(a.x != a.y, "aa_activation: in-place operation is not supported (every output reads 12 neighbours)");
  MI355_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)stream;
  if (a.C % 64 == 0 || a.C > 128) launch_aa<64>(a, st);        // wide tensors: at most one partly filled channel tile
  else if (a.C % 32 == 0) launch_aa<32>(a, st);
  else if (a.C % 16 == 0) launch_aa<16>(a, st);
  else if (a.C % 8 == 0 || a.C < 8) launch_aa<8>(a, st);
  else launch_aa<64>(a, st);
  MI355_LAUNCH_CHECK("aa_activation");
  return MI355_OK;
}

namespace {

// ---- polyphase FIR sample-rate conversion: one thread = one output sample, one workgroup = 256 consecutive outputs of one row.  Output n sits at
// t = (n + first) * down on the up-sampled grid: only the taps of phase t % up meet a non-zero sample, so it is a K-term dot product of the taps of
// phase t % up with the inputs t / up, t / up - 1, ... (edge-clamped: resample_poly's padtype="edge").  Products and the running sum are float64, like
// scipy's upfirdn on float64 taps (the float32 results agree to the last bit up to summation-order ties), which makes the float64 FMA pipe the bound:
// K multiply-adds per 4-byte output.  The input window of the workgroup is converted to float64 ONCE while it is staged in LDS (every input is read
// by ~K * up / down threads), and the table is tap-major [K][up], so the 64 phases a wave touches for one k lie in at most ceil(8 * up / 128)
// cache lines (10 for 44.1 -> 16 kHz) instead of one line per lane.
__global__ __launch_bounds__(256) void resample_poly_kernel(const mi355_resample_args a, const int span) {
  extern __shared__ double xw[];
  const int row = blockIdx.y, n0 = blockIdx.x * 256;
  const float* x = a.x + (int64_t)row * a.x_bstride;
  const int64_t q_first = ((int64_t)(n0 + a.first) * a.down) / a.up - (a.K - 1);  // lowest input index any output of this workgroup reads
  for (int i = threadIdx.x; i < span; i += 256) {
    const int64_t src = min(max(q_first + i, (int64_t)0), (int64_t)a.n_in - 1);
    xw[i] = (double)x[src];
  }
  __syncthreads();
  const int n = n0 + threadIdx.x;
  if (n >= a.n_out) return;
  const int64_t t = (int64_t)(n + a.first) * a.down;
  const double* h = a.table + (int)(t % a.up);
  const double* xp = xw + (int)(t / a.up - q_first);  // input t / up inside the window; tap k reads xp[-k]
  double acc = 0.0;
  for (int k = 0; k < a.K; ++k) acc += h[(int64_t)k * a.up] * xp[-k];
  a.y[(int64_t)row * a.y_bstride + n] = (float)acc;
}

// The same conversion with R = 4 outputs per thread that share a phase (n, n + up, n + 2 up, n + 3 up: their taps are the same table column and
// their inputs lie `down` apart), which is what the one-output kernel above is bound by: it issues an 8-byte table read and an 8-byte LDS read per
// float64 multiply-add (2.3 - 4.8 TFLOP/s measured, call 47).  Here a table value feeds four multiply-adds and the window sits in LDS as float32
// (half the LDS bytes; the conversion is a second float64-rate instruction).  Thread u of the launch owns outputs (u / up) * 4 up + u % up + j up:
// consecutive threads still write consecutive samples for each j.  A workgroup's 256 threads touch at most 256 / up + 2 groups of 4 up outputs.
constexpr int RS_R = 4;

__device__ __forceinline__ int64_t rs_base(int64_t u, int up) { return (u / up) * (RS_R * (int64_t)up) + u % up; }

__global__ __launch_bounds__(256) void resample_poly4_kernel(const mi355_resample_args a) {
  extern __shared__ float xf[];
  const int row = blockIdx.y;
  const int64_t u0 = (int64_t)blockIdx.x * 256;
  const float* x = a.x + (int64_t)row * a.x_bstride;
  const int64_t n_min = rs_base(u0, a.up), n_max = rs_base(u0 + 255, a.up) + (RS_R - 1) * (int64_t)a.up;
  const int64_t q_lo = ((n_min + a.first) * a.down) / a.up - (a.K - 1);
  const int len = (int)(((n_max + a.first) * a.down) / a.up - q_lo + 1);
  for (int i = threadIdx.x; i < len; i += 256) xf[i] = x[min(max(q_lo + i, (int64_t)0), (int64_t)a.n_in - 1)];
  __syncthreads();
  const int64_t nb = rs_base(u0 + threadIdx.x, a.up);
  if (nb >= a.n_out) return;
  const int64_t t = (nb + a.first) * a.down;
  const double* h = a.table + (int)(t % a.up);
  const float* xp = xf + (int)(t / a.up - q_lo);   // output j, tap k reads xp[j * down - k]
  double acc[RS_R] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < a.K; ++k) {
    const double hk = h[(int64_t)k * a.up];
#pragma unroll
    for (int j = 0; j < RS_R; ++j) acc[j] += hk * (double)xp[j * a.down - k];
  }
  float* y = a.y + (int64_t)row * a.y_bstride;
#pragma unroll
  for (int j = 0; j < RS_R; ++j)
    if (nb + j * (int64_t)a.up < a.n_out) y[nb + j * (int64_t)a.up] = (float)acc[j];
}

}  // namespace

extern "C" int mi355_resample_poly(const mi355_resample_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y && ap->table, "resample_poly: null tensor");
  const mi355_resample_args a = *ap;
  MI355_REQUIRE(a.rows > 0 && a.n_in > 0 && a.n_out > 0 && a.up > 0 && a.down > 0 && a.K > 0 && a.first >= 0, "resample_poly: bad shape");
  MI355_REQUIRE(a.rows <= 65535, "resample_poly: at most 65535 rows per call");
  hipStream_t st = (hipStream_t)stream;
  // four-outputs-per-thread kernel: the 256 threads of a workgroup span at most (255 / up + 1) * 4 up + up - 1 + 3 up outputs (see rs_base)
  const int64_t dn = (255 / (int64_t)a.up + 1) * RS_R * a.up + a.up - 1 + (RS_R - 1) * (int64_t)a.up;
  const int64_t span4 = (dn * a.down) / a.up + 2 + (a.K - 1);
  if (span4 * 4 <= 64 * 1024) {
    MI355_CLEAR_ERROR();
    const int64_t threads = ((a.n_out + RS_R * (int64_t)a.up - 1) / (RS_R * (int64_t)a.up)) * a.up;
    hipLaunchKernelGGL(resample_poly4_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)a.rows), dim3(256), (size_t)span4 * 4, st, a);
  } else {
    // one output per thread: inputs from (n0 + first) * down / up - (K - 1) to (n0 + 255 + first) * down / up
    const int64_t span = (255 * (int64_t)a.down) / a.up + 2 + (a.K - 1);
    MI355_REQUIRE(span * 8 <= 64 * 1024, "resample_poly: the input window of 256 outputs (%lld float64 samples) exceeds the 64 KB LDS budget",
                  (long long)span);
    MI355_CLEAR_ERROR();
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((a.n_out + 255) / 256), (unsigned)a.rows), dim3(256), (size_t)span * 8, st, a, (int)span);
  }
  MI355_LAUNCH_CHECK("resample_poly");
  return MI355_OK;
}

extern "C" int mi355_gather_rows(const mi355_gather_rows_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->table && ap->idx && ap->y, "gather_rows: null tensor");
  const mi355_gather_rows_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0, "gather_rows: bad shape");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((int64_t)a.B * a.L)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("gather_rows");
  return MI355_OK;
}

namespace {
// one thread = 8 consecutive d of one (g, b, h, t): two float4 in, one 16-byte store out; a block = 32 rows t of one (g, b, h) when dh = 64
__global__ __launch_bounds__(256) void kv_head_major16_kernel(const float* __restrict__ kv, const int64_t bstride, const int ld, const int B, const int T, const int G, const int H,
                                                               const int dh, uint4* __restrict__ out, const int bf16) {
  const int cpr = dh >> 3;                                    // 16-byte pieces per (t, head)
  const int64_t per_head = (int64_t)T * cpr;
  const int64_t total = (int64_t)G * B * H * per_head;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t blk = i / per_head;                         // (g * B + b) * H + h
    const int r = (int)(i - blk * per_head);
    const int t = r / cpr, c = r - t * cpr;
    const int h = (int)(blk % H);
    const int64_t gb = blk / H;
    const int b = (int)(gb % B), g = (int)(gb / B);
    const float* src = kv + (int64_t)b * bstride + (int64_t)t * ld + (g * H + h) * dh + c * 8;
    const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
    uint4 o;
    if (bf16) {
      o.x = pack_bf16x2(v0.x, v0.y); o.y = pack_bf16x2(v0.z, v0.w); o.z = pack_bf16x2(v1.x, v1.y); o.w = pack_bf16x2(v1.z, v1.w);
    } else {
      o.x = pack_f16x2(v0.x, v0.y); o.y = pack_f16x2(v0.z, v0.w); o.z = pack_f16x2(v1.x, v1.y); o.w = pack_f16x2(v1.z, v1.w);
    }
    out[i] = o;
  }
}
}  // namespace

extern "C" int mi355_kv_head_major16(const float* kv, int64_t kv_bstride, int32_t ld, int32_t B, int32_t T, int32_t G, int32_t H, int32_t dh, void* out,
                                     int32_t out_dtype, void* stream) {
  MI355_REQUIRE(kv && out, "kv_head_major16: null tensor");
  MI355_REQUIRE(B > 0 && T > 0 && G > 0 && H > 0 && dh > 0 && dh % 8 == 0, "kv_head_major16: bad shape (dh must be a multiple of 8)");
  MI355_REQUIRE(out_dtype == MI355_KV_BF16 || out_dtype == MI355_KV_F16, "kv_head_major16: out_dtype must be MI355_KV_BF16 or MI355_KV_F16");
  MI355_REQUIRE(ld >= G * H * dh && ld % 4 == 0 && kv_bstride % 4 == 0 && ((uintptr_t)kv) % 16 == 0 && ((uintptr_t)out) % 16 == 0,
                "kv_head_major16: ld must cover G * H * dh and be a multiple of 4; 16-byte aligned pointers");
  const int64_t total = (int64_t)G * B * H * T * (dh / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(kv_head_major16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, kv, kv_bstride, (int)ld, (int)B, (int)T, (int)G, (int)H, (int)dh, (uint4*)out,
                     out_dtype == MI355_KV_BF16 ? 1 : 0);
  MI355_LAUNCH_CHECK("kv_head_major16");
  return MI355_OK;
}

extern "C" int mi355_broadcast_rows(const float* v, int32_t ldv, int32_t C, float* y, int64_t y_bstride, int32_t ldy,
                                    int32_t L, const int32_t* lens, int32_t B, void* stream) {
  MI355_REQUIRE(v && y && B > 0 && L > 0 && C > 0, "broadcast_rows: bad arguments");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((int64_t)B * L)), dim3(256), 0, (hipStream_t)stream, v, ldv, C,
                     y, y_bstride, ldy, L, lens, B);
  MI355_LAUNCH_CHECK("broadcast_rows");
  return MI355_OK;
}

extern "C" int mi355_duration_align(const mi355_duration_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->dur && ap->frames && ap->idx, "duration_align: null tensor");
  const mi355_duration_args a = *ap;
  MI355_REQUIRE(a.forced_dur || a.logits, "duration_align: need logits or forced durations");
  MI355_REQUIRE(a.T > 0 && a.T <= 512, "duration_align: T must be in [1, 512] (got %d)", a.T);
  MI355_REQUIRE(a.speed > 0.f || a.forced_dur, "duration_align: speed must be positive");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(duration_align_kernel, dim3(a.B), dim3(512), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("duration_align");
  return MI355_OK;
}

extern "C" int mi355_adain_pool_up2(const mi355_pool_up2_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y && ap->scale && ap->shift && ap->w && ap->bias, "adain_pool_up2: null tensor");
  const mi355_pool_up2_args a = *ap;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(pool_up2_kernel, dim3((unsigned)((int64_t)a.B * 2 * a.L)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("adain_pool_up2");
  return MI355_OK;
}

extern "C" int mi355_conv1d_c1_k3s2(const float* x, int32_t ldx_b, int32_t Lin, const int32_t* lens_in, float w0, float w1,
                                    float w2, float bias, float* y, int64_t y_bstride, int32_t ldy, int32_t col, int32_t Lout,
                                    int32_t B, void* stream) {
  MI355_REQUIRE(x && y && B > 0 && Lout > 0, "conv1d_c1_k3s2: bad arguments");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(conv1d_c1_k3s2_kernel, dim3((Lout + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, ldx_b, Lin,
                     lens_in, w0, w1, w2, bias, y, y_bstride, ldy, col, Lout, B);
  MI355_LAUNCH_CHECK("conv1d_c1_k3s2");
  return MI355_OK;
}
