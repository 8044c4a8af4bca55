// ECAPA-TDNN speaker encoder of Qwen3-TTS (tts/models/qwen3_tts/speaker_encoder.py): the three row-wise pieces that sit BETWEEN its convolutions
// (the convolutions themselves are mi355_conv_gemm launches: reflect-padded TDNN k 5 / 3 / 1 with the ReLU in the epilogue).
//   mi355_ecapa_rows        y[b, r] = f(x[b, reflect(r - pad)]) * sigmoid(gate[b]) + res[b, reflect(r - pad)]
//                           = reflect_pad_1d (:11-26), the "chunk + previous output" input of the Res2Net blocks (:98-101), the tanh in front of the
//                           attention conv (:213), and SqueezeExcitationBlock's x * se (:141) joined with the block residual (:180) -- one pass each
//   mi355_time_moments      per (utterance, channel) mean and sqrt(biased variance + eps) over time: the squeeze of the SE block (:133) and the
//                           global context AttentiveStatisticsPooling concatenates to every frame (:201-202)
//   mi355_attentive_pool    softmax over TIME per channel, weighted mean, sqrt(clip(weighted variance, eps)) (:219-228): three sweeps of one
//                           [T, 64-channel] column block per workgroup, nothing but the [B, 2C] result is written
// The speaker encoder runs once per reference clip (T = a few hundred mel frames, C <= 1536): these kernels are HBM-trivial; they exist so that the
// clip's x-vector never leaves the device and no step of the path falls back to a host loop.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ecapa_rows_kernel(const mi355_ecapa_rows_args a) {
  const int b = blockIdx.y;
  const int rows_out = a.T + 2 * a.pad;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  const float* rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
  const float* gb = a.gate ? a.gate + (int64_t)b * a.gate_ld : nullptr;
  float* yb = a.y + (int64_t)b * a.y_bstride;
  for (int r = blockIdx.x; r < rows_out; r += gridDim.x) {
    int s = r - a.pad;
    if (s < 0) s = -s;                          // mirror without repeating the edge sample
    if (s >= a.T) s = 2 * (a.T - 1) - s;
    for (int c = threadIdx.x; c < a.C; c += 256) {
      float v = xb[(int64_t)s * a.ldx + c];
      if (a.pre_tanh) v = tanhf(v);
      if (gb) v *= 1.f / (1.f + expf(-gb[c]));
      if (rb) v += rb[(int64_t)s * a.ldr + c];
      yb[(int64_t)r * a.ldy + c] = v;
    }
  }
}

// 256 threads = 4 time phases x 64 channels; a workgroup owns 64 channels of one utterance.
__device__ __forceinline__ float sum4(float v, float* red, int phase, int cl) {
  __syncthreads();                 // the previous use of red[] is over
  red[phase * 64 + cl] = v;
  __syncthreads();
  return red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
}
__device__ __forceinline__ float max4(float v, float* red, int phase, int cl) {
  __syncthreads();
  red[phase * 64 + cl] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[cl], red[64 + cl]), fmaxf(red[128 + cl], red[192 + cl]));
}

__global__ __launch_bounds__(256) void time_moments_kernel(const mi355_time_moments_args a) {
  __shared__ float red[256];
  const int b = blockIdx.y, cl = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool live = c < a.C;
  const float* xb = a.x + (int64_t)b * a.x_bstride + (live ? c : 0);
  float s = 0.f;
  if (live)
    for (int t = phase; t < a.T; t += 4) s += xb[(int64_t)t * a.ldx];
  const float mean = sum4(s, red, phase, cl) / (float)a.T;
  float q = 0.f;
  if (live)
    for (int t = phase; t < a.T; t += 4) {
      const float d = xb[(int64_t)t * a.ldx] - mean;
      q = fmaf(d, d, q);
    }
  const float var = sum4(q, red, phase, cl) / (float)a.T;
  if (live && phase == 0) {
    a.mean[(int64_t)b * a.out_ld + c] = mean;
    if (a.std) a.std[(int64_t)b * a.out_ld + c] = sqrtf(var + a.eps);
  }
}

__global__ __launch_bounds__(256) void attentive_pool_kernel(const mi355_attentive_pool_args a) {
  __shared__ float red[256];
  const int b = blockIdx.y, cl = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool live = c < a.C;
  const float* xb = a.x + (int64_t)b * a.x_bstride + (live ? c : 0);
  const float* lb = a.logits + (int64_t)b * a.l_bstride + (live ? c : 0);
  float m = -INFINITY;
  if (live)
    for (int t = phase; t < a.T; t += 4) m = fmaxf(m, lb[(int64_t)t * a.ldl]);
  m = max4(m, red, phase, cl);
  float z = 0.f, s1 = 0.f;
  if (live)
    for (int t = phase; t < a.T; t += 4) {
      const float p = expf(lb[(int64_t)t * a.ldl] - m);
      z += p;
      s1 = fmaf(p, xb[(int64_t)t * a.ldx], s1);
    }
  z = sum4(z, red, phase, cl);
  const float mean = sum4(s1, red, phase, cl) / z;
  float q = 0.f;
  if (live)
    for (int t = phase; t < a.T; t += 4) {
      const float p = expf(lb[(int64_t)t * a.ldl] - m);
      const float d = xb[(int64_t)t * a.ldx] - mean;
      q = fmaf(p * d, d, q);
    }
  const float var = sum4(q, red, phase, cl) / z;
  if (live && phase == 0) {
    a.out[(int64_t)b * a.out_ld + c] = mean;
    a.out[(int64_t)b * a.out_ld + a.C + c] = sqrtf(fmaxf(var, a.eps));
  }
}

}  // namespace

extern "C" int mi355_ecapa_rows(const mi355_ecapa_rows_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y, "ecapa_rows: null tensor");
  const mi355_ecapa_rows_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.T > 0 && a.C > 0 && a.pad >= 0 && a.ldx >= a.C && a.ldy >= a.C, "ecapa_rows: bad shape");
  MI355_REQUIRE(a.pad < a.T, "ecapa_rows: reflect padding of %d rows needs more than %d input rows", a.pad, a.pad);
  MI355_REQUIRE(!a.res || a.ldr >= a.C, "ecapa_rows: bad residual row stride");
  MI355_REQUIRE(!a.gate || a.gate_ld >= a.C, "ecapa_rows: bad gate row stride");
  MI355_REQUIRE(a.B <= 65535, "ecapa_rows: at most 65535 utterances per launch");
  const int rows_out = a.T + 2 * a.pad;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(ecapa_rows_kernel, dim3((unsigned)(rows_out < 4096 ? rows_out : 4096), (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("ecapa_rows");
  return MI355_OK;
}

extern "C" int mi355_time_moments(const mi355_time_moments_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->mean, "time_moments: null tensor");
  const mi355_time_moments_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.B <= 65535 && a.T > 0 && a.C > 0 && a.ldx >= a.C && a.out_ld >= a.C, "time_moments: bad shape");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(time_moments_kernel, dim3((unsigned)((a.C + 63) / 64), (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("time_moments");
  return MI355_OK;
}

extern "C" int mi355_attentive_pool(const mi355_attentive_pool_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->logits && ap->out, "attentive_pool: null tensor");
  const mi355_attentive_pool_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.B <= 65535 && a.T > 0 && a.C > 0 && a.ldx >= a.C && a.ldl >= a.C && a.out_ld >= 2 * a.C, "attentive_pool: bad shape");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(attentive_pool_kernel, dim3((unsigned)((a.C + 63) / 64), (unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("attentive_pool");
  return MI355_OK;
}
