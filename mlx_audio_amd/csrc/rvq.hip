// Residual vector quantisation, ENCODE side (gfx950): nearest-codeword search layer by layer with the residual kept on chip.
// Replaces ResidualVectorQuantization.encode (codec/models/mimi/modules/quantization.py:84-96) over EuclideanCodebook.encode (:37-45):
//   per layer  idx = argmin_b (|e_b|^2 / 2 - x . e_b)   (the first minimum, like mx.argmin),   x <- x - e_idx   in float32
// -- the encode half of Mimi (CSM's audio context, sesame.py:527-559) and of the Qwen3-TTS speech tokenizer (speech_tokenizer.py:1037-1058), which
// the reference runs as nq x (matmul + argmin + take) MLX ops per call.  One workgroup per frame: the residual lives in LDS for all layers; the
// codebooks are read TRANSPOSED ([layer][d][bin]: the 64 lanes of a wave read 64 consecutive bins of one d -- coalesced) and stay in L2 across
// frames.  Also returns the gap between the best and the second-best score of every decision: a caller (and the tests' margin rule) can see
// which codes sit on a knife edge of float32 rounding.
#include <stdlib.h>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void rvq_encode_kernel(const mi355_rvq_encode_args a) {
  extern __shared__ float sm[];          // residual [D], then the reduction scratch
  float* r = sm;
  float* red = sm + a.D;                 // [4 waves][3]: best score, best index (as float bits), second-best score
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  for (int d = tid; d < a.D; d += 256) r[d] = a.x[row * a.ldx + d];
  __syncthreads();
  for (int l = 0; l < a.n_layers; ++l) {
    const float* et = a.tables_t + (int64_t)l * a.D * a.bins;
    const float* c2 = a.c2 + (int64_t)l * a.bins;
    float best = INFINITY, second = INFINITY;
    int bi = 0x7fffffff;
    for (int b = tid; b < a.bins; b += 256) {
      float dot = 0.f;
      for (int d = 0; d < a.D; ++d) dot = fmaf(r[d], et[(int64_t)d * a.bins + b], dot);
      const float s = c2[b] - dot;
      if (s < best) { second = best; best = s; bi = b; }       // bins ascend per thread: the first minimum wins ties
      else if (s < second) second = s;
    }
    // wave reduction of (best, index, second): ties go to the lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const bool take = ob < best || (ob == best && oi < bi);
      const float loser = take ? best : ob;
      if (take) { best = ob; bi = oi; }
      second = fminf(fminf(second, os), loser);
    }
    if (lane == 0) { red[3 * wave] = best; red[3 * wave + 1] = __int_as_float(bi); red[3 * wave + 2] = second; }
    __syncthreads();
    best = red[0]; bi = __float_as_int(red[1]); second = red[2];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ob = red[3 * w], os = red[3 * w + 2];
      const int oi = __float_as_int(red[3 * w + 1]);
      const bool take = ob < best || (ob == best && oi < bi);
      const float loser = take ? best : ob;
      if (take) { best = ob; bi = oi; }
      second = fminf(fminf(second, os), loser);
    }
    if (tid == 0) {
      a.codes[row * a.ld_codes + l] = bi;
      if (a.margins) a.margins[row * a.ld_codes + l] = second - best;
    }
    const float* e = a.tables + ((int64_t)l * a.bins + bi) * a.D;
    __syncthreads();                     // every thread has read red[]
    for (int d = tid; d < a.D; d += 256) r[d] = r[d] - e[d];
    __syncthreads();
  }
}

// Several frames per workgroup (round 5): with one frame per workgroup every workgroup re-reads the whole table of every layer from L2 (EnCodec: 512 KB per
// layer and frame -- 7.7 TB/s of L2 traffic for 12 000 frames x 8 layers, 6.4 ms); here R frames share each table read.  A thread still owns bins
// tid, tid + 256, ... and walks d in ascending order with one fmaf per (frame, d) -- the same arithmetic, so codes and margins are bit-identical to
// rvq_encode_kernel's; the residuals of the R frames sit in LDS ([R][D], read as broadcast float4), the per-frame (best, index, second) reductions are
// the one-frame kernel's, R times.
template <int R>
__global__ __launch_bounds__(256) void rvq_encode_rows_kernel(const mi355_rvq_encode_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // residuals [R][D], then red [R][4 waves][3], then winners [R]
  const int D = a.D;
  float* r = sm;
  float* red = sm + R * D;
  int* win = (int*)(red + R * 12);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const int nr = (int)min((int64_t)R, (int64_t)a.rows - row0);
  for (int i = tid; i < R * D; i += 256) {
    const int rr = i / D, d = i - rr * D;
    r[i] = rr < nr ? a.x[(row0 + rr) * a.ldx + d] : 0.f;
  }
  __syncthreads();
  for (int l = 0; l < a.n_layers; ++l) {
    const float* et = a.tables_t + (int64_t)l * D * a.bins;
    const float* c2 = a.c2 + (int64_t)l * a.bins;
    float best[R], second[R];
    int bi[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) { best[rr] = INFINITY; second[rr] = INFINITY; bi[rr] = 0x7fffffff; }
    for (int b = tid; b < a.bins; b += 256) {
      float dot[R];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) dot[rr] = 0.f;
      for (int d = 0; d < D; d += 4) {
        const float e0 = et[(int64_t)d * a.bins + b], e1 = et[(int64_t)(d + 1) * a.bins + b], e2 = et[(int64_t)(d + 2) * a.bins + b],
                    e3 = et[(int64_t)(d + 3) * a.bins + b];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          const float4 rv = *(const float4*)(r + rr * D + d);
          dot[rr] = fmaf(rv.x, e0, dot[rr]);
          dot[rr] = fmaf(rv.y, e1, dot[rr]);
          dot[rr] = fmaf(rv.z, e2, dot[rr]);
          dot[rr] = fmaf(rv.w, e3, dot[rr]);
        }
      }
      const float cb = c2[b];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const float s = cb - dot[rr];
        if (s < best[rr]) { second[rr] = best[rr]; best[rr] = s; bi[rr] = b; }
        else if (s < second[rr]) second[rr] = s;
      }
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      float bb = best[rr], ss = second[rr];
      int ii = bi[rr];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(bb, o, 64), os = __shfl_xor(ss, o, 64);
        const int oi = __shfl_xor(ii, o, 64);
        const bool take = ob < bb || (ob == bb && oi < ii);
        const float loser = take ? bb : ob;
        if (take) { bb = ob; ii = oi; }
        ss = fminf(fminf(ss, os), loser);
      }
      if (lane == 0) { red[(rr * 4 + wave) * 3] = bb; red[(rr * 4 + wave) * 3 + 1] = __int_as_float(ii); red[(rr * 4 + wave) * 3 + 2] = ss; }
    }
    __syncthreads();
    if (tid < R) {
      const int rr = tid;
      float bb = red[rr * 12], ss = red[rr * 12 + 2];
      int ii = __float_as_int(red[rr * 12 + 1]);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float ob = red[(rr * 4 + w) * 3], os = red[(rr * 4 + w) * 3 + 2];
        const int oi = __float_as_int(red[(rr * 4 + w) * 3 + 1]);
        const bool take = ob < bb || (ob == bb && oi < ii);
        const float loser = take ? bb : ob;
        if (take) { bb = ob; ii = oi; }
        ss = fminf(fminf(ss, os), loser);
      }
      win[rr] = ii;
      if (rr < nr) {
        a.codes[(row0 + rr) * a.ld_codes + l] = ii;
        if (a.margins) a.margins[(row0 + rr) * a.ld_codes + l] = ss - bb;
      }
    }
    __syncthreads();
    const float* tab = a.tables + (int64_t)l * a.bins * D;
    for (int i = tid; i < R * D; i += 256) {
      const int rr = i / D, d = i - rr * D;
      r[i] = r[i] - tab[(int64_t)win[rr] * D + d];
    }
    __syncthreads();
  }
}

template <int R>
void launch_rvq_rows(const mi355_rvq_encode_args& a, hipStream_t st) {
  const size_t lds = ((size_t)R * a.D + R * 12 + R) * sizeof(float);
  hipLaunchKernelGGL(rvq_encode_rows_kernel<R>, dim3((unsigned)((a.rows + R - 1) / R)), dim3(256), lds, st, a);
}

}  // namespace

extern "C" int mi355_rvq_encode(const mi355_rvq_encode_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->tables && ap->tables_t && ap->c2 && ap->codes, "rvq_encode: null tensor");
  const mi355_rvq_encode_args a = *ap;
  MI355_REQUIRE(a.rows > 0 && a.D > 0 && a.D <= 4096 && a.bins > 1 && a.n_layers > 0 && a.ldx >= a.D && a.ld_codes >= a.n_layers, "rvq_encode: bad shape");
  MI355_CLEAR_ERROR();
  // several frames per workgroup once there are enough frames to keep every CU busy with them (A/B knob MI355_RVQ_ROWS = 1 / 2 / 4 / 8 forces a form)
  static const int rows_env = getenv("MI355_RVQ_ROWS") ? atoi(getenv("MI355_RVQ_ROWS")) : 0;
  const bool can = a.D % 4 == 0 && a.D <= 512 && a.ldx % 4 == 0 && ((uintptr_t)a.x) % 16 == 0;   // (float4 rows of the LDS residuals; x itself is read by scalars)
  const int R = !can ? 1 : (rows_env > 0 ? rows_env : (a.rows >= 4096 ? 8 : (a.rows >= 2048 ? 4 : (a.rows >= 1024 ? 2 : 1))));
  hipStream_t st = (hipStream_t)stream;
  if (R >= 8) launch_rvq_rows<8>(a, st);
  else if (R >= 4) launch_rvq_rows<4>(a, st);
  else if (R >= 2) launch_rvq_rows<2>(a, st);
  else hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)a.rows), dim3(256), (size_t)(a.D + 12) * sizeof(float), st, a);
  MI355_LAUNCH_CHECK("rvq_encode");
  return MI355_OK;
}
