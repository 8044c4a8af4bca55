// Residual vector quantisation, ENCODE side (gfx950): nearest-codeword search layer by layer with the residual kept on chip.
// Replaces ResidualVectorQuantization.encode (codec/models/mimi/modules/quantization.py:84-96) over EuclideanCodebook.encode (:37-45):
//   per layer  idx = argmin_b (|e_b|^2 / 2 - x . e_b)   (the first minimum, like mx.argmin),   x <- x - e_idx   in float32
// -- the encode half of Mimi (CSM's audio context, sesame.py:527-559) and of the Qwen3-TTS speech tokenizer (speech_tokenizer.py:1037-1058), which
// the reference runs as nq x (matmul + argmin + take) MLX ops per call.  One workgroup per frame: the residual lives in LDS for all layers; the
// codebooks are read TRANSPOSED ([layer][d][bin]: the 64 lanes of a wave read 64 consecutive bins of one d -- coalesced) and stay in L2 across
// frames.  Also returns the gap between the best and the second-best score of every decision: a caller (and the tests' margin rule) can see
// which codes sit on a knife edge of float32 rounding.
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

}  // namespace

extern "C" int mi355_rvq_encode(const mi355_rvq_encode_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->tables && ap->tables_t && ap->c2 && ap->codes, "rvq_encode: null tensor");
  const mi355_rvq_encode_args a = *ap;
  MI355_REQUIRE(a.rows > 0 && a.D > 0 && a.D <= 4096 && a.bins > 1 && a.n_layers > 0 && a.ldx >= a.D && a.ld_codes >= a.n_layers, "rvq_encode: bad shape");
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)a.rows), dim3(256), (size_t)(a.D + 12) * sizeof(float), (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("rvq_encode");
  return MI355_OK;
}
