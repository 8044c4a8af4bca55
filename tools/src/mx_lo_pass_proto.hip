// Prototype of the split "fp16 hi pass + block-scaled e4m3 lo pass" on real hardware (tools, not product; DESIGN section 8, gap 1).
//   Y[M, N] = X[M, K] (float32 activations) . W[N, K]^T (bf16-valued weights)
// as   sum_k fp16(x) fp16(w)                                   4 x v_mfma_f32_32x32x16_f16 per 64 K
//    + sum_k q8(x - fp16(x)) q8(w)                             1 x v_mfma_scale_f32_32x32x64_f8f6f4 per 64 K, into the SAME accumulator
// where q8 = OCP MX: e4m3 elements with one E8M0 scale per 32 consecutive K (shared exponent = floor(log2(amax)) - 8, elements saturate at 448).
// The activation side is converted ON THE DEVICE the way a conv producer wave would do it (fp16 rounding, residual, block maximum across the two lanes
// that share a K block, scale byte, v_cvt_pk_fp8_f32); the weight side is packed on the host (a second weight image next to the 16-bit one).
// Operand layout of the scaled instruction as measured by tools/src/mfma_mx_probe.hip: byte j of lane l holds K = 32 (j >> 4) + 16 (l >> 5) + (j & 15), the
// scale of K block b of row i is byte 0 of lane i + 32 b's scale register.
// One wave per 32 x 32 output tile, nothing tuned: this binary answers "is the arithmetic what tools/study_split_formats.py assumed?" -- it prints the error of
// the device result against the exact float64 product, against the same scheme evaluated on the host, and what today's bf16 hi + lo would give.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(64) void proto(const float* __restrict__ x, const _Float16* __restrict__ w16, const uint8_t* __restrict__ w8, const int* __restrict__ wsc,
                                            float* __restrict__ y, int M, int N, int K, int use_lo) {
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const float* xr = x + (size_t)(m0 + i) * K;
  const _Float16* wr = w16 + (size_t)(n0 + i) * K;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 64) {
    // ---- hi pass: lane (i, g) holds K = k0 + 16 s + 8 g + 0..7 for sub-step s
    for (int s = 0; s < 4; ++s) {
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)xr[k0 + 16 * s + 8 * g + e];
        b[e] = wr[k0 + 16 * s + 8 * g + e];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if (!use_lo) continue;
    // ---- lo pass: this lane's 32 residuals in the instruction's K order, the two block maxima shared with the lane 32 away
    float r[32];
    float amax[2] = {0.f, 0.f};
    for (int j = 0; j < 32; ++j) {
      const float v = xr[k0 + 32 * (j >> 4) + 16 * g + (j & 15)];
      r[j] = v - (float)(_Float16)v;
      amax[j >> 4] = fmaxf(amax[j >> 4], fabsf(r[j]));
    }
    int sb[2];
    for (int b = 0; b < 2; ++b) {
      amax[b] = fmaxf(amax[b], __shfl_xor(amax[b], 32, 64));
      const int ef = (int)((__float_as_uint(amax[b]) >> 23) & 255u);       // biased exponent of the block maximum (0 for zero / subnormal: scale 2^-127)
      sb[b] = ef > 8 ? ef - 8 : 0;
    }
    i32x8 a8;
    for (int q = 0; q < 8; ++q) {
      const float mul = __uint_as_float((uint32_t)(254 - sb[q >> 2]) << 23);   // 2^(127 - sb): the inverse of the block scale
      float v[4];
      for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(r[4 * q + e] * mul, -448.f), 448.f);
      int pk = 0;
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
      a8[q] = pk;
    }
    const int scale_a = sb[g];                                                // lane i + 32 b carries block b's scale
    const uint8_t* wp = w8 + ((size_t)(n0 / 32) * (K / 64) + k0 / 64) * 2048 + l * 32;
    i32x8 b8;
    for (int q = 0; q < 8; ++q) b8[q] = ((const int*)wp)[q];
    const int scale_b = wsc[((size_t)(n0 / 32) * (K / 64) + k0 / 64) * 64 + l];
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 0, 0, 0, scale_a, 0, scale_b);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
    y[(size_t)(m0 + row) * N + n0 + i] = acc[r];
  }
}

static float e4m3_dec(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const float x = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
  return s ? -x : x;
}
static uint8_t e4m3_enc(float v) {   // round to nearest, ties to the even code, saturating at 448
  v = fminf(fmaxf(v, -448.f), 448.f);
  int best = 0;
  float bd = INFINITY;
  for (int c = 0; c < 256; ++c) {
    if ((c & 0x7f) == 0x7f) continue;   // NaN
    const float d = fabsf(e4m3_dec((uint8_t)c) - v);
    if (d < bd || (d == bd && !(c & 1) && (best & 1))) { bd = d; best = c; }
  }
  if (v == 0.f) best = 0;
  return (uint8_t)best;
}
static float bf16r(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}
static int block_scale_byte(const float* v, int n) {
  float amax = 0.f;
  for (int e = 0; e < n; ++e) amax = fmaxf(amax, fabsf(v[e]));
  uint32_t u;
  memcpy(&u, &amax, 4);
  const int ef = (int)((u >> 23) & 255u);
  return ef > 8 ? ef - 8 : 0;
}

int main(int argc, char** argv) {
  const int M = 128, N = 64, K = argc > 1 ? atoi(argv[1]) : 768;
  std::vector<float> X((size_t)M * K), W((size_t)N * K);
  uint32_t s = 4242u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : X) { const float a = rnd(), b = rnd(); v = a * (1.0f + 3.0f * b * b); }          // activations: a few units, heavier tails
  for (auto& v : W) v = bf16r(rnd() * 0.08f);                                                          // bf16-valued weights of conv scale
  std::vector<_Float16> W16((size_t)N * K);
  for (size_t e = 0; e < W.size(); ++e) W16[e] = (_Float16)W[e];
  // second weight image: per (32 output channels, 64 K) 64 lanes x 32 bytes in the instruction's order + one scale word per lane
  std::vector<uint8_t> W8((size_t)(N / 32) * (K / 64) * 2048);
  std::vector<int> WS((size_t)(N / 32) * (K / 64) * 64);
  std::vector<float> Wq((size_t)N * K);                                                                 // what the lo pass multiplies by, for the host evaluation
  for (int nb = 0; nb < N / 32; ++nb)
    for (int kb = 0; kb < K / 64; ++kb)
      for (int i = 0; i < 32; ++i) {
        const float* wr = &W[(size_t)(nb * 32 + i) * K + kb * 64];
        for (int b = 0; b < 2; ++b) {
          const int sbyte = block_scale_byte(wr + 32 * b, 32);
          const float sc = ldexpf(1.f, sbyte - 127);
          WS[((size_t)nb * (K / 64) + kb) * 64 + i + 32 * b] = sbyte;
          for (int kk = 0; kk < 32; ++kk) {
            const uint8_t c = e4m3_enc(wr[32 * b + kk] / sc);
            Wq[(size_t)(nb * 32 + i) * K + kb * 64 + 32 * b + kk] = e4m3_dec(c) * sc;
            const int g = kk >> 4, j = 16 * b + (kk & 15);                                              // K = 32 b + 16 g + (j & 15)
            W8[((size_t)nb * (K / 64) + kb) * 2048 + (size_t)(i + 32 * g) * 32 + j] = c;
          }
        }
      }
  float *dx, *dy;
  _Float16* dw16;
  uint8_t* dw8;
  int* dws;
  (void)hipMalloc(&dx, X.size() * 4); (void)hipMalloc(&dy, (size_t)M * N * 4); (void)hipMalloc(&dw16, W16.size() * 2);
  (void)hipMalloc(&dw8, W8.size()); (void)hipMalloc(&dws, WS.size() * 4);
  (void)hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dw16, W16.data(), W16.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dw8, W8.data(), W8.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(dws, WS.data(), WS.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> Y((size_t)M * N), Yhi((size_t)M * N);
  proto<<<dim3(N / 32, M / 32), 64>>>(dx, dw16, dw8, dws, dy, M, N, K, 1);
  (void)hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
  proto<<<dim3(N / 32, M / 32), 64>>>(dx, dw16, dw8, dws, dy, M, N, K, 0);
  (void)hipMemcpy(Yhi.data(), dy, Yhi.size() * 4, hipMemcpyDeviceToHost);
  // host: exact product, the same scheme, today's bf16 hi + lo, one fp16 pass
  double peak = 0, e_dev = 0, e_host = 0, e_dev_vs_host = 0, e_bf = 0, e_hi = 0, e_dev_hi = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double exact = 0, sch = 0, bf = 0, hi = 0;
      for (int kb = 0; kb < K / 32; ++kb) {
        float res[32];
        for (int kk = 0; kk < 32; ++kk) {
          const float xv = X[(size_t)m * K + kb * 32 + kk];
          res[kk] = xv - (float)(_Float16)xv;
        }
        const int sbyte = block_scale_byte(res, 32);
        const float sc = ldexpf(1.f, sbyte - 127);
        for (int kk = 0; kk < 32; ++kk) {
          const size_t k = (size_t)kb * 32 + kk;
          const float xv = X[(size_t)m * K + k], wv = W[(size_t)n * K + k];
          exact += (double)xv * wv;
          const float h = (float)(_Float16)xv;
          hi += (double)h * wv;
          sch += (double)h * wv + (double)(e4m3_dec(e4m3_enc(res[kk] / sc)) * sc) * Wq[(size_t)n * K + k];
          const float hb = bf16r(xv);
          bf += ((double)hb + (double)bf16r(xv - hb)) * wv;
        }
      }
      const double d = Y[(size_t)m * N + n];
      peak = fmax(peak, fabs(exact));
      e_dev = fmax(e_dev, fabs(d - exact));
      e_host = fmax(e_host, fabs(sch - exact));
      e_dev_vs_host = fmax(e_dev_vs_host, fabs(d - sch));
      e_bf = fmax(e_bf, fabs(bf - exact));
      e_hi = fmax(e_hi, fabs(hi - exact));
      e_dev_hi = fmax(e_dev_hi, fabs((double)Yhi[(size_t)m * N + n] - hi));
    }
  printf("{\"M\": %d, \"N\": %d, \"K\": %d, \"peak\": %.4g, \"device_fp16hi_plus_mx8lo_vs_exact\": %.3g, \"host_same_scheme_vs_exact\": %.3g, \"device_vs_host_same_scheme\": %.3g, "
         "\"bf16_hi_lo_vs_exact\": %.3g, \"fp16_hi_only_vs_exact\": %.3g, \"device_hi_only_vs_host_hi_only\": %.3g}\n",
         M, N, K, peak, e_dev / peak, e_host / peak, e_dev_vs_host / peak, e_bf / peak, e_hi / peak, e_dev_hi / peak);
  return 0;
}
