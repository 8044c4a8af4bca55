// How does v_mfma_scale_f32_32x32x64_f8f6f4 read FP4 (e2m1) operands and their block scales, and what exactly does v_cvt_scalef32_pk_fp4_f32 produce?
// (tools, not product; the ISA tables are not in this image.)  Companion of mfma_mx_probe.hip (the e4m3 answers: call 54 of round 3).
//   A  one wave, A [32 x 64] and B [64 x 32] of random e2m1 values packed as: nibble j (bits 4 j .. 4 j + 3 of the lane's 128-bit operand) of lane l holds
//      K = 32 (j >> 4) + 16 (l >> 5) + (j & 15) for row / column l & 31 (the e4m3 layout with 4-bit elements); D against the exact product under
//        S0 scales ignored | S1 a lane's scale covers its own 32 nibbles | S2 one scale covers nibbles 0..15 of a lane AND of the lane 32 above it
//   B  v_cvt_scalef32_pk_fp4_f32 on a table of inputs: which nibble src0 lands in, the byte op_sel picks, rounding, saturation, the scale's meaning
// Output: JSON lines.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OPSEL>
__global__ void probe(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* d) {
  const int l = threadIdx.x;
  i32x8 fa, fb;
  for (int j = 0; j < 8; ++j) {
    fa[j] = j < 4 ? ((const int*)(a + l * 16))[j] : 0;
    fb[j] = j < 4 ? ((const int*)(b + l * 16))[j] : 0;
  }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 4, 4, OPSEL, sa[l], OPSEL, sb[l]);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

__global__ void cvt_probe(const float* x, const float* scl, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  unsigned v0 = 0xffffffffu, v1 = 0xffffffffu, v2 = 0xffffffffu, v3 = 0xffffffffu;
  v0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(v0, x[2 * i], x[2 * i + 1], scl[i], 0);
  v1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(v1, x[2 * i], x[2 * i + 1], scl[i], 1);
  v2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(v2, x[2 * i], x[2 * i + 1], scl[i], 2);
  v3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(v3, x[2 * i], x[2 * i + 1], scl[i], 3);
  out[4 * i] = v0; out[4 * i + 1] = v1; out[4 * i + 2] = v2; out[4 * i + 3] = v3;
}

static float e2m1_to_float(int v) {
  static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  return (v & 8) ? -mag[v & 7] : mag[v & 7];
}

int main() {
  uint8_t ha[64 * 16], hb[64 * 16];
  float A[32][64], B[64][32];
  uint32_t s = 4242u;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      s = s * 1664525u + 1013904223u;
      const int va = (s >> 20) & 15;
      s = s * 1664525u + 1013904223u;
      const int vb = (s >> 20) & 15;
      uint8_t& pa = ha[l * 16 + (j >> 1)];
      uint8_t& pb = hb[l * 16 + (j >> 1)];
      if (j & 1) { pa = (uint8_t)((pa & 0x0f) | (va << 4)); pb = (uint8_t)((pb & 0x0f) | (vb << 4)); }
      else { pa = (uint8_t)va; pb = (uint8_t)vb; }
      const int k = 32 * (j >> 4) + 16 * (l >> 5) + (j & 15);
      A[l & 31][k] = e2m1_to_float(va);
      B[k][l & 31] = e2m1_to_float(vb);
    }
  uint8_t *da, *db;
  int *dsa, *dsb;
  float* dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 16 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  struct Case { const char* name; int mode; int opsel; } cases[] = {
      {"unit scales", 0, 0}, {"scale_a = 2^(lane >> 5)", 1, 0}, {"scale_a = 2^(row % 3), scale_b = 2^(col % 2)", 2, 0},
      {"scale bytes differ, opsel 0", 3, 0}, {"scale bytes differ, opsel 1", 3, 1}, {"scale bytes differ, opsel 2", 3, 2}, {"scale bytes differ, opsel 3", 3, 3}};
  for (const Case& c : cases) {
    int hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
      int ea = 127, eb = 127;
      if (c.mode == 1) ea = 127 + (l >> 5);
      if (c.mode == 2) { ea = 127 + ((l & 31) % 3); eb = 127 + ((l & 31) % 2); }
      hsa[l] = ea * 0x01010101;
      hsb[l] = eb * 0x01010101;
      if (c.mode == 3) { hsa[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24); hsb[l] = 0x7f7f7f7f; }
    }
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    if (c.opsel == 0) probe<0><<<1, 64>>>(da, db, dsa, dsb, dd);
    if (c.opsel == 1) probe<1><<<1, 64>>>(da, db, dsa, dsb, dd);
    if (c.opsel == 2) probe<2><<<1, 64>>>(da, db, dsa, dsb, dd);
    if (c.opsel == 3) probe<3><<<1, 64>>>(da, db, dsa, dsb, dd);
    float hd[64 * 16];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    double err[3] = {0, 0, 0}, peak = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double want[3] = {0, 0, 0};
        for (int k = 0; k < 64; ++k) {
          // under the packing above element k sits in lane (row | col) + 32 ((k >> 4) & 1), K block (S2) = k >> 5
          const int half = (k >> 4) & 1, blk = k >> 5;
          const double ab = (double)A[row][k] * (double)B[k][col];
          auto ex = [&](const int* sv, int lane) { return ldexp(1.0, ((sv[lane] >> (8 * c.opsel)) & 255) - 127); };
          want[0] += ab;
          want[1] += ab * ex(hsa, row + 32 * half) * ex(hsb, col + 32 * half);   // S1: the holding lane's scale
          want[2] += ab * ex(hsa, row + 32 * blk) * ex(hsb, col + 32 * blk);     // S2: block b's scale comes from lane i + 32 b
        }
        for (int h = 0; h < 3; ++h) err[h] = fmax(err[h], fabs(want[h] - (double)hd[l * 16 + r]));
        peak = fmax(peak, fabs(want[2]));
      }
    printf("{\"part\": \"A\", \"case\": \"%s\", \"peak\": %.4g, \"err_scales_ignored\": %.3g, \"err_S1_lane_owns_its_32_nibbles\": %.3g, \"err_S2_interleaved_16\": %.3g}\n",
           c.name, peak, err[0], err[1], err[2]);
  }
  // ---- part B: the conversion
  const float xs[] = {0.f, 0.24f, 0.25f, 0.26f, 0.5f, 0.74f, 0.75f, 0.76f, 1.0f, 1.25f, 1.5f, 1.75f, 2.0f, 2.5f, 3.0f, 3.5f, 4.0f, 5.0f, 6.0f, 7.0f, 8.0f, 100.f,
                      -0.25f, -0.75f, -1.25f, -2.5f, -5.0f, -7.0f, 1e-30f, -1e-30f};
  const int nx = sizeof(xs) / sizeof(xs[0]);
  const float scales[] = {1.0f, 2.0f, 0.5f, 3.0f /* mantissa bits set: is only the exponent used? */, 1.0f / 1024.f};
  const int nsc = sizeof(scales) / sizeof(scales[0]);
  float hx[2 * 64], hs[64];
  unsigned ho[4 * 64];
  float *dx, *dsc;
  unsigned* dout;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dsc, sizeof(hs)); hipMalloc(&dout, sizeof(ho));
  for (int si = 0; si < nsc; ++si) {
    for (int base = 0; base < nx; base += 64) {
      const int n = nx - base < 64 ? nx - base : 64;
      for (int i = 0; i < n; ++i) { hx[2 * i] = xs[base + i] * scales[si]; hx[2 * i + 1] = -0.5f * scales[si]; hs[i] = scales[si]; }   // src1 = -0.5 * scale: code 9 wherever it lands
      hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
      hipMemcpy(dsc, hs, sizeof(hs), hipMemcpyHostToDevice);
      cvt_probe<<<1, 64>>>(dx, dsc, dout, n);
      hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
      for (int i = 0; i < n; ++i)
        printf("{\"part\": \"B\", \"scale\": %g, \"x_over_scale\": %g, \"sel0\": \"%08x\", \"sel1\": \"%08x\", \"sel2\": \"%08x\", \"sel3\": \"%08x\"}\n", scales[si], xs[base + i],
               ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
    }
  }
  return 0;
}
