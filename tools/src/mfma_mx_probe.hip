// How does v_mfma_scale_f32_32x32x64_f8f6f4 read its operands and its block scales?  (tools, not product; the ISA tables are not in this image.)
// A lo pass of the conv kernels on the block-scaled 8-bit pipe (gap 1) needs three facts:
//   (1) which K index a lane's byte j holds               -> only matters for the scales: with one packing for A and B any bijection gives the product
//   (2) which bytes one E8M0 scale covers                 -> hypotheses S1 / S2 below
//   (3) where the C / D elements land                     -> the guide says: as for every 32x32 shape (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
// One workgroup of one wave: A [32 x 64] and B [64 x 32] of random e4m3 values, packed under the "obvious" layout H1 (lane l: row / column l & 31,
// K = 32 (l >> 5) + j for byte j of its eight VGPRs), one MFMA, D compared on the host with the exact product under each scale hypothesis:
//   unit scales                     -> checks (1) + (3)
//   scale_a = 2^(l >> 5) per lane   -> S1: a lane's scale covers its own 32 bytes (K block = lane's bytes)
//                                      S2: bytes 0..15 of every lane form one block with the OTHER half's bytes 0..15 (a K-interleaved block)
//   scale_a varies with the row     -> the scale belongs to the row the lane holds
//   opsel 1..3                      -> which byte of the scale VGPR is read
// RESULT (call 54, profiles/r3_mfma_mx_probe_call54.jsonl): products and C / D placement are right under H1 with unit scales (8e-6 of the peak); the scale cases
// match S2 only: ONE E8M0 SCALE COVERS BYTES 0..15 OF A LANE TOGETHER WITH BYTES 0..15 OF THE LANE 32 ABOVE IT, i.e. byte j of lane l holds
//   K = 32 (j >> 4) + 16 (l >> 5) + (j & 15),     the scale of K block b (32 elements) of row / column i is read from lane i + 32 b,
// byte OPSEL of the scale VGPR.  A kernel that packs both operands under H1 still gets the right product with unit scales (any bijection does).
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
    fa[j] = ((const int*)(a + l * 32))[j];
    fb[j] = ((const int*)(b + l * 32))[j];
  }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 0, 0, OPSEL, sa[l], OPSEL, sb[l]);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

static float e4m3_to_float(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
  return s ? -x : x;
}

int main() {
  uint8_t ha[64 * 32], hb[64 * 32];
  float A[32][64], B[64][32];
  uint32_t s = 777u;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      s = s * 1664525u + 1013904223u;
      const uint8_t va = (uint8_t)(((s >> 31) << 7) | ((5 + ((s >> 9) % 5)) << 3) | ((s >> 13) & 7));
      s = s * 1664525u + 1013904223u;
      const uint8_t vb = (uint8_t)(((s >> 31) << 7) | ((5 + ((s >> 9) % 5)) << 3) | ((s >> 13) & 7));
      ha[l * 32 + j] = va;
      hb[l * 32 + j] = vb;
      A[l & 31][32 * (l >> 5) + j] = e4m3_to_float(va);   // H1
      B[32 * (l >> 5) + j][l & 31] = e4m3_to_float(vb);
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
    // hypotheses for the A-side scale of element (row i, k): S1 = scale of the lane that holds it; S2 = blocks interleave the two lane halves in
    // groups of 16 bytes (block 0 = bytes 0..15 of both halves); S0 = scales ignored
    double err[4] = {0, 0, 0, 0}, peak = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double want[4] = {0, 0, 0, 0};
        for (int k = 0; k < 64; ++k) {
          const int lane_a = row + 32 * (k >> 5), lane_b = col + 32 * (k >> 5), j = k & 31;
          const double ab = (double)A[row][k] * (double)B[k][col];
          auto ex = [&](const int* sv, int lane) { return ldexp(1.0, ((sv[lane] >> (8 * c.opsel)) & 255) - 127); };
          want[0] += ab;
          want[1] += ab * ex(hsa, lane_a) * ex(hsb, lane_b);
          want[2] += ab * ex(hsa, row + 32 * (j >> 4)) * ex(hsb, col + 32 * (j >> 4));
          want[3] += ab * ldexp(1.0, (hsa[lane_a] & 255) - 127) * ldexp(1.0, (hsb[lane_b] & 255) - 127);   // S1 with byte 0 whatever opsel says
        }
        for (int h = 0; h < 4; ++h) err[h] = fmax(err[h], fabs(want[h] - (double)hd[l * 16 + r]));
        peak = fmax(peak, fabs(want[1]));
      }
    printf("{\"case\": \"%s\", \"peak\": %.4g, \"err_scales_ignored\": %.3g, \"err_S1_lane_owns_its_32_bytes\": %.3g, \"err_S2_interleaved_16\": %.3g, "
           "\"err_S1_byte0_regardless_of_opsel\": %.3g}\n", c.name, peak, err[0], err[1], err[2], err[3]);
  }
  return 0;
}
