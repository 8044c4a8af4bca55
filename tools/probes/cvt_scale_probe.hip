// probe: semantics of v_cvt_scalef32_pk_fp8_f32 on gfx950 (does the scale operand divide or multiply; which bits of it are used)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o, const float* x, const float* scale, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  v2s old = {0, 0};
  v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[0], x[1], scale[i], false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[2], x[3], scale[i], true);
  o[2 * i] = __builtin_bit_cast(unsigned, r);
  int pk = 0;   // the two-step path of conv_ws4.h: multiply by 1 / scale (exact power of two), then v_cvt_pk_fp8_f32
  const float m = 1.0f / scale[i];
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * m, x[1] * m, pk, false);
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * m, x[3] * m, pk, true);
  o[2 * i + 1] = (unsigned)pk;
}
int main() {
  float hx[4] = {0.3f, -1.7f, 5.25f, 100.0f};
  float hs[6] = {1.0f, 2.0f, 0.5f, 0.0078125f, 3.0f, 64.0f};
  float *dx, *ds; unsigned* dout; unsigned ho[12];
  hipMalloc(&dx, 16); hipMalloc(&ds, 24); hipMalloc(&dout, 48);
  hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice); hipMemcpy(ds, hs, 24, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, dx, ds, 6);
  hipMemcpy(ho, dout, 48, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("scale %g: scalef32 %08x   mul-by-1/scale then cvt %08x   %s\n", hs[i], ho[2 * i], ho[2 * i + 1], ho[2 * i] == ho[2 * i + 1] ? "SAME (divides by scale)" : "differs");
  return 0;
}
