// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream;
// "other access widths are uncalibrated").  Four copy kernels over the same 1 GiB: 16 B per lane, 4 B per lane (one 128-B line per half
// wave: the row-per-register conv epilogue / fold pattern), 16 B per lane at a 512-B row pitch (the transposed conv epilogue pattern:
// 32 rows x 32 B per wave instruction), and 8 B per lane.  Run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void copy_x4(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void copy_x1(const float* __restrict__ s, float* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void copy_x2(const float2* __restrict__ s, float2* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// [rows][128 floats]: a wave covers 32 rows x 8 floats per instruction (lane l: row l & 31, floats 4 * (l >> 5) + 8 * q .. +4), q = 0..15
__global__ void copy_rows32B(const float* __restrict__ s, float* __restrict__ d, size_t rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r0 = wave * 32; r0 < rows; r0 += nw * 32) {
    const size_t base = (r0 + (lane & 31)) * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 16; ++q) *(float4*)(d + base + 8 * q) = *(const float4*)(s + base + 8 * q);
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float *s, *d;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes);
  hipMemset(d, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    float ms[4];
    hipEventRecord(e0); copy_x4<<<2048, 256>>>((const float4*)s, (float4*)d, bytes / 16); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
    hipEventRecord(e0); copy_x1<<<2048, 256>>>(s, d, bytes / 4); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
    hipEventRecord(e0); copy_x2<<<2048, 256>>>((const float2*)s, (float2*)d, bytes / 8); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
    hipEventRecord(e0); copy_rows32B<<<2048, 256>>>(s, d, bytes / 512); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[3], e0, e1);
    printf("rep %d: 1 GiB read + 1 GiB written per kernel: x4 %.3f ms (%.2f TB/s)  x1 %.3f ms (%.2f)  x2 %.3f ms (%.2f)  rows32B %.3f ms (%.2f)\n", rep,
           ms[0], 2.0 * bytes / ms[0] / 1e9, ms[1], 2.0 * bytes / ms[1] / 1e9, ms[2], 2.0 * bytes / ms[2] / 1e9, ms[3], 2.0 * bytes / ms[3] / 1e9);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status %s\n", hipGetErrorString(e));
  return e != hipSuccess;
}
