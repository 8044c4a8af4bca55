// Can this box run its matrix pipe and its HBM stream at their separate rates AT THE SAME TIME?  (tools, not product.)
// The conv kernel's measured times obey  t ~= t_mfma(at the sustained MFMA rate) + t_hbm(at the streaming rate)  shape by shape
// (k = 3 / 7 / 11 at C = 128: profiles/r3_conv_concurrent_streams_call17.txt "alone" row) -- as if nothing overlapped, although its producer and
// consumer waves do run concurrently.  This probe separates the two explanations: a scheduling problem of that kernel, or a shared power budget.
// One kernel, 8 waves per workgroup, 2 workgroups per CU: waves 0-3 issue back-to-back v_mfma_f32_32x32x16_bf16 on register operands (random
// bit patterns), waves 4-7 stream a large buffer with 16-byte loads (each wave 1 KiB per instruction, `depth` loads in flight), both for a fixed
// wall time.  Modes: matrix pipe only, stream only, both.  Reported per mode: TFLOP/s of the MFMA waves, TB/s of the streaming waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ticks of the constant 100 MHz counter
__device__ __forceinline__ uint64_t wall() { return wall_clock64(); }

__global__ __launch_bounds__(512) void mix_kernel(const uint4* __restrict__ src, const uint4* __restrict__ big, const size_t big_n16, float* __restrict__ sink,
                                                  unsigned long long* __restrict__ counts, const int do_mfma, const int do_stream, const uint64_t ticks,
                                                  const int stream_every, const int prio, const int yield_mode) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint64_t t_end = wall() + ticks;
  if (wave < 4) {
    if (!do_mfma) return;
    if ((prio >> 2) == 3) __builtin_amdgcn_s_setprio(3);
    const int t = blockIdx.x * 256 + tid;
    const uint4 ra0 = src[(t * 4 + 0) & 65535], ra1 = src[(t * 4 + 1) & 65535], rb0 = src[(t * 4 + 2) & 65535], rb1 = src[(t * 4 + 3) & 65535];
    f32x16 a0, a1, a2, a3;
#pragma unroll
    for (int j = 0; j < 16; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; a3[j] = 0.f; }
    unsigned long long n = 0;
    while (wall() < t_end) {
#pragma unroll 1
      for (int it = 0; it < 64; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra0), __builtin_bit_cast(bf16x8, rb0), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra1), __builtin_bit_cast(bf16x8, rb0), a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra0), __builtin_bit_cast(bf16x8, rb1), a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra1), __builtin_bit_cast(bf16x8, rb1), a3, 0, 0, 0);
        if (yield_mode == 1) __builtin_amdgcn_s_sleep(1);            // 64 idle cycles after every 4 MFMAs (128 busy cycles)
        else if (yield_mode == 2) { if ((it & 3) == 3) __builtin_amdgcn_s_sleep(1); }   // ... after every 16
        else if (yield_mode == 3) { if ((it & 15) == 15) __builtin_amdgcn_s_sleep(1); } // ... after every 64
        else if (yield_mode == 4) asm volatile("s_nop 0" ::: "memory");
      }
      n += 256;
    }
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += a0[j] + a1[j] + a2[j] + a3[j];
    if (tot == 12345.678f) sink[t] = tot;
    if (lane == 0) atomicAdd(&counts[0], n);
  } else {
    if (!do_stream) return;
    if ((prio & 3) == 3) __builtin_amdgcn_s_setprio(3);
    // every streaming wave walks the buffer in 8 KiB steps (8 loads of 1 KiB in flight), waves interleaved over the whole grid
    const size_t nwaves = (size_t)gridDim.x * 4, me = (size_t)blockIdx.x * 4 + (wave - 4);
    size_t pos = me * 512 + lane;   // uint4 index; a wave covers 512 uint4 = 8 KiB per trip
    float tot = 0.f;
    unsigned long long n = 0;
    while (wall() < t_end) {
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)big + ((pos + 64 * u) & (big_n16 - 1)));
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += __builtin_bit_cast(float, v[u][0] ^ v[u][3]);
        pos = (pos + nwaves * 512) & (big_n16 - 1);
        if (stream_every == 4) __builtin_amdgcn_s_sleep(4);
        else if (stream_every == 16) __builtin_amdgcn_s_sleep(16);
        else if (stream_every == 64) __builtin_amdgcn_s_sleep(64);
      }
      n += 8 * 8;
    }
    if (tot == 12345.678f) sink[blockIdx.x * 512 + tid] = tot;
    if (lane == 0) atomicAdd(&counts[1], n);   // 1 KiB loads
  }
}

__global__ void fill_kernel(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double ms = argc > 1 ? atof(argv[1]) : 200.0;
  uint4 *src, *big;
  float* sink;
  unsigned long long* counts;
  const size_t nsrc = 65536, big_bytes = (size_t)8 << 30, big_n16 = big_bytes / 16;
  hipMalloc(&src, nsrc * 16);
  hipMalloc(&big, big_bytes);
  hipMalloc(&sink, (size_t)cus * 2 * 512 * 4);
  hipMalloc(&counts, 16);
  fill_kernel<<<dim3(4096), dim3(256)>>>((uint32_t*)big, big_bytes / 4);   // pseudo-random words: the activity factor of real data on the HBM pins
  hipDeviceSynchronize();
  uint16_t* h = (uint16_t*)malloc(nsrc * 16);
  uint32_t s = 12345u;
  for (size_t i = 0; i < nsrc * 8; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = (uint16_t)(((s >> 31) << 15) | ((120 + ((s >> 8) & 7)) << 7) | ((s >> 12) & 0x7f));
  }
  hipMemcpy(src, h, nsrc * 16, hipMemcpyHostToDevice);
  const uint64_t ticks = (uint64_t)(ms * 1e-3 * 100e6);
  struct { const char* name; int m, st, sleep, prio, yield; } modes[] = {
      {"mfma_only", 1, 0, 0, 0, 0}, {"stream_only", 0, 1, 0, 0, 0}, {"both", 1, 1, 0, 0, 0}, {"both_stream_prio3", 1, 1, 0, 3, 0},
      {"mfma_only_yield4", 1, 0, 0, 0, 1}, {"both_yield4", 1, 1, 0, 0, 1}, {"mfma_only_yield16", 1, 0, 0, 0, 2}, {"both_yield16", 1, 1, 0, 0, 2},
      {"mfma_only_yield64", 1, 0, 0, 0, 3}, {"both_yield64", 1, 1, 0, 0, 3}, {"both_yield64_stream_prio3", 1, 1, 0, 3, 3}, {"both_nop", 1, 1, 0, 0, 4},
      {"both_yield16_stream_sleep16", 1, 1, 16, 0, 2}, {"both_yield16_stream_sleep64", 1, 1, 64, 0, 2}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto& md : modes) {
    float el = 0.f;
    unsigned long long hc[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep) {   // the second run is reported (clock settled)
      hipMemset(counts, 0, 16);
      hipEventRecord(e0, 0);
      mix_kernel<<<dim3(cus * 2), dim3(512)>>>(src, big, big_n16, sink, counts, md.m, md.st, ticks, md.sleep, md.prio, md.yield);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&el, e0, e1);
      hipMemcpy(hc, counts, 16, hipMemcpyDeviceToHost);
    }
    const double tflops = (double)hc[0] * 2.0 * 32 * 32 * 16 / (el * 1e-3) / 1e12;
    const double tbs = (double)hc[1] * 1024.0 / (el * 1e-3) / 1e12;
    printf("{\"mode\": \"%s\", \"ms\": %.2f, \"mfma_tflops\": %.1f, \"mfma_frac_of_2500\": %.3f, \"stream_TBps\": %.3f}\n", md.name, el, tflops, tflops / 2500.0, tbs);
    fflush(stdout);
  }
  return 0;
}
