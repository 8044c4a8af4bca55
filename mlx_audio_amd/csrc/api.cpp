// Error reporting, ABI version and host-side weight packing for libmi355audio.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void mi355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_abi_version(void) { return 36; }

extern "C" int mi355_device_info(int dev, char* name, int name_cap, int* cu_count, int* lds_bytes) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    mi355_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return MI355_ERR_LAUNCH;
  }
  if (name && name_cap > 0) snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName);
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  return MI355_OK;
}

// ---- conv weight packing: [Cout, K, Cin] fp32 -> MFMA 32x32x16 B-fragment order, bf16.
// element index = ((((chunk*K + tap)*NTp + nt)*2 + kk)*64 + lane)*8 + j with
//   n = nt*32 + (lane & 31),  c = chunk*32 + kk*16 + (lane >> 5)*8 + j
extern "C" int64_t mi355_packed_conv_weight_elems(int32_t Cout, int32_t K, int32_t Cin) {
  int64_t chunks = (Cin + 31) / 32;
  int64_t ntp = ((Cout + 127) / 128) * 4;
  return chunks * K * ntp * 2 * 512;
}

static int pack_conv_weight(const float* w, int32_t Cout, int32_t K, int32_t Cin, int32_t dtype, uint16_t* out) {
  MI355_REQUIRE(w && out && Cout > 0 && K > 0 && Cin > 0, "pack_conv_weight: bad arguments");
  MI355_REQUIRE(dtype == MI355_W_BF16 || dtype == MI355_W_F16, "pack_conv_weight: dtype must be MI355_W_BF16 or MI355_W_F16");
  const int chunks = (Cin + 31) / 32;
  const int ntp = ((Cout + 127) / 128) * 4;
  size_t o = 0;
  for (int ch = 0; ch < chunks; ++ch)
    for (int tap = 0; tap < K; ++tap)
      for (int nt = 0; nt < ntp; ++nt)
        for (int kk = 0; kk < 2; ++kk)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = nt * 32 + (lane & 31);
            const int c0 = ch * 32 + kk * 16 + (lane >> 5) * 8;
            for (int j = 0; j < 8; ++j) {
              const int c = c0 + j;
              float v = (n < Cout && c < Cin) ? w[((size_t)n * K + tap) * Cin + c] : 0.0f;
              out[o++] = dtype == MI355_W_F16 ? host_f32_to_f16(v) : host_f32_to_bf16(v);
            }
          }
  return MI355_OK;
}

// ---- MX image (precision 5: fp16 hi pass + block-scaled e4m3 lo pass).  Per 32-channel chunk: K fp16 tap slices in the layout above, then
// NP = ceil(K / 2) e4m3 tap-PAIR slices of the same size (NTp x 2 KB): byte index within a pair slice = ((nt * 2 + h) * 64 + lane) * 16 + j with
//   n = nt*32 + (lane & 31),  tap = 2 p + h,  c = chunk*32 + (lane >> 5)*16 + j      (h = K block of v_mfma_scale_f32_32x32x64_f8f6f4),
// value = e4m3(w / 2^e[n]), zero for tap >= K or padding.  After the last slice: NTp*32 E8M0 bytes, one per output column:
// e[n] = floor(log2(max |w[n, :, :]|)) - 7 (the scaled column maximum lies in [128, 256) < 448), byte = e + 127; 127 for an all-zero column.
extern "C" int64_t mi355_packed_conv_weight_mx_bytes(int32_t Cout, int32_t K, int32_t Cin) {
  const int64_t chunks = (Cin + 31) / 32;
  const int64_t ntp = ((Cout + 127) / 128) * 4;
  return chunks * (K + (K + 1) / 2) * ntp * 2048 + ntp * 32;
}

extern "C" int mi355_pack_conv_weight_mx_host(const float* w, int32_t Cout, int32_t K, int32_t Cin, uint8_t* out) {
  MI355_REQUIRE(w && out && Cout > 0 && K > 0 && Cin > 0, "pack_conv_weight_mx: bad arguments");
  const int chunks = (Cin + 31) / 32;
  const int ntp = ((Cout + 127) / 128) * 4;
  const int np = (K + 1) / 2;
  uint8_t* scales = out + (size_t)chunks * (K + np) * ntp * 2048;
  std::vector<float> inv(ntp * 32, 1.0f);
  for (int n = 0; n < ntp * 32; ++n) {
    float amax = 0.f;
    if (n < Cout)
      for (size_t e = 0; e < (size_t)K * Cin; ++e) {
        const float v = fabsf(w[(size_t)n * K * Cin + e]);
        MI355_REQUIRE(v == v && v <= 3.0e38f, "pack_conv_weight_mx: non-finite weight in output channel %d", n);
        amax = v > amax ? v : amax;
      }
    int e = 0;
    if (amax > 0.f) {
      (void)frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1): floor(log2(amax)) = e - 1
      e = e - 1 - 7;
      if (e < -127) e = -127;
      if (e > 120) e = 120;
    }
    scales[n] = (uint8_t)(e + 127);
    inv[n] = ldexpf(1.0f, -e);
  }
  size_t o = 0;
  uint16_t* o16 = (uint16_t*)out;
  for (int ch = 0; ch < chunks; ++ch) {
    for (int tap = 0; tap < K; ++tap)
      for (int nt = 0; nt < ntp; ++nt)
        for (int kk = 0; kk < 2; ++kk)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = nt * 32 + (lane & 31);
            const int c0 = ch * 32 + kk * 16 + (lane >> 5) * 8;
            for (int j = 0; j < 8; ++j) {
              const int c = c0 + j;
              const float v = (n < Cout && c < Cin) ? w[((size_t)n * K + tap) * Cin + c] : 0.0f;
              o16[o++] = host_f32_to_f16(v);
            }
          }
    uint8_t* o8 = out + o * 2;
    size_t b = 0;
    for (int p = 0; p < np; ++p)
      for (int nt = 0; nt < ntp; ++nt)
        for (int h = 0; h < 2; ++h)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = nt * 32 + (lane & 31);
            const int tap = 2 * p + h;
            const int c0 = ch * 32 + (lane >> 5) * 16;
            for (int j = 0; j < 16; ++j) {
              const int c = c0 + j;
              const float v = (n < Cout && c < Cin && tap < K) ? w[((size_t)n * K + tap) * Cin + c] : 0.0f;
              o8[b++] = host_f32_to_e4m3(v * inv[n]);   // power-of-two scaling: exact
            }
          }
    o += b / 2;
  }
  return MI355_OK;
}

// ---- MX4 image (precision 6: fp16 hi pass + block-scaled FP4 lo pass).  The MX image's geometry (same byte count, same fp16 tap slices, same slot of
// NTp x 2 KB per tap PAIR, same trailer of NTp*32 E8M0 column bytes); a pair slot holds, per 32-column group nt, ONE kilobyte of e2m1 codes at
// ((nt * 2) * 64 + lane) * 16 + j / 2 (low nibble = even j) and one kilobyte of zeros:
//   n = nt*32 + (lane & 31),  tap = 2 p + (lane >> 5),  c = chunk*32 + j,  j = 0..31
// (for 4-bit operands v_mfma_scale_f32_32x32x64_f8f6f4 takes a lane's 32 elements as ONE scale block -- probed: tools/src/mfma_fp4_probe.hip,
// profiles/r6_mfma_fp4_probe_call2.jsonl -- so K block b = tap 2 p + b sits whole in lane half b), value = e2m1(w / 2^e[n]), zero for tap >= K or
// padding; e[n] = floor(log2(max |w[n, :, :]|)) - 2 (OCP MX: the scaled column maximum lies in [4, 8), elements above 6 saturate), byte = e + 127.
extern "C" int mi355_pack_conv_weight_mx4_host(const float* w, int32_t Cout, int32_t K, int32_t Cin, uint8_t* out) {
  MI355_REQUIRE(w && out && Cout > 0 && K > 0 && Cin > 0, "pack_conv_weight_mx4: bad arguments");
  const int chunks = (Cin + 31) / 32;
  const int ntp = ((Cout + 127) / 128) * 4;
  const int np = (K + 1) / 2;
  uint8_t* scales = out + (size_t)chunks * (K + np) * ntp * 2048;
  std::vector<float> inv(ntp * 32, 1.0f);
  for (int n = 0; n < ntp * 32; ++n) {
    float amax = 0.f;
    if (n < Cout)
      for (size_t e = 0; e < (size_t)K * Cin; ++e) {
        const float v = fabsf(w[(size_t)n * K * Cin + e]);
        MI355_REQUIRE(v == v && v <= 3.0e38f, "pack_conv_weight_mx4: non-finite weight in output channel %d", n);
        amax = v > amax ? v : amax;
      }
    int e = 0;
    if (amax > 0.f) {
      (void)frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1): floor(log2(amax)) = e - 1
      e = e - 1 - 2;
      if (e < -127) e = -127;
      if (e > 120) e = 120;
    }
    scales[n] = (uint8_t)(e + 127);
    inv[n] = ldexpf(1.0f, -e);
  }
  size_t o = 0;
  uint16_t* o16 = (uint16_t*)out;
  for (int ch = 0; ch < chunks; ++ch) {
    for (int tap = 0; tap < K; ++tap)
      for (int nt = 0; nt < ntp; ++nt)
        for (int kk = 0; kk < 2; ++kk)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = nt * 32 + (lane & 31);
            const int c0 = ch * 32 + kk * 16 + (lane >> 5) * 8;
            for (int j = 0; j < 8; ++j) {
              const int c = c0 + j;
              const float v = (n < Cout && c < Cin) ? w[((size_t)n * K + tap) * Cin + c] : 0.0f;
              o16[o++] = host_f32_to_f16(v);
            }
          }
    uint8_t* o8 = out + o * 2;
    memset(o8, 0, (size_t)np * ntp * 2048);
    for (int p = 0; p < np; ++p)
      for (int nt = 0; nt < ntp; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = nt * 32 + (lane & 31);
          const int tap = 2 * p + (lane >> 5);
          uint8_t* dst = o8 + ((size_t)p * ntp + nt) * 2048 + (size_t)lane * 16;
          for (int j = 0; j < 32; ++j) {
            const int c = ch * 32 + j;
            const float v = (n < Cout && c < Cin && tap < K) ? w[((size_t)n * K + tap) * Cin + c] : 0.0f;
            const uint8_t code = host_f32_to_e2m1(v * inv[n]);   // power-of-two scaling: exact
            dst[j >> 1] |= (uint8_t)((j & 1) ? (code << 4) : code);
          }
        }
    o += (size_t)np * ntp * 1024;
  }
  return MI355_OK;
}

extern "C" int mi355_pack_conv_weight_host(const float* w, int32_t Cout, int32_t K, int32_t Cin, uint16_t* out) {
  return pack_conv_weight(w, Cout, K, Cin, MI355_W_BF16, out);
}

extern "C" int mi355_pack_conv_weight_host_dt(const float* w, int32_t Cout, int32_t K, int32_t Cin, int32_t dtype, uint16_t* out) {
  return pack_conv_weight(w, Cout, K, Cin, dtype, out);
}

// ---- LSTM recurrent weight packing: Wh [4H, H] fp32 (fwd, bwd) -> [2][H/8][4H][8] bf16
extern "C" int mi355_pack_lstm_wh_host(const float* wf, const float* wb, int32_t H, uint16_t* out) {
  MI355_REQUIRE(wf && wb && out && H > 0 && H % 8 == 0, "pack_lstm_wh: bad arguments");
  const int G = 4 * H;
  size_t o = 0;
  for (int d = 0; d < 2; ++d) {
    const float* w = d ? wb : wf;
    for (int k8 = 0; k8 < H / 8; ++k8)
      for (int r = 0; r < G; ++r)
        for (int j = 0; j < 8; ++j) out[o++] = host_f32_to_bf16(w[(size_t)r * H + k8 * 8 + j]);
  }
  return MI355_OK;
}

extern "C" int mi355_pack_lstm_wh16_host(const float* wf, const float* wb, int32_t H, int32_t f16, uint16_t* out) {
  if (!f16) return mi355_pack_lstm_wh_host(wf, wb, H, out);
  MI355_REQUIRE(wf && wb && out && H > 0 && H % 8 == 0, "pack_lstm_wh16: bad arguments");
  const int G = 4 * H;
  size_t o = 0;
  for (int d = 0; d < 2; ++d) {
    const float* w = d ? wb : wf;
    for (int k8 = 0; k8 < H / 8; ++k8)
      for (int r = 0; r < G; ++r)
        for (int j = 0; j < 8; ++j) out[o++] = host_f32_to_f16(w[(size_t)r * H + k8 * 8 + j]);
  }
  return MI355_OK;
}
