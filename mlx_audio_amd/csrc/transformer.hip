// Row-wise glue kernels of the decoder-only transformer blocks under mlx_audio/tts (Qwen3-TTS talker / code predictor /
// codec transformer, CSM Llama backbone + depth decoder) and mlx_audio/codec (Mimi), gfx950.  All HBM-bound, one pass each:
//
//   rmsnorm          nn.RMSNorm (talker.py:366-369, llama.py:100-104): one wave per row, x * rsqrt(mean(x^2) + eps) * w
//   head_norm_rope   per-head q/k RMSNorm (talker.py:264-266, 541-542) fused with the rotary embedding
//                    (rotate-half: talker.py:14-36; interleaved pairs = nn.RoPE(traditional=True) in Mimi,
//                    codec/models/mimi/modules/transformer.py:75-77, and CSM's Llama-3 RoPE, sesame/attention.py:41-105),
//                    reading cos / sin from host-built tables so the angles are the reference's own float32 values;
//                    the output pointer is separate, so K is rotated straight into its KV-cache slot
//   swiglu           silu(gate) * up on interleaved (gate, up) columns (talker.py:312-330 TalkerMLP)
//   embed_sum        sum over slots of embedding rows (+ optional text row): the 16 codec embeddings of a Qwen3 frame
//                    (qwen3_tts.py:985-1015), the 32 audio-codebook embeddings of CSM (sesame.py:361-404) and RVQ decode
//                    (quantization.py:187-191 / speech_tokenizer.py:786-800: sum_q codebook_q[codes])
//   dwconv           depthwise conv1d (ConvNeXt dwconv k7, speech_tokenizer.py ConvNeXtBlock) and depthwise
//                    conv_transpose1d (Mimi upsample, mimi.py:296-320), causal / streaming-trimmed
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- rmsnorm
__global__ __launch_bounds__(256) void rmsnorm_kernel(const mi355_rmsnorm_args a) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * a.L) return;
  const int b = (int)(row / a.L), l = (int)(row - (int64_t)b * a.L);
  const int len = a.lens ? a.lens[b] : a.L;
  if (l >= len) return;
  const float* xr = a.x + (int64_t)b * a.x_bstride + (int64_t)l * a.ldx;
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)l * a.ldy;
  float ss = 0.f;
  for (int c = lane * 4; c < a.C; c += 256) {
    const float4 t = *(const float4*)(xr + c);
    ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
  }
  const float r = rsqrtf(wave_sum(ss) / (float)a.C + a.eps);
  for (int c = lane * 4; c < a.C; c += 256) {
    const float4 t = *(const float4*)(xr + c);
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.weight) w = *(const float4*)(a.weight + c);
    *(float4*)(yr + c) = make_float4(t.x * r * w.x, t.y * r * w.y, t.z * r * w.z, t.w * r * w.w);
  }
}

// ---------------------------------------------------------------------------------------------- head norm + rope
// one wave per (row, head); dh in {64, 128}: lane p owns the rotation pair p (and p + 64 is not needed: dh/2 <= 64)
__global__ __launch_bounds__(256) void head_norm_rope_kernel(const mi355_head_rope_args a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ht = a.heads + a.heads2;  // second tensor (k heads, written to the KV-cache slot) shares the launch with the first (q heads)
  const int64_t total = (int64_t)a.B * a.L * ht;
  if (wid >= total) return;
  int h = (int)(wid % ht);
  const int64_t row = wid / ht;
  const int b = (int)(row / a.L), l = (int)(row - (int64_t)b * a.L);
  const int len = a.lens ? a.lens[b] : a.L;
  if (l >= len) return;
  const int half = a.dh >> 1;
  const bool second = h >= a.heads;
  if (second) h -= a.heads;
  const float* xr = second ? a.x2 + (int64_t)b * a.x2_bstride + (int64_t)l * a.ldx2 + h * a.dh
                           : a.x + (int64_t)b * a.x_bstride + (int64_t)l * a.ldx + h * a.dh;
  float* yr = second ? a.y2 + (int64_t)b * a.y2_bstride + (int64_t)l * a.ldy2 + h * a.dh
                     : a.y + (int64_t)b * a.y_bstride + (int64_t)l * a.ldy + h * a.dh;
  const float* nw = second ? a.norm_weight2 : a.norm_weight;
  const bool act = lane < half;
  int i0, i1;  // the two elements of this lane's rotation pair
  if (a.rope_mode == 1) { i0 = 2 * lane; i1 = 2 * lane + 1; }   // interleaved (traditional)
  else { i0 = lane; i1 = lane + half; }                          // rotate-half
  float x0 = 0.f, x1 = 0.f;
  if (act) { x0 = xr[i0]; x1 = xr[i1]; }
  if (nw) {
    const float ss = wave_sum_fast(sumsq2(x0, x1));   // every lane of the wave is here; the fused decode-attention prologue reduces the same way
    const float r = rsqrtf(ss / (float)a.dh + a.eps);
    if (act) { x0 = x0 * r * nw[i0]; x1 = x1 * r * nw[i1]; }
  }
  if (a.cos_table && act) {
    int pos = (a.pos ? a.pos[(int64_t)b * a.pos_ld + l] : a.pos0 + l) - (a.pos_sub ? a.pos_sub[b] : 0);
    pos = pos < 0 ? 0 : (pos >= a.rope_rows ? a.rope_rows - 1 : pos);  // left-padding rows sit before position 0; host-known ranges are checked at the entry point
    const float c = a.cos_table[(int64_t)pos * half + lane], s = a.sin_table[(int64_t)pos * half + lane];
    // (x * cos) + (rotate(x) * sin): pair (x0, x1) -> (x0 c - x1 s, x1 c + x0 s), two roundings per term like the reference
    float y0, y1;
    rope_pair(x0, x1, c, s, y0, y1);
    x0 = y0; x1 = y1;
  }
  if (act) { yr[i0] = x0; yr[i1] = x1; }
}

// ---------------------------------------------------------------------------------------------- swiglu
__global__ void swiglu_kernel(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int I) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * I) return;
  const int64_t r = i / I;
  const int c = (int)(i - r * I);
  const float2 gu = *(const float2*)(x + r * ldx + 2 * c);
  y[r * ldy + c] = (gu.x / (1.0f + expf(-gu.x))) * gu.y;
}

// ---------------------------------------------------------------------------------------------- embed_sum
__global__ __launch_bounds__(256) void embed_sum_kernel(const mi355_embed_sum_args a) {
  const int64_t row = blockIdx.x;  // b * L + l
  const int b = (int)(row / a.L), l = (int)(row - (int64_t)b * a.L);
  const int len = a.lens ? a.lens[b] : a.L;
  if (l >= len) return;
  const int32_t* ids = a.ids + (int64_t)b * a.ids_bstride + (int64_t)l * a.ids_ld;
  float* yr = a.y + (int64_t)b * a.y_bstride + (int64_t)l * a.ldy;
  for (int c = threadIdx.x * 4; c < a.C; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.add) {
      const float* ar = a.add + (int64_t)b * a.add_bstride + (int64_t)l * a.add_ld;
      acc = *(const float4*)(ar + c);
    }
    // eight slots at a time: their ids (and slot offsets) first, then their eight table rows, all in flight together and unconditional (a masked
    // slot reads row 0 and adds nothing) -- as `id = ids[q]; if (id >= 0) acc += table[id]` the sum was 2 Q dependent round trips (16 codebooks: ~6 us)
    for (int q0 = 0; q0 < a.Q; q0 += 8) {
      int id[8], so[8];
      const bool has_so = a.slot_offset != nullptr;
      const int32_t* const sop = has_so ? a.slot_offset : ids;   // absent: any valid int32 (dropped by the select below)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + u < a.Q ? q0 + u : a.Q - 1;
        id[u] = ids[(int64_t)q * a.ids_qstride];
        so[u] = sop[has_so ? q : 0];
      }
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = id[u] >= 0 ? (int64_t)(has_so ? so[u] : 0) + id[u] : 0;
        t[u] = *(const float4*)(a.table + r * a.ld_table + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < a.Q && id[u] >= 0) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }   // slot order: the reference's sum order
      }
    }
    *(float4*)(yr + c) = make_float4(acc.x * a.scale, acc.y * a.scale, acc.z * a.scale, acc.w * a.scale);
  }
}

// ---------------------------------------------------------------------------------------------- depthwise conv / convT
// y[b, t, c] = bias[c] + sum_k w[c, k] * snake(x[b, t + k*dil - pad, c])               (transpose == 0; snake optional)
// y[b, n, c] = bias[c] + sum_{t, k: t*stride + k - pad == n} w[c, k] * x[b, t, c]    (transpose == 1)
__global__ void dwconv_kernel(const mi355_dwconv_args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.B * a.Lout * a.C;
  if (i >= total) return;
  const int c = (int)(i % a.C);
  const int64_t r = i / a.C;
  const int n = (int)(r % a.Lout), b = (int)(r / a.Lout);
  const int len_in = a.lens_in ? a.lens_in[b] : a.Lin;
  const float* xb = a.x + (int64_t)b * a.x_bstride;
  float acc = a.bias ? a.bias[c] : 0.f;
  if (!a.transpose) {
    const int dil = a.dil > 0 ? a.dil : 1;
    const float al = a.pre_alpha ? a.pre_alpha[c] : 0.f, inv = a.pre_alpha ? a.pre_inv[c] : 0.f;
    for (int k = 0; k < a.K; ++k) {
      const int t = n + k * dil - a.pad;
      if (t >= 0 && t < len_in) {
        float v = xb[(int64_t)t * a.ldx + c];
        if (a.pre_alpha) { const float sn = sinf(al * v); v = v + inv * (sn * sn); }  // Snake of the input element (zero padding stays zero)
        acc = fmaf(a.w[c * a.K + k], v, acc);
      }
    }
  } else {
    for (int k = 0; k < a.K; ++k) {
      const int u = n + a.pad - k;
      if (u < 0 || u % a.stride) continue;
      const int t = u / a.stride;
      if (t < len_in) acc = fmaf(a.w[c * a.K + k], xb[(int64_t)t * a.ldx + c], acc);
    }
  }
  a.y[(int64_t)b * a.y_bstride + (int64_t)n * a.ldy + c] = acc;
}

// Depthwise k = 7 conv, one "dilated comb" per thread: the J outputs n0 + j * dil (j < J) of one channel share J + 6 input samples, so the
// optional Snake prologue (v_sin_f32 via __sinf, like conv_gemm's) runs 1.75x per output instead of 7x and every input is loaded once per comb.
// Threads are adjacent in the channel axis (coalesced 256-byte rows).  SNAC's ResidualUnit convs (dilation 1 / 3 / 9) and ConvNeXt's dwconv.
template <int J>
__global__ __launch_bounds__(256) void dwconv7_comb_kernel(const mi355_dwconv_args a, const int dil, const int ncomb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.B * ncomb * dil * a.C;
  if (i >= total) return;
  const int c = (int)(i % a.C);
  int64_t r = i / a.C;
  const int ph = (int)(r % dil);
  r /= dil;
  const int q = (int)(r % ncomb), b = (int)(r / ncomb);
  const int len_in = a.lens_in ? a.lens_in[b] : a.Lin;
  const float* xb = a.x + (int64_t)b * a.x_bstride + c;
  const int n0 = q * J * dil + ph;            // first output of the comb
  const float al = a.pre_alpha ? a.pre_alpha[c] : 0.f, inv = a.pre_alpha ? a.pre_inv[c] : 0.f;
  float w[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) w[k] = a.w[c * 7 + k];
  float v[J + 6];
#pragma unroll
  for (int e = 0; e < J + 6; ++e) {
    const int t = n0 + e * dil - a.pad;
    float u = (t >= 0 && t < len_in) ? xb[(int64_t)t * a.ldx] : 0.f;
    if (a.pre_alpha) { const float sn = __sinf(al * u); u = u + inv * (sn * sn); }
    v[e] = u;
  }
  const float bias = a.bias ? a.bias[c] : 0.f;
  float* yb = a.y + (int64_t)b * a.y_bstride + c;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int n = n0 + j * dil;
    if (n >= a.Lout) break;
    float acc = bias;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = fmaf(w[k], v[j + k], acc);
    yb[(int64_t)n * a.ldy] = acc;
  }
}

}  // namespace

extern "C" int mi355_rmsnorm(const mi355_rmsnorm_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y, "rmsnorm: null tensor");
  const mi355_rmsnorm_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.C > 0 && a.C % 4 == 0, "rmsnorm: C must be a positive multiple of 4");
  MI355_REQUIRE(a.ldx % 4 == 0 && a.ldy % 4 == 0 && a.x_bstride % 4 == 0 && a.y_bstride % 4 == 0, "rmsnorm: strides must be multiples of 4");
  MI355_CLEAR_ERROR();
  const int64_t rows = (int64_t)a.B * a.L;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("rmsnorm");
  return MI355_OK;
}

extern "C" int mi355_head_norm_rope(const mi355_head_rope_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->y, "head_norm_rope: null tensor");
  const mi355_head_rope_args a = *ap;
  MI355_REQUIRE(a.dh == 64 || a.dh == 128, "head_norm_rope: head dim must be 64 or 128");
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.heads > 0, "head_norm_rope: bad shape");
  MI355_REQUIRE((a.cos_table == nullptr) == (a.sin_table == nullptr), "head_norm_rope: cos / sin tables come together");
  MI355_REQUIRE(a.rope_mode == 0 || a.rope_mode == 1, "head_norm_rope: rope_mode must be 0 (rotate-half) or 1 (interleaved)");
  MI355_CLEAR_ERROR();
  MI355_REQUIRE(a.heads2 >= 0 && (a.heads2 == 0 || (a.x2 && a.y2)), "head_norm_rope: second tensor needs x2 / y2");
  MI355_REQUIRE(!a.cos_table || a.rope_rows > 0, "head_norm_rope: rope_rows (rows of the cos / sin tables) must be set");
  MI355_REQUIRE(!a.cos_table || a.pos || a.pos0 + a.L <= a.rope_rows, "head_norm_rope: positions %d..%d run past the %d-row rotary tables", a.pos0,
                a.pos0 + a.L - 1, a.rope_rows);
  const int64_t waves = (int64_t)a.B * a.L * (a.heads + a.heads2);
  hipLaunchKernelGGL(head_norm_rope_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("head_norm_rope");
  return MI355_OK;
}

extern "C" int mi355_swiglu(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t I, void* stream) {
  MI355_REQUIRE(x && y && rows > 0 && I > 0 && ldx % 2 == 0 && ((uintptr_t)x) % 8 == 0, "swiglu: bad arguments");
  MI355_CLEAR_ERROR();
  const int64_t n = rows * I;
  hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, (int)I);
  MI355_LAUNCH_CHECK("swiglu");
  return MI355_OK;
}

extern "C" int mi355_embed_sum(const mi355_embed_sum_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->table && ap->ids && ap->y, "embed_sum: null tensor");
  mi355_embed_sum_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.L > 0 && a.Q > 0 && a.C > 0 && a.C % 4 == 0 && a.ld_table % 4 == 0 && a.ldy % 4 == 0, "embed_sum: bad shape");
  MI355_REQUIRE(!a.add || (a.add_ld % 4 == 0 && a.add_bstride % 4 == 0), "embed_sum: add strides must be multiples of 4");
  if (a.scale == 0.f) a.scale = 1.f;
  MI355_CLEAR_ERROR();
  hipLaunchKernelGGL(embed_sum_kernel, dim3((unsigned)((int64_t)a.B * a.L)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("embed_sum");
  return MI355_OK;
}

extern "C" int mi355_dwconv(const mi355_dwconv_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->x && ap->w && ap->y, "dwconv: null tensor");
  const mi355_dwconv_args a = *ap;
  MI355_REQUIRE(a.B > 0 && a.Lin > 0 && a.Lout > 0 && a.C > 0 && a.K > 0, "dwconv: bad shape");
  MI355_REQUIRE(!a.transpose || a.stride >= 1, "dwconv: transposed conv needs stride >= 1");
  MI355_REQUIRE(!a.transpose || (a.dil <= 1 && !a.pre_alpha), "dwconv: dilation / Snake prologue exist for the plain depthwise conv only");
  MI355_REQUIRE(!a.pre_alpha || a.pre_inv, "dwconv: pre_alpha without pre_inv");
  MI355_CLEAR_ERROR();
  if (!a.transpose && a.K == 7 && a.stride <= 1) {  // the k = 7 depthwise convs of ConvNeXt / SNAC: one dilated comb of 8 outputs per thread
    constexpr int J = 8;
    const int dil = a.dil > 0 ? a.dil : 1;
    const int ncomb = (a.Lout + J * dil - 1) / (J * dil);
    const int64_t nt = (int64_t)a.B * ncomb * dil * a.C;
    hipLaunchKernelGGL(dwconv7_comb_kernel<J>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, dil, ncomb);
    MI355_LAUNCH_CHECK("dwconv(comb)");
    return MI355_OK;
  }
  const int64_t n = (int64_t)a.B * a.Lout * a.C;
  hipLaunchKernelGGL(dwconv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  MI355_LAUNCH_CHECK("dwconv");
  return MI355_OK;
}
