// Native decode-step runner for the decoder-only transformer stacks (host side of libmi355audio.so, gfx950).
//
// One call = one single-position step of a whole stack (all layers) for up to 64 sequences (1..8: the GEMV kernels, 9..64: gemm_rows.hip): every kernel of the step is launched from
// this C++ loop instead of from Python.  The reference drives the same work op by op from Python on MLX's lazy graph
// (tts/models/qwen3_tts/talker.py:385-500 TalkerDecoderLayer / Qwen3TTSTalkerModel.__call__, lm/models/llama.py:160-198,
// stt/models/whisper/whisper.py:405-416, 476-498 ResidualAttentionBlock / TextDecoder with the KV cache); at 100-800 kernels per generated
// frame the per-launch cost of a Python / ctypes call (~15 us) was the whole frame time, so the schedule itself had to become native.
//
// Per layer: [pre-norm + q|k|v GEMV, k|v straight into the KV-cache slot] -> [per-head RMSNorm + RoPE of q and k, one launch] ->
// [KV-streaming attention] -> [o-proj GEMV + LayerScale + residual] -> (cross-attention: [pre-norm + q GEMV] -> [attention over the
// precomputed cross K|V] -> [o-proj GEMV + residual]) -> [pre-norm + up GEMV with fused SwiGLU / GELU] -> [down GEMV + LayerScale + residual].
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

thread_local int t_w_policy = 0;   // mi355_stack_desc.w_policy of the step being enqueued

int gemv_call(const float* x, int ldx, int M, int K, const uint16_t* w, int N, int wdtype, const float* bias, int act, const float* colscale,
              const float* res, int ldr, int glu, float* y, int ldy, int norm, const float* nw, const float* nb, float eps, float* y2, int ldy2,
              int split, void* stream, const float* wscale = nullptr, const float* rope_cos = nullptr, const float* rope_sin = nullptr, int rope_dh = 0,
              int rope_cols = 0, int y2_dtype = MI355_KV_F32, float* split_ws = nullptr, int32_t* split_cnt = nullptr) {
  mi355_gemv_args g;
  memset(&g, 0, sizeof(g));
  g.x = x; g.ldx = ldx; g.M = M; g.K = K; g.w = w; g.ldw = K; g.wdtype = wdtype; g.N = N; g.bias = bias; g.post_act = act; g.colscale = colscale;
  g.res = res; g.ldr = ldr; g.out_scale = 1.f; g.glu = glu; g.y = y; g.ldy = ldy; g.norm = norm; g.norm_weight = nw; g.norm_bias = nb;
  g.norm_eps = eps; g.y2 = y2; g.ldy2 = ldy2; g.split = split; g.wscale = wscale;
  g.rope_cos = rope_cos; g.rope_sin = rope_sin; g.rope_dh = rope_dh; g.rope_cols = rope_cols; g.y2_dtype = y2_dtype;
  g.split_ws = split_ws; g.split_cnt = split_cnt;
  g.w_policy = t_w_policy;
  return mi355_gemv(&g, stream);
}

int attn_call(const float* q, int ldq, const float* k, const float* v, int64_t kv_bstride, int ldkv, int heads, int kv_heads, int dh, int Tk,
              int causal, int window, float scale, int B, float* out, int ldo, void* stream, int64_t hstride = 0, float* split_ws = nullptr,
              int32_t* split_cnt = nullptr, const int32_t* k_start = nullptr, int kv_dtype = MI355_KV_F32) {
  mi355_flash_attn_args a;
  memset(&a, 0, sizeof(a));
  a.k_hstride = hstride; a.v_hstride = hstride; a.split_ws = split_ws; a.split_cnt = split_cnt;
  a.q = q; a.q_bstride = ldq; a.ldq = ldq; a.k = k; a.k_bstride = kv_bstride; a.ldk = ldkv; a.v = v; a.v_bstride = kv_bstride; a.ldv = ldkv;
  a.heads = heads; a.kv_heads = kv_heads; a.dh = dh; a.Tq = 1; a.Tk = Tk; a.causal = causal; a.window = window; a.scale = scale; a.B = B;
  a.mode = 2; a.out = out; a.out_bstride = ldo; a.ldo = ldo; a.k_start = k_start; a.kv_dtype = kv_dtype;
  return mi355_flash_attention(&a, stream);
}


// ---------------------------------------------------------------------------------------------- steps for 9..64 sequences: the rows pipeline (rows_pipe.hip)
struct RowsWs {
  uint16_t* px; uint16_t* pa; uint16_t* pm; float* part; int64_t part_floats; int64_t bytes;
};

int rows_R(int B) { return B <= 16 ? 16 : (B <= 32 ? 32 : 64); }
int round8(int n) { return (n + 7) / 8 * 8; }

RowsWs rows_layout(const mi355_stack_desc& d, int B, void* base) {
  const int R = rows_R(B), D = d.d_model, nq = d.heads * d.dh, nkv = 2 * d.kv_heads * d.dh, n_in = d.glu ? 2 * d.d_ff : d.d_ff;
  RowsWs w;
  w.px = (uint16_t*)base;
  w.pa = w.px + (int64_t)2 * R * D;
  w.pm = w.pa + (int64_t)2 * R * nq;
  w.part = (float*)(w.pm + (int64_t)2 * R * d.d_ff);
  const int64_t f_qkv = (int64_t)mi355_rows_kgroups(nq + nkv, D) * B * round8(nq + nkv), f_o = (int64_t)mi355_rows_kgroups(D, nq) * B * round8(D);
  const int64_t f_in = (int64_t)mi355_rows_kgroups(n_in, D) * B * round8(n_in), f_out = (int64_t)mi355_rows_kgroups(D, d.d_ff) * B * round8(D);
  w.part_floats = f_qkv > f_o ? f_qkv : f_o;
  if (f_in > w.part_floats) w.part_floats = f_in;
  if (f_out > w.part_floats) w.part_floats = f_out;
  w.bytes = ((char*)w.part - (char*)base) + w.part_floats * 4;
  return w;
}

int rows_gemm_call(const uint16_t* planes, const uint16_t* wt, int wdtype, int N, int K, int B, int R, float* part, int* kg, void* stream) {
  mi355_rows_gemm_args g;
  memset(&g, 0, sizeof(g));
  *kg = mi355_rows_kgroups(N, K);
  g.wt = wt; g.wdtype = wdtype; g.N = N; g.K = K; g.planes = planes; g.M = B; g.R = R; g.part = part; g.ldp = round8(N);
  g.kg_stride = (int64_t)B * g.ldp; g.kgroups = *kg;
  return mi355_rows_gemm(&g, stream);
}

int tall_step(const mi355_stack_desc& d, float* x, int B, int offset, float* ws, float* out, void* stream) {
  MI355_REQUIRE(d.d_model % 64 == 0 && d.d_ff % 64 == 0 && (d.heads * d.dh) % 64 == 0, "stack_decode_step: 9..64 sequences per step need widths that are multiples of 64");
  MI355_REQUIRE(d.rows_ws && ((uintptr_t)d.rows_ws) % 16 == 0, "stack_decode_step: 9..64 sequences per step need the rows workspace (mi355_stack_rows_ws_bytes)");
  const int D = d.d_model, H = d.heads, G = d.kv_heads, dh = d.dh, R = rows_R(B);
  const int nq = H * dh, nkv = 2 * G * dh, n_in = d.glu ? 2 * d.d_ff : d.d_ff;
  const RowsWs w = rows_layout(d, B, d.rows_ws);
  const int pdt = d.wdtype == MI355_W_F16 ? MI355_W_F16 : MI355_W_BF16;   // element type of the planes (an fp8 image is decoded to bf16)
  const bool fp8 = d.wdtype == MI355_W_FP8;
  MI355_REQUIRE(w.bytes <= d.rows_ws_bytes, "stack_decode_step: rows workspace too small (%lld bytes, need %lld)", (long long)d.rows_ws_bytes, (long long)w.bytes);
  float* q = ws;                       // [B, nq]
  float* att = q + (size_t)B * nq;     // [B, nq]
  const float scale = d.attn_scale > 0.f ? d.attn_scale : 1.0f / sqrtf((float)dh);
  const int kvsz = d.kv_dtype == MI355_KV_F32 ? 4 : 2;
  auto finish = [&](mi355_rows_finish_args& f, int N, int kg) {
    f.part = w.part; f.kgroups = kg; f.ldp = round8(N); f.kg_stride = kg > 1 ? (int64_t)B * f.ldp : 0; f.M = B; f.N = N; f.out_scale = 1.f;
    f.R = R; f.planes_dtype = pdt;
    return mi355_rows_finish(&f, stream);
  };
  int rc, kg;
  {  // the step input as planes, normalised by layer 0's attention norm
    mi355_rows_finish_args f;
    memset(&f, 0, sizeof(f));
    f.part = x; f.kgroups = 1; f.ldp = D; f.M = B; f.N = D; f.out_scale = 1.f; f.norm = d.norm; f.norm_weight = d.layers[0].attn_norm_w;
    f.norm_bias = d.layers[0].attn_norm_b; f.norm_eps = d.eps; f.planes = w.px; f.R = R; f.planes_dtype = pdt;
    rc = mi355_rows_finish(&f, stream);
    if (rc) return rc;
  }
  for (int i = 0; i < d.n_layers; ++i) {
    const mi355_layer_desc& L = d.layers[i];
    MI355_REQUIRE(L.wqkv_t && L.wo_t && L.w_in_t && L.w_out_t && L.kv, "stack_decode_step: layer %d has no tile images (steps of 9..64 sequences)", i);
    MI355_REQUIRE(!L.cross_k || (L.wcq_t && L.wco_t && L.cross_v && L.cross_len > 0),
                  "stack_decode_step: layer %d has cross K but no tile images of its cross projections / no V (steps of 9..64 sequences)", i);
    MI355_REQUIRE(offset < L.kv_capacity, "stack_decode_step: KV cache of layer %d is full (offset %d, capacity %d)", i, offset, L.kv_capacity);
    float* slot = (float*)((char*)L.kv + (int64_t)offset * nkv * kvsz);
    const float* vbase = (const float*)((const char*)L.kv + (int64_t)G * dh * kvsz);
    // causal stacks take the fused attention step (slab sums + bias + optional per-head norms / rotary embedding + cache store inside the attention
    // launch, output as planes); a stack without norms / rotary embedding (Whisper's decoder) simply has nothing applied there
    const bool rope_in_attn = d.causal != 0;
    MI355_REQUIRE(d.kv_dtype == MI355_KV_F32 || rope_in_attn || !(L.q_norm || d.cos),
                  "stack_decode_step: a 16-bit KV cache needs the rotary embedding inside the attention step");
    MI355_REQUIRE(!d.slot_lens_k || rope_in_attn, "stack_decode_step: slot caches need the fused attention step (per-head norms / rotary embedding, causal)");
    // ---- q | k | v
    rc = rows_gemm_call(w.px, L.wqkv_t, d.wdtype, nq + nkv, D, B, R, w.part, &kg, stream);
    if (rc) return rc;
    if (rope_in_attn) {
      // the fused attention step adds the K-group slabs of its own head itself (+ bias), applies the per-head norms / rotary embedding, files k, v
      // into the cache and leaves its output row as planes for the o-proj GEMM: no row epilogue and no converter launch around it
      const int ld = round8(nq + nkv);
      mi355_flash_attn_args a;
      memset(&a, 0, sizeof(a));
      a.q = w.part; a.q_bstride = ld; a.ldq = ld; a.k = L.kv; a.k_bstride = L.kv_bstride; a.ldk = nkv; a.v = vbase; a.v_bstride = L.kv_bstride; a.ldv = nkv;
      a.kv_dtype = d.kv_dtype;
      a.heads = H; a.kv_heads = G; a.dh = dh; a.Tq = 1; a.Tk = offset + 1; a.causal = 1; a.window = d.window; a.scale = scale; a.B = B; a.mode = 2;
      a.out_planes = w.pa; a.planes_R = R; a.planes_dtype = pdt; a.out_bstride = nq; a.ldo = nq; a.k_start = d.k_start; a.nsplit = 1;
      if (fp8) { a.q_wscale = L.s_qkv; a.k_wscale = L.s_qkv + nq; a.v_wscale = L.s_qkv + nq + G * dh; }
      a.lens_k = d.slot_lens_k;
      a.new_k = w.part + nq; a.new_v = w.part + nq + G * dh; a.new_bstride = ld;
      a.in_kgroups = kg; a.in_kg_stride = (int64_t)B * ld;
      if (L.bqkv) { a.q_bias = L.bqkv; a.k_bias = L.bqkv + nq; a.v_bias = L.bqkv + nq + G * dh; }
      a.q_norm_w = L.q_norm; a.k_norm_w = L.k_norm; a.norm_eps = d.eps;
      a.rope_cos = d.cos; a.rope_sin = d.sin; a.rope_rows = d.rope_rows; a.rope_mode = d.rope_mode; a.rope_pos = offset;
      rc = mi355_flash_attention(&a, stream);
      if (rc) return rc;
    } else {
      {
        mi355_rows_finish_args f;
        memset(&f, 0, sizeof(f));
        f.bias = L.bqkv; f.y = q; f.ldy = nq; f.split = nq; f.y2 = slot; f.ldy2 = (int)L.kv_bstride; f.y2_dtype = d.kv_dtype; f.wscale = fp8 ? L.s_qkv : nullptr;
        rc = finish(f, nq + nkv, kg);
        if (rc) return rc;
      }
      if (L.q_norm || d.cos) {
        mi355_head_rope_args r;
        memset(&r, 0, sizeof(r));
        r.x = q; r.x_bstride = nq; r.ldx = nq; r.heads = H; r.dh = dh; r.L = 1; r.B = B; r.norm_weight = L.q_norm; r.eps = d.eps;
        r.cos_table = d.cos; r.sin_table = d.sin; r.pos0 = offset; r.pos_sub = d.k_start; r.rope_rows = d.rope_rows; r.rope_mode = d.rope_mode;
        r.y = q; r.y_bstride = nq; r.ldy = nq;
        r.x2 = slot; r.x2_bstride = L.kv_bstride; r.ldx2 = nkv; r.heads2 = G; r.norm_weight2 = L.k_norm; r.y2 = slot; r.y2_bstride = L.kv_bstride;
        r.ldy2 = nkv;
        rc = mi355_head_norm_rope(&r, stream);
        if (rc) return rc;
      }
      rc = attn_call(q, nq, L.kv, vbase, L.kv_bstride, nkv, H, G, dh, offset + 1, d.causal, d.window, scale, B, att, nq, stream, 0, nullptr, nullptr, d.k_start,
                     d.kv_dtype);
      if (rc) return rc;
      mi355_rows_finish_args f;   // attention output -> planes
      memset(&f, 0, sizeof(f));
      f.part = att; f.kgroups = 1; f.ldp = nq; f.M = B; f.N = nq; f.out_scale = 1.f; f.planes = w.pa; f.R = R; f.planes_dtype = pdt;
      rc = mi355_rows_finish(&f, stream);
      if (rc) return rc;
    }
    // ---- o-proj + LayerScale + residual; the new residual stream also leaves as planes normalised for the MLP
    rc = rows_gemm_call(w.pa, L.wo_t, d.wdtype, D, nq, B, R, w.part, &kg, stream);
    if (rc) return rc;
    {
      mi355_rows_finish_args f;
      memset(&f, 0, sizeof(f));
      f.bias = L.bo; f.colscale = L.ls1; f.res = x; f.ldr = D; f.y = x; f.ldy = D; f.wscale = fp8 ? L.s_o : nullptr;
      f.norm = d.norm; f.norm_eps = d.eps; f.planes = w.px;
      f.norm_weight = L.cross_k ? L.cross_norm_w : L.mlp_norm_w;   // the planes feed the cross-attention's q projection when there is one
      f.norm_bias = L.cross_k ? L.cross_norm_b : L.mlp_norm_b;
      rc = finish(f, D, kg);
      if (rc) return rc;
    }
    // ---- cross-attention (Whisper decoder): K | V precomputed once per window; q projection, attention over all cross_len keys, o-proj + residual
    if (L.cross_k) {
      rc = rows_gemm_call(w.px, L.wcq_t, d.wdtype, nq, D, B, R, w.part, &kg, stream);
      if (rc) return rc;
      {
        mi355_rows_finish_args f;
        memset(&f, 0, sizeof(f));
        f.bias = L.bcq; f.y = q; f.ldy = nq; f.wscale = fp8 ? L.s_cq : nullptr;
        rc = finish(f, nq, kg);
        if (rc) return rc;
      }
      {  // >= 5 x heads (query, head) items: no key split; the output row leaves as planes for the o-proj GEMM (no converter launch)
        mi355_flash_attn_args a;
        memset(&a, 0, sizeof(a));
        a.k_hstride = L.cross_hstride; a.v_hstride = L.cross_hstride;
        a.q = q; a.q_bstride = nq; a.ldq = nq; a.k = L.cross_k; a.k_bstride = L.cross_bstride; a.ldk = L.cross_ld; a.v = L.cross_v; a.v_bstride = L.cross_bstride;
        a.ldv = L.cross_ld; a.heads = H; a.kv_heads = G; a.dh = dh; a.Tq = 1; a.Tk = L.cross_len; a.scale = scale; a.B = B; a.mode = 2; a.nsplit = 1;
        a.out_planes = w.pa; a.planes_R = R; a.planes_dtype = pdt; a.out_bstride = nq; a.ldo = nq; a.kv_dtype = L.cross_kv_dtype;
        rc = mi355_flash_attention(&a, stream);
        if (rc) return rc;
      }
      rc = rows_gemm_call(w.pa, L.wco_t, d.wdtype, D, nq, B, R, w.part, &kg, stream);
      if (rc) return rc;
      {
        mi355_rows_finish_args f;
        memset(&f, 0, sizeof(f));
        f.bias = L.bco; f.res = x; f.ldr = D; f.y = x; f.ldy = D; f.wscale = fp8 ? L.s_co : nullptr;
        f.norm = d.norm; f.norm_weight = L.mlp_norm_w; f.norm_bias = L.mlp_norm_b; f.norm_eps = d.eps; f.planes = w.px;
        rc = finish(f, D, kg);
        if (rc) return rc;
      }
    }
    // ---- MLP
    // SwiGLU in the GEMM's own epilogue (one K group: a workgroup holds complete sums) saves the row-epilogue launch; worth it whenever the column
    // tiles alone fill the chip (n_in >= 4096: >= 128 workgroups of two tiles) -- measured on the code predictor's 6144 x 1024 image (call 7)
    if (d.glu && n_in % 128 == 0 && (mi355_rows_kgroups(n_in, D) == 1 || n_in >= 4096)) {
      mi355_rows_gemm_args g;
      memset(&g, 0, sizeof(g));
      g.wt = L.w_in_t; g.wdtype = d.wdtype; g.N = n_in; g.K = D; g.planes = w.px; g.M = B; g.R = R; g.kgroups = 1;
      g.glu_planes_out = w.pm; g.glu_bias = L.b_in; g.wscale = fp8 ? L.s_in : nullptr;
      rc = mi355_rows_gemm(&g, stream);
      if (rc) return rc;
    } else {
    rc = rows_gemm_call(w.px, L.w_in_t, d.wdtype, n_in, D, B, R, w.part, &kg, stream);
    if (rc) return rc;
    {
      mi355_rows_finish_args f;
      memset(&f, 0, sizeof(f));
      f.bias = L.b_in; f.glu = d.glu; f.post_act = d.glu ? MI355_ACT_NONE : d.act; f.planes = w.pm; f.wscale = fp8 ? L.s_in : nullptr;
      rc = finish(f, n_in, kg);
      if (rc) return rc;
    }
    }
    rc = rows_gemm_call(w.pm, L.w_out_t, d.wdtype, D, d.d_ff, B, R, w.part, &kg, stream);
    if (rc) return rc;
    {
      mi355_rows_finish_args f;
      memset(&f, 0, sizeof(f));
      f.bias = L.b_out; f.colscale = L.ls2; f.res = x; f.ldr = D; f.y = x; f.ldy = D; f.wscale = fp8 ? L.s_out : nullptr;
      if (i + 1 < d.n_layers) {   // the next layer's input planes
        f.norm = d.norm; f.norm_weight = d.layers[i + 1].attn_norm_w; f.norm_bias = d.layers[i + 1].attn_norm_b; f.norm_eps = d.eps; f.planes = w.px;
      } else if (out && d.final_norm_w) {
        f.norm = d.norm; f.norm_weight = d.final_norm_w; f.norm_bias = d.final_norm_b; f.norm_eps = d.eps; f.yn = out; f.ldyn = D;
      }
      rc = finish(f, D, kg);
      if (rc) return rc;
    }
  }
  return MI355_OK;
}

}  // namespace

extern "C" int64_t mi355_stack_rows_ws_bytes(const mi355_stack_desc* dp, int32_t B) {
  if (!dp || B < 1 || B > 64 || dp->d_model % 64 || dp->d_ff % 64 || (dp->heads * dp->dh) % 64) return 0;
  return rows_layout(*dp, B, nullptr).bytes;
}

extern "C" int mi355_stack_decode_step(const mi355_stack_desc* dp, float* x, int32_t B, int32_t offset, float* ws, float* out, void* stream) {
  MI355_REQUIRE(dp && x && ws && dp->layers, "stack_decode_step: null argument");
  const mi355_stack_desc d = *dp;
  MI355_REQUIRE(B >= 1 && B <= 64, "stack_decode_step: 1..64 sequences per step (got %d)", B);
  MI355_REQUIRE(B <= 8 || (d.d_model % 64 == 0 && d.d_ff % 64 == 0 && (d.heads * d.dh) % 64 == 0),
                "stack_decode_step: 9..64 sequences per step need widths that are multiples of 64");
  static const bool rows_off = getenv("MI355_ROWS_PIPE") != nullptr && getenv("MI355_ROWS_PIPE")[0] == '0';   // A/B knob: 9..64 rows through mi355_gemv (gemm_rows.hip)
  MI355_REQUIRE(d.n_layers > 0 && d.d_model % 8 == 0 && d.d_ff % 8 == 0 && (d.dh == 64 || d.dh == 128), "stack_decode_step: bad dimensions");
  MI355_REQUIRE(d.wdtype != MI355_W_FP8 || (d.d_model % 16 == 0 && d.d_ff % 16 == 0), "stack_decode_step: fp8 images need d_model, d_ff multiples of 16");
  MI355_REQUIRE(d.norm == 1 || d.norm == 2, "stack_decode_step: norm must be 1 (LayerNorm) or 2 (RMSNorm)");
  MI355_REQUIRE(offset >= 0, "stack_decode_step: negative offset");
  MI355_REQUIRE(d.kv_dtype >= MI355_KV_F32 && d.kv_dtype <= MI355_KV_F16, "stack_decode_step: bad kv_dtype");
  MI355_REQUIRE(!d.cos || (d.rope_rows > 0 && offset < d.rope_rows), "stack_decode_step: position %d is past the %d-row rotary tables", offset,
                d.rope_rows);
  // the rows pipeline also serves 5..8 sequences (16-row planes): measured faster than the 5..8-row matrix-pipe GEMV it replaces there (MI355_ROWS_MIN=9: old split)
  static const int rows_min = getenv("MI355_ROWS_MIN") ? atoi(getenv("MI355_ROWS_MIN")) : 5;
  if ((B > 8 || B >= rows_min) && !rows_off && d.layers[0].wqkv_t && (!d.layers[0].cross_k || d.layers[0].wcq_t)) return tall_step(d, x, B, offset, ws, out, stream);
  const int D = d.d_model, H = d.heads, G = d.kv_heads, dh = d.dh;
  const int nq = H * dh, nkv = 2 * G * dh;
  MI355_REQUIRE(d.w_policy == 0 || d.w_policy == 1, "stack_decode_step: w_policy must be 0 or 1");
  t_w_policy = d.w_policy;
  float* q = ws;                 // [B, nq]
  float* att = q + (size_t)B * nq;   // [B, nq]
  float* mid = att + (size_t)B * nq; // [B, d_ff]
  const float scale = d.attn_scale > 0.f ? d.attn_scale : 1.0f / sqrtf((float)dh);
  for (int i = 0; i < d.n_layers; ++i) {
    const mi355_layer_desc& L = d.layers[i];
    MI355_REQUIRE(L.wqkv && L.wo && L.w_in && L.w_out && L.kv, "stack_decode_step: layer %d is missing a tensor", i);
    MI355_REQUIRE(d.wdtype != MI355_W_FP8 || (L.s_qkv && L.s_o && L.s_in && L.s_out && (!L.cross_k || (L.s_cq && L.s_co))),
                  "stack_decode_step: layer %d has fp8 images but no scales", i);
    MI355_REQUIRE(offset < L.kv_capacity, "stack_decode_step: KV cache of layer %d is full (offset %d, capacity %d)", i, offset, L.kv_capacity);
    const int kvsz = d.kv_dtype == MI355_KV_F32 ? 4 : 2;   // bytes per cache element
    float* slot = (float*)((char*)L.kv + (int64_t)offset * nkv * kvsz);  // row `offset` of item 0; items are kv_bstride elements apart
    const float* vbase = (const float*)((const char*)L.kv + (int64_t)G * dh * kvsz);  // the v columns of row 0
    // ---- self-attention
    // interleaved RoPE without per-head q / k norms (CSM Llama, Mimi): the rotation of a pair (2i, 2i + 1) is the epilogue of the wave that owns
    // those two columns of the q | k | v projection -- one launch less per layer (MI355_GEMV_ROPE=0 keeps the separate kernel for A/B runs)
    static const bool rope_off = getenv("MI355_GEMV_ROPE") != nullptr && getenv("MI355_GEMV_ROPE")[0] == '0';
    const bool rope_in_gemv = d.cos && d.rope_mode == 1 && !L.q_norm && !L.k_norm && B <= 4 && !rope_off && !d.k_start && !d.slot_lens_k;  // one table row for all rows
    const float* rc_row = rope_in_gemv ? d.cos + (int64_t)offset * (dh / 2) : nullptr;
    const float* rs_row = rope_in_gemv ? d.sin + (int64_t)offset * (dh / 2) : nullptr;
    // per-head q / k norms and / or rotate-half RoPE (Qwen3 talker / code predictor / codec transformer): the attention kernel applies them to q
    // and to the new k itself and files k, v into the cache (mi355_flash_attn_args.new_k) -- the raw k | v of this step go to a scratch row (the
    // tail of the workspace) instead of the cache slot.  MI355_ATTN_FUSE_ROPE=0 keeps the separate head_norm_rope launch.
    static const bool fuse_off = getenv("MI355_ATTN_FUSE_ROPE") != nullptr && getenv("MI355_ATTN_FUSE_ROPE")[0] == '0';
    const bool rope_in_attn = !rope_in_gemv && (L.q_norm || d.cos) && (!fuse_off || d.kv_dtype != MI355_KV_F32 || d.slot_lens_k) && d.causal;
    MI355_REQUIRE(!d.slot_lens_k || rope_in_attn, "stack_decode_step: slot caches need the fused attention step (per-head norms / rotary embedding, causal)");
    MI355_REQUIRE(d.kv_dtype == MI355_KV_F32 || rope_in_gemv || rope_in_attn || !(L.q_norm || d.cos),
                  "stack_decode_step: a 16-bit KV cache needs the rotary embedding inside the q|k|v GEMV or inside the attention step");
    // one sequence, rotary pairs already applied by the q|k|v GEMV, at most 64 cached positions, no window / left padding: attention as the
    // prologue of the o-proj GEMV
    // Measured (profiles/r4_kernel_stats_csm_attn_prologue_call5.txt): the fused launch takes 13.3 us against 5.7 + 4.4 us for the two it replaces --
    // 256 workgroups re-reading the same 64 KB of K | V at the same moment queue on the same L2 lines -- so it is OPT-IN: MI355_ATTN_IN_OPROJ=1.
    static const bool aio_off = getenv("MI355_ATTN_IN_OPROJ") == nullptr || getenv("MI355_ATTN_IN_OPROJ")[0] != '1';
    const bool attn_in_oproj = !aio_off && B == 1 && (rope_in_gemv || !(L.q_norm || d.cos)) && !rope_in_attn && offset + 1 <= 64 && nq <= 2048 && H <= 32 && d.causal &&
                               !d.window && !d.k_start && !d.slot_lens_k && !L.cross_k;
    float* kvtmp = mid + (size_t)B * d.d_ff;  // [B, nkv]: the tail of the workspace
    int rc = gemv_call(x, D, B, D, L.wqkv, nq + nkv, d.wdtype, L.bqkv, MI355_ACT_NONE, nullptr, nullptr, 0, 0, q, nq, d.norm, L.attn_norm_w,
                       L.attn_norm_b, d.eps, rope_in_attn ? kvtmp : slot, rope_in_attn ? nkv : (int)L.kv_bstride, nq, stream, L.s_qkv, rc_row, rs_row, dh,
                       rope_in_gemv ? nq + G * dh : 0, rope_in_attn ? MI355_KV_F32 : d.kv_dtype);
    if (rc) return rc;
    if (rope_in_attn) {
      mi355_flash_attn_args a;
      memset(&a, 0, sizeof(a));
      a.q = q; a.q_bstride = nq; a.ldq = nq; a.k = L.kv; a.k_bstride = L.kv_bstride; a.ldk = nkv; a.v = vbase; a.v_bstride = L.kv_bstride; a.ldv = nkv;
      a.kv_dtype = d.kv_dtype;
      a.heads = H; a.kv_heads = G; a.dh = dh; a.Tq = 1; a.Tk = offset + 1; a.causal = 1; a.window = d.window; a.scale = scale; a.B = B; a.mode = 2;
      a.out = att; a.out_bstride = nq; a.ldo = nq; a.k_start = d.k_start; a.nsplit = 1; a.lens_k = d.slot_lens_k;
      a.new_k = kvtmp; a.new_v = kvtmp + G * dh; a.new_bstride = nkv; a.q_norm_w = L.q_norm; a.k_norm_w = L.k_norm; a.norm_eps = d.eps;
      a.rope_cos = d.cos; a.rope_sin = d.sin; a.rope_rows = d.rope_rows; a.rope_mode = d.rope_mode; a.rope_pos = offset;
      rc = mi355_flash_attention(&a, stream);
      if (rc) return rc;
    } else if (attn_in_oproj) {
      // short context, one sequence: the attention row is recomputed by every workgroup of the o-proj GEMV as its prologue (mi355_gemv_args.attn_*)
    } else {
    if (!rope_in_gemv && (L.q_norm || d.cos)) {
      mi355_head_rope_args r;
      memset(&r, 0, sizeof(r));
      r.x = q; r.x_bstride = nq; r.ldx = nq; r.heads = H; r.dh = dh; r.L = 1; r.B = B; r.norm_weight = L.q_norm; r.eps = d.eps;
      r.cos_table = d.cos; r.sin_table = d.sin; r.pos0 = offset; r.pos_sub = d.k_start; r.rope_rows = d.rope_rows; r.rope_mode = d.rope_mode;
      r.y = q; r.y_bstride = nq; r.ldy = nq;
      r.x2 = slot; r.x2_bstride = L.kv_bstride; r.ldx2 = nkv; r.heads2 = G; r.norm_weight2 = L.k_norm; r.y2 = slot; r.y2_bstride = L.kv_bstride;
      r.ldy2 = nkv;
      rc = mi355_head_norm_rope(&r, stream);
      if (rc) return rc;
    }
    rc = attn_call(q, nq, L.kv, vbase, L.kv_bstride, nkv, H, G, dh, offset + 1, d.causal, d.window, scale, B, att, nq, stream, 0, d.attn_split_ws,
                   d.attn_split_cnt, d.k_start, d.kv_dtype);
    if (rc) return rc;
    }
    if (attn_in_oproj) {
      mi355_gemv_args g;
      memset(&g, 0, sizeof(g));
      g.x = q; g.ldx = nq; g.M = 1; g.K = nq; g.w = L.wo; g.ldw = nq; g.wdtype = d.wdtype; g.N = D; g.bias = L.bo; g.post_act = MI355_ACT_NONE; g.colscale = L.ls1;
      g.res = x; g.ldr = D; g.out_scale = 1.f; g.y = x; g.ldy = D; g.wscale = L.s_o; g.w_policy = t_w_policy;
      g.attn_k = L.kv; g.attn_v = vbase; g.attn_ld = nkv; g.attn_Tk = offset + 1; g.attn_heads = H; g.attn_kv_heads = G; g.attn_dh = dh; g.attn_scale = scale;
      g.attn_kv_dtype = d.kv_dtype;
      rc = mi355_gemv(&g, stream);
    } else {
      rc = gemv_call(att, nq, B, nq, L.wo, D, d.wdtype, L.bo, MI355_ACT_NONE, L.ls1, x, D, 0, x, D, 0, nullptr, nullptr, 0.f, nullptr, 0, 0, stream, L.s_o);
    }
    if (rc) return rc;
    // ---- cross-attention (Whisper decoder): K | V precomputed once per window
    if (L.cross_k) {
      MI355_REQUIRE(L.wcq && L.wco && L.cross_v && L.cross_len > 0, "stack_decode_step: layer %d has cross K but no cross projections / V", i);
      rc = gemv_call(x, D, B, D, L.wcq, nq, d.wdtype, L.bcq, MI355_ACT_NONE, nullptr, nullptr, 0, 0, q, nq, d.norm, L.cross_norm_w, L.cross_norm_b,
                     d.eps, nullptr, 0, 0, stream, L.s_cq);
      if (rc) return rc;
      rc = attn_call(q, nq, L.cross_k, L.cross_v, L.cross_bstride, L.cross_ld, H, G, dh, L.cross_len, 0, 0, scale, B, att, nq, stream, L.cross_hstride,
                     d.attn_split_ws, d.attn_split_cnt, nullptr, L.cross_kv_dtype);
      if (rc) return rc;
      rc = gemv_call(att, nq, B, nq, L.wco, D, d.wdtype, L.bco, MI355_ACT_NONE, nullptr, x, D, 0, x, D, 0, nullptr, nullptr, 0.f, nullptr, 0, 0, stream,
                     L.s_co);
      if (rc) return rc;
    }
    // ---- MLP
    rc = gemv_call(x, D, B, D, L.w_in, d.glu ? 2 * d.d_ff : d.d_ff, d.wdtype, L.b_in, d.glu ? MI355_ACT_NONE : d.act, nullptr, nullptr, 0, d.glu, mid,
                   d.d_ff, d.norm, L.mlp_norm_w, L.mlp_norm_b, d.eps, nullptr, 0, 0, stream, L.s_in);
    if (rc) return rc;
    rc = gemv_call(mid, d.d_ff, B, d.d_ff, L.w_out, D, d.wdtype, L.b_out, MI355_ACT_NONE, L.ls2, x, D, 0, x, D, 0, nullptr, nullptr, 0.f, nullptr, 0, 0,
                   stream, L.s_out, nullptr, nullptr, 0, 0, MI355_KV_F32, d.gemv_split_ws, d.gemv_split_cnt);
    if (rc) return rc;
  }
  if (out && d.final_norm_w) {
    if (d.norm == 2) {
      mi355_rmsnorm_args n;
      memset(&n, 0, sizeof(n));
      n.x = x; n.x_bstride = D; n.ldx = D; n.C = D; n.L = 1; n.B = B; n.weight = d.final_norm_w; n.eps = d.eps; n.y = out; n.y_bstride = D; n.ldy = D;
      return mi355_rmsnorm(&n, stream);
    }
    mi355_layernorm_args n;
    memset(&n, 0, sizeof(n));
    n.x = x; n.x_bstride = D; n.ldx = D; n.C = D; n.L = 1; n.B = B; n.weight = d.final_norm_w; n.bias = d.final_norm_b; n.eps = d.eps; n.y = out;
    n.y_bstride = D; n.ldy = D;
    return mi355_layernorm(&n, stream);
  }
  return MI355_OK;
}
