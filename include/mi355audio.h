/*
 * mi355audio.h -- C ABI of libmi355audio.so, the MI355X (gfx950 / CDNA4) kernel library
 * under the mlx-audio hot path (Kokoro-82M TTS: PL-BERT, bi-LSTM prosody predictor,
 * AdaIN / iSTFTNet vocoder; mlx_audio.dsp STFT / iSTFT / mel).
 *
 * The reference (Blaizzy/mlx-audio v0.5.0) has no FFI of its own: its operator boundary is the
 * Python `mlx.core` API.  Each entry point below therefore names the
 * `mlx.core` / `mlx.nn` call sites it replaces (file:line relative to the reference tree); the
 * Python layer `mlx_audio_amd` binds them with ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the field name ends in `_host`;
 *   - activations are float32, channels-last: element (b, l, c) of a tensor lives at
 *     base + b*bstride + l*ld + c   (the NLC layout mx.conv1d uses), ld >= C;
 *   - `lens` (nullable) holds the valid row count of every batch item (ragged batches are padded
 *     to a common L; rows >= lens[b] read as zeros and are never written);
 *   - weights are bfloat16 (or float16 for fp16 checkpoints), pre-packed by mi355_pack_* into MFMA fragment order;
 *   - `stream` is a hipStream_t passed as void*; all functions are asynchronous on it;
 *   - return value: 0 = ok, negative = error (text via mi355_last_error()); nothing is allocated,
 *     no global state is kept, distinct streams may be driven from distinct host threads.
 */
#ifndef MI355AUDIO_H
#define MI355AUDIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_ARG (-1)
#define MI355_ERR_LAUNCH (-2)
#define MI355_ERR_UNSUPPORTED (-3)

const char* mi355_last_error(void);
/* ABI version of this header; bumped on any struct change. */
int mi355_abi_version(void);
/* Name/arch/CU count of device `dev` into caller buffers (used by bench.py). */
int mi355_device_info(int dev, char* name, int name_cap, int* cu_count, int* lds_bytes);

/* ------------------------------------------------------------------------------------------
 * conv_gemm: conv1d / linear / polyphase conv_transpose1d as an implicit GEMM on MFMA
 * (v_mfma_f32_32x32x16_bf16; fp32 activations split on the fly into bf16 hi+lo so that the
 * result keeps ~16 mantissa bits, weights bf16; or v_mfma_f32_32x32x16_f16 on fp16-rounded activations).
 * Replaces: mx.conv1d / mx.conv_transpose1d in ConvWeighted (tts/models/kokoro/istftnet.py:
 * 128-170), nn.Conv1d (istftnet.py:771-786), nn.Linear (modules.py:15,60-90,477-600,
 * kokoro.py:84), the per-step x-projection mx.addmm of LSTM (modules.py:156-163), and the fused
 * neighbours the reference runs as separate MLX ops: AdaIN affine + Snake / LeakyReLU in front
 * (istftnet.py:337,379-380,907-908,818), bias / residual / scaling behind (istftnet.py:394,932,
 * 822,829).
 * ------------------------------------------------------------------------------------------ */
enum { MI355_ACT_NONE = 0, MI355_ACT_LEAKY = 1, MI355_ACT_SNAKE = 2, MI355_ACT_GELU = 3,
       MI355_ACT_ELU = 4,        /* prologue or epilogue: x > 0 ? x : expm1(x)  (Mimi SEANet, codec/models/mimi/modules/seanet.py) */
       MI355_ACT_SILU = 5,       /* epilogue */
       MI355_ACT_GELU_TANH = 6,  /* epilogue: nn.gelu_approx / GELU(tanh) */
       MI355_ACT_TANH = 7 };     /* epilogue */

typedef struct {
  /* input activation */
  const float* x;      /* [B, Lin, ldx] */
  int64_t x_bstride;   /* elements between batch items */
  int32_t ldx;         /* elements between rows (for the flattened strided conv: stride*Cin) */
  int32_t x_off;       /* element offset added to every row start (flattened conv: -pad*Cin) */
  int32_t Cin;         /* logical input channels per tap */
  int32_t Lin;         /* padded input rows */
  const int32_t* lens_in; /* [B] valid input rows, nullable => Lin */
  int32_t flat_valid;  /* 0: validity by row; >0: validity by flat element index in
                          [0, lens_in[b]*flat_valid)  (flat_valid = true channel count) */
  /* packed weights */
  const uint16_t* w;   /* from mi355_pack_conv_weight */
  int32_t Cout;        /* GEMM N (for conv_transpose: up_s * Cout_real) */
  int32_t K;           /* taps */
  int32_t dil;         /* dilation */
  int32_t pad;         /* left zero padding in rows */
  /* prologue: t = x*pre_scale[b,c] + pre_shift[b,c]; then activation */
  const float* pre_scale; /* [B, pre_ld] nullable */
  const float* pre_shift; /* [B, pre_ld] nullable (must be set iff pre_scale is) */
  int32_t pre_ld;
  int32_t pre_act;     /* MI355_ACT_NONE / LEAKY / SNAKE */
  float pre_slope;     /* leaky slope */
  const float* pre_alpha; /* [Cin padded to 32] snake alpha, nullable unless SNAKE */
  /* epilogue, in this order: v = act(acc + bias[n]) * colscale[n]; v += res; v += (accumulate ? old y : 0); y = v * out_scale
     (out_scale multiplies the accumulated value too: the BigVGAN / iSTFTNet stage mean sets out_scale = 1 / num_kernels on the LAST accumulating block only) */
  const float* bias;   /* [Cout] nullable */
  int32_t post_act;    /* MI355_ACT_NONE / LEAKY / GELU */
  float post_slope;
  const float* res;    /* residual [B, *, ldr], row = out_row >> res_shift; nullable */
  int64_t res_bstride;
  int32_t ldr;
  int32_t res_shift;
  float out_scale;     /* applied after the residual add */
  int32_t accumulate;  /* 1: y_new = y_old + value (resblock mean, istftnet.py:824-829) */
  float* y;            /* [B, Lout, ldy] */
  int64_t y_bstride;
  int32_t ldy;
  int32_t Lout;        /* GEMM rows per item (conv: output length; convT: Lin + K - 1) */
  const int32_t* lens_out; /* [B] valid GEMM rows, nullable => Lout */
  /* polyphase conv_transpose store: GEMM column n = r*up_cout + co is written to
     row u*up_s + r - up_p + up_row_off, column co, if 0 <= row < up_Lout (lens_up[b]) */
  int32_t up_s;        /* 0 = plain store */
  int32_t up_p;
  int32_t up_cout;
  int32_t up_row_off;
  int32_t up_Lout;
  const int32_t* lens_up; /* [B] nullable */
  int32_t B;
  int32_t precision;   /* 2 = bf16 hi+lo split (default; ~16 mantissa bits of the fp32 activation), 1 = single bf16 pass,
                          3 = single fp16 pass: activations rounded to fp16 (saturating), weights packed with MI355_W_F16,
                          4 = fp16 hi+lo split (~22 mantissa bits of the activation) on MI355_W_F16 weights: fp16 checkpoints
                              (Whisper) at fp32-activation accuracy,
                          5 = fp16 hi pass + block-scaled e4m3 lo pass on an MX image (mi355_pack_conv_weight_mx_host): the residual
                              t - fp16(t) (2^-11 of the value) and a second, e4m3 copy of the weights go through
                              v_mfma_scale_f32_32x32x64_f8f6f4 at twice the 16-bit rate, two taps per instruction, E8M0 scale per window
                              row and 32 channels / per output column: ~15 significant bits of the activation (conv sweep: <= 2e-5 of
                              the peak).  Launches the wave-specialised kernel does not take (few tiles, K % 4 != 3, thin outputs) run
                              the precision-4 arithmetic on the image's fp16 slices,
                          6 = fp16 hi pass + block-scaled FP4 (OCP e2m1) lo pass on an MX4 image (mi355_pack_conv_weight_mx4_host): as 5
                              with 4-bit elements on both sides of the lo product -- the matrix pipe runs them at 4x the 16-bit rate (the
                              lo pass costs ~0.27 of a 16-bit pass instead of ~0.55), the lo planes in LDS and the lo weight stream
                              halve; ~13 significant bits of the activation (waveform: ~4e-4 of the peak, 70 dB, against the 2e-3 /
                              50 dB bars: profiles/r6_split_format_study_fp6.txt).  Same eligibility and fallback as 5 */
  int32_t tile;        /* 0 = auto, else BM*1000+BN (128128, 64128, 64064), 6128128 / 7128128 = wave-specialised 8-wave kernels (ws4 / ws3),
                          2064128 / 2064064 (+ 10000000 * groups) = split-K on 64-row tiles (needs split_ws) */
  /* optional instance-norm statistics of the STORED output, fused into the epilogue (plain stores only): per block of
     MI355_STATS_ROWS output rows and per channel the pair (sum, sum of squared deviations from the block mean), written
     (never accumulated) to stats_partial[b][row / MI355_STATS_ROWS][c][0..1]; consumed by mi355_adain_from_partials.
     Saves the separate read pass of InstanceNorm1d over the tensor this conv just produced (istftnet.py:173-338). */
  float* stats_partial;   /* nullable; [B, ceil(Lout / MI355_STATS_ROWS), Cout, 2] */
  int64_t stats_bstride;  /* elements between batch items */
  /* SnakeBeta prologue (Qwen3 codec decoder, speech_tokenizer.py SnakeBeta): with pre_act == MI355_ACT_SNAKE and
     pre_inv_beta set, t + pre_inv_beta[c] * sin^2(pre_alpha[c] * t)  (alpha = exp(log_alpha), inv_beta = 1/(exp(log_beta)+eps),
     evaluated once on the host); null = plain Snake with 1/alpha. */
  const float* pre_inv_beta; /* [Cin padded to 32] nullable */
  /* per-output-column scale applied after the epilogue activation and before the residual add (LayerScale:
     x + scale[c] * f(x), codec/models/mimi/modules/transformer.py LayerScale; ConvNeXt gamma). */
  const float* post_colscale; /* [Cout] nullable */
  /* optional workspace for launches of few output tiles (one utterance per call): with it the dispatcher may cut the input-channel range of a
     tile over several workgroups (fp32 partial tiles [B][groups][Lout][Cout] in the workspace) and apply the epilogue above in a second kernel
     that sums the groups in a fixed order.  NULL = never split.  The workspace is scratch of this call only (stream-ordered reuse is safe). */
  void* split_ws;
  int64_t split_ws_bytes;
  /* fake-quantised module input (KittenTTS activation_quant_modules, kitten_tts/quant.py:4-24): [B][2] = {-min, max} of the PROLOGUE'S OUTPUT over
     the utterance, both joined with 0, as mi355_fake_quant_extrema leaves them; the prologue then ends with the reference's dynamic uint8
     quantise / dequantise of every value (float32 op by op), so the quantised tensor is never materialised.  NULL = no quantisation. */
  const float* pre_fq;
  /* optional per-block extrema of the STORED output (round 5, ABI 33; KittenTTS): (min, max) per block of MI355_STATS_ROWS output rows and per channel,
     written (never accumulated) to ext_partial[b][row / MI355_STATS_ROWS][c][0..1], laid out like stats_partial.  The producer of a fake-quantised
     conv's input thereby hands over what that conv's extrema pass (mi355_fake_quant_extrema: one more read of the whole tensor) would compute: every
     quantised prologue in use is monotone per channel, so mi355_fake_quant_extrema_from_partials needs only these.  Produced by the quantising-prologue
     instantiations of the wave-specialised kernel (a fake-quantised model's convs feed fake-quantised convs): a launch with ext_partial set must
     satisfy mi355_conv_gemm_ext_supported and then always runs on that kernel. */
  float* ext_partial;     /* nullable; [B, ceil(Lout / MI355_STATS_ROWS), Cout, 2] */
  int64_t ext_bstride;    /* elements between batch items */
  /* pre-split activations (round 6, ABI 35): the wide linears of the transformer encoders (Whisper: 768 -> 768 / 2304 / 3072 at 96 000 rows) are bound
     by the producers' fp32 -> hi + lo conversion, which every one of the C_out / 128 column tiles of a row tile repeats
     (profiles/r6_conv_big_gemm_b64_call11.txt).  A SPLIT tensor has the layout of the fp32 tensor it replaces, each 32-bit word holding the 16-bit hi
     part of the value (bits 0-15) and the 16-bit lo residual (bits 16-31) in the type of the launch's precision (2: bfloat16, 4: IEEE half) -- exactly
     the two numbers the fp32 path's prologue would have produced, so both paths give the same bits.  x_split = 2 / 4: x holds split words (same
     strides; must equal `precision`; no prologue; wave-specialised kernel only: 16-byte aligned rows).  y_split = 2 / 4: the epilogue stores split
     words of its result instead of floats (for the launch that consumes y; res / accumulate / statistics still work on the float value).  Producers
     other than a conv epilogue: mi355_split16 (elementwise), mi355_layernorm (split field). */
  int32_t x_split;
  int32_t y_split;
} mi355_conv_gemm_args;
#define MI355_STATS_ROWS 64

int mi355_conv_gemm(const mi355_conv_gemm_args* a, void* stream);
/* 1 = this launch can write ext_partial (pre_fq set, precision 2, plain store, a prologue / epilogue pair the wave-specialised kernel's quantising
 * instantiations carry, 16-byte aligned channels-last rows); 0 otherwise.  ext_partial itself need not be set in the probe. */
int mi355_conv_gemm_ext_supported(const mi355_conv_gemm_args* a);
/* Timeline probe of the wave-specialised kernel (tile code 46128128, tools/conv_timeline.py): device buffer of
 * [ceil(grid / 16)][8][4] uint64 s_memtime stamps (consumer wave 0 of every 16th workgroup, its first 8 tiles: tile start, first window
 * staged, main loop done, stores issued); NULL = off. */
int mi355_conv_ws4_debug_buffer(void* device_buffer);
/* Persistent workgroups the wave-specialised kernel's NEXT launches may take (process-wide, read on the host at launch time; 0 = default, two per
 * CU): lets independent convs of one stage run side by side on separate streams, each on a share of the CUs (the three resblocks of an MRF stage:
 * istftnet.py:797-835 loops over them one after the other). */
int mi355_conv_ws4_resident(int wgs);
/* Host-side packing (CPU, run once at load time).
 * w: float32 [Cout, K, Cin] in the MLX conv layout (values already weight-normed / bf16 rounded),
 * out: uint16 buffer of mi355_packed_conv_weight_elems(Cout, K, Cin) elements. */
int64_t mi355_packed_conv_weight_elems(int32_t Cout, int32_t K, int32_t Cin);
int mi355_pack_conv_weight_host(const float* w_host, int32_t Cout, int32_t K, int32_t Cin, uint16_t* out_host);
/* Same with an explicit element type: MI355_W_BF16 (precision 1 / 2) or MI355_W_F16 (precision 3; bf16-valued checkpoint
 * weights are exactly representable in fp16 down to 2^-17). */
enum { MI355_W_BF16 = 0, MI355_W_F16 = 1, MI355_W_FP8 = 2 };   /* MI355_W_FP8: row-major GEMV images only (mi355_pack_rowmajor_fp8_host) */
int mi355_pack_conv_weight_host_dt(const float* w_host, int32_t Cout, int32_t K, int32_t Cin, int32_t dtype, uint16_t* out_host);
/* The MX image of precision 5 (bf16-valued checkpoint weights are exact in its fp16 part): per 32-channel chunk K fp16 tap slices followed
 * by ceil(K / 2) e4m3 tap-pair slices, then one E8M0 scale byte per (padded) output column.  out_host: mi355_packed_conv_weight_mx_bytes bytes. */
int64_t mi355_packed_conv_weight_mx_bytes(int32_t Cout, int32_t K, int32_t Cin);
int mi355_pack_conv_weight_mx_host(const float* w_host, int32_t Cout, int32_t K, int32_t Cin, uint8_t* out_host);
/* The MX4 image of precision 6: the MX image's geometry and byte count (mi355_packed_conv_weight_mx_bytes), the tap-pair slots holding one kilobyte
 * of e2m1 codes per 32-column group (K block b of the matrix instruction = tap 2 p + b = lane half b) and the column scales of the 4-bit grid
 * (e[n] = floor(log2(max |w[n]|)) - 2). */
int mi355_pack_conv_weight_mx4_host(const float* w_host, int32_t Cout, int32_t K, int32_t Cin, uint8_t* out_host);

/* ------------------------------------------------------------------------------------------
 * Instance-norm statistics + AdaIN coefficients.
 * Replaces InstanceNorm1d / AdaIN1d (istftnet.py:173-338): per (b, c) mean / biased variance over
 * time, then scale = (1+gamma)*rsqrt(var+eps), shift = beta - mean*scale with
 * [gamma, beta] = fc(style) (computed by conv_gemm into `gb`).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t C; int32_t L; const int32_t* lens;
  int32_t B;
  double* sums;        /* workspace [B, C, 2], zeroed by the call */
  const float* gb;     /* [B, gb_ld]: gamma at [0,C), beta at [C,2C) */
  int32_t gb_ld;
  float eps;
  float* scale; float* shift; int32_t out_ld; /* [B, out_ld] */
  int32_t reuse_sums;  /* 1: `sums` already holds the statistics of this x (a previous call with the same x, lens): only the coefficients are
                          recomputed from another gb -- the three AdaINResBlock1 of a generator stage normalise the SAME input (istftnet.py:822-829) */
} mi355_adain_coef_args;
int mi355_adain_coef(const mi355_adain_coef_args* a, void* stream);

/* AdaIN coefficients from the per-block partial statistics a conv_gemm epilogue wrote (see stats_partial): merges the
 * blocks in float64 (Chan's parallel-variance formula; block row counts follow from lens / L), then the same
 * scale = (1+gamma)*rsqrt(var+eps), shift = beta - mean*scale as mi355_adain_coef. */
typedef struct {
  const float* partials; int64_t bstride;  /* [B, ceil(L / MI355_STATS_ROWS), C, 2] */
  int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* gb; int32_t gb_ld; float eps;
  float* scale; float* shift; int32_t out_ld;
} mi355_adain_partials_args;
int mi355_adain_from_partials(const mi355_adain_partials_args* a, void* stream);

/* LayerNorm over the channel axis of rows, optionally fused with a residual add in front and
 * (1+gamma)*xhat+beta / weight*xhat+bias and LeakyReLU behind.
 * Replaces nn.LayerNorm (modules.py:35,445,481,523,537,551), AdaLayerNorm (modules.py:71-90). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;
  const float* res; int64_t res_bstride; int32_t ldr; /* nullable: normalise x + res */
  int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* weight; const float* bias;   /* [C] nullable */
  const float* ada_gb; int32_t ada_ld;      /* [B, ada_ld]: gamma [0,C), beta [C,2C); nullable */
  float eps;
  int32_t post_act; float post_slope;
  float* y; int64_t y_bstride; int32_t ldy;
  int32_t y_split;   /* 0, or 2 / 4: y receives SPLIT words (mi355_conv_gemm_args.x_split of the linear that consumes it) instead of floats (ABI 35) */
} mi355_layernorm_args;
int mi355_layernorm(const mi355_layernorm_args* a, void* stream);
/* Elementwise fp32 -> SPLIT words (x_split of mi355_conv_gemm_args; fmt 2 = bfloat16 hi | lo << 16, 4 = IEEE half) over n contiguous values, n a
 * multiple of 4, both pointers 16-byte aligned; y may alias x.  For activations no conv epilogue / LayerNorm produces (attention outputs). */
int mi355_split16(const float* x, void* y, int64_t n, int32_t fmt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bidirectional LSTM recurrence (the x-projection is a conv_gemm).
 * Replaces the per-time-step Python loops of LSTM._forward_direction/_backward_direction
 * (modules.py:150-240): gates i,f,g,o; c = f*c + i*g; h = o*tanh(c).
 * xp: [B, L, ldxp] gate pre-activations x@Wx^T + b_ih + b_hh, forward gates in columns [0,4H),
 * backward in [4H,8H).  wh: packed by mi355_pack_lstm_wh_host ([2][H/8][4H][8] bf16).
 * out: [B, L, ldo], forward h in columns [0,H), backward in [H,2H).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* xp; int64_t xp_bstride; int32_t ldxp;
  const uint16_t* wh; int32_t H;
  int32_t L; const int32_t* lens; int32_t B;
  float* out; int64_t out_bstride; int32_t ldo;
  int32_t quant_h;  /* 1: the recurrent product sees fake_quant_dynamic_u8(h) of each step's hidden vector; the emitted h is not quantised
                     * (KittenTTS LSTM with activation_quant, kitten_tts/modules.py:178,224) */
  int32_t wh_f16;   /* 1: wh holds IEEE half values (mi355_pack_lstm_wh16_host with f16 = 1: float32 checkpoints, 11 significant bits); 0: bf16 */
  float wh_scale;   /* with wh_f16: the packed values are W / wh_scale (wh_scale a power of two; 0 = 1): the recurrent sum is multiplied by it.  A bf16
                     * checkpoint scaled by 2^k into the normal range of IEEE half is held EXACTLY (8 significant bits), every fp32 product and sum is
                     * the unscaled one times 2^k, so the result is bit-identical to the bf16 image's -- at half the VALU work (v_fma_mix_f32 reads a
                     * half operand directly; a bf16 pair costs a shift / mask per weight on top of the FMA) */
} mi355_lstm_args;
int mi355_lstm_bidir(const mi355_lstm_args* a, void* stream);
int mi355_pack_lstm_wh_host(const float* wh_fwd_host, const float* wh_bwd_host, int32_t H, uint16_t* out_host);
/* same layout, element type chosen by f16 (0 = bf16 like mi355_pack_lstm_wh_host, 1 = IEEE half, round to nearest even) */
int mi355_pack_lstm_wh16_host(const float* wh_fwd_host, const float* wh_bwd_host, int32_t H, int32_t f16, uint16_t* out_host);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention over short sequences (PL-BERT, T <= 512), fp32.
 * Replaces AlbertSelfAttention's QK^T / softmax / PV (modules.py:493-508).
 * qkv: [B, T, ld] with q at column 0, k at column D, v at 2D (D = heads*dh); keys >= lens[b]
 * get the reference's additive -10000 mask.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* qkv; int64_t bstride; int32_t ld;
  int32_t heads; int32_t dh; int32_t T; const int32_t* lens; int32_t B;
  float* out; int64_t out_bstride; int32_t ldo;
} mi355_attention_args;
int mi355_attention(const mi355_attention_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small glue kernels of Kokoro's Model.__call__ (kokoro.py:111-177).
 * ------------------------------------------------------------------------------------------ */
/* y[b, l, 0:C] = table[idx[b, l], :] (+ table2[l, :] + row2[:]), rows >= lens[b] zeroed.
 * nn.Embedding (modules.py:24,429-431, 466-470) and the duration alignment gather that the
 * reference writes as a one-hot matmul (kokoro.py:161-169). */
typedef struct {
  const float* table; int32_t ld_table;          /* [rows, ld_table] or per-batch if table_bstride != 0 */
  int64_t table_bstride;
  const float* pos_table; int32_t ld_pos;        /* nullable: + pos_table[l, :] */
  const float* add_row;                          /* nullable: + add_row[:] */
  const int32_t* idx; int32_t idx_ld;            /* [B, idx_ld] */
  int32_t C; int32_t L; const int32_t* lens; int32_t B;
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_gather_rows_args;
int mi355_gather_rows(const mi355_gather_rows_args* a, void* stream);

/* k | v projection rows -> the 16-bit head-major blocks the decode-step attention streams (round 6, ABI 36; stt/models/whisper/whisper.py:360-365: the reference
 * computes k and v of the cross-attention once per window and caches them in the model's 16-bit dtype).  kv: float32 [B, T, ld] with G * H * dh value columns
 * (G groups -- k, v -- of H heads); out: [G, B, H, T, dh] of MI355_KV_BF16 / MI355_KV_F16, contiguous; round to nearest even.  One pass (read 4 bytes, write 2
 * per value) instead of a permute copy and a conversion copy per group.  dh a multiple of 8, ld a multiple of 4, 16-byte aligned pointers. */
int mi355_kv_head_major16(const float* kv, int64_t kv_bstride, int32_t ld, int32_t B, int32_t T, int32_t G, int32_t H, int32_t dh, void* out, int32_t out_dtype,
                          void* stream);

/* y[b, l, c] = v[b, c] for l < lens[b] (style broadcast, modules.py:393-395). */
int mi355_broadcast_rows(const float* v, int32_t ldv, int32_t C, float* y, int64_t y_bstride, int32_t ldy,
                         int32_t L, const int32_t* lens, int32_t B, void* stream);

/* duration head (kokoro.py:140-147): dur = clip(round(sum_j sigmoid(logits[b,t,j]) / speed), 1, 100),
 * then per item the exclusive scan and the frame->token index (kokoro.py:148-160).
 * frames[b] = sum of durations; idx[b, f] = token of frame f (f < frames[b], up to idx_ld). */
typedef struct {
  const float* logits; int64_t bstride; int32_t ld; int32_t bins;
  int32_t T; const int32_t* lens; int32_t B; float speed;
  const int32_t* forced_dur; /* nullable [B, T]: use these instead of the prediction */
  int32_t* dur;      /* [B, T] */
  float* dur_raw;    /* [B, T] nullable: pre-round value (tests check the rounding margin) */
  int32_t* frames;   /* [B] */
  int32_t* idx; int32_t idx_ld; /* [B, idx_ld], entries >= frames[b] untouched */
  int32_t max_frames; /* upper clip of one duration: 0 = 100 (Kokoro, kokoro.py:145-147); < 0 = none (KittenTTS, kitten_tts.py:398) */
} mi355_duration_args;
int mi355_duration_align(const mi355_duration_args* a, void* stream);

/* Dynamic per-tensor uint8 fake quantisation of one activation tensor per utterance (tts/models/kitten_tts/quant.py:4-24, the
 * ``maybe_fake_quant`` the KittenTTS modules apply to their inputs): y[b] = fq(act(scale[b,c] * x[b] + shift[b,c])) over the valid rows
 * of x [B, L, C] (rows >= lens[b] are neither read nor written).  The optional affine + activation prologue is what the consuming conv
 * would otherwise have fused (AdaIN + LeakyReLU / Snake): a quantised input has to exist in memory, because its extrema are needed first.
 * x == y is allowed.  ``minmax`` [B, 2] float32 is scratch (overwritten: {-min, max} per utterance). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* pre_scale; const float* pre_shift; int32_t pre_ld;   /* nullable pair [B, pre_ld] */
  int32_t pre_act; float pre_slope; const float* pre_alpha;          /* MI355_ACT_NONE / LEAKY / SNAKE (alpha [C]) */
  float* y; int64_t y_bstride; int32_t ldy;
  float* minmax;
} mi355_fake_quant_args;
/* {-min, max} per utterance of act(scale x + shift) from the per-block, per-channel extrema a producing conv left (mi355_conv_gemm_args.ext_partial):
 * here x = the partials [B, ceil(L / MI355_STATS_ROWS), C, 2], x_bstride = elements between items (ldx, y unused), everything else as in
 * mi355_fake_quant_extrema.  Per channel the prologue is evaluated at the channel's min and max (both ends: the affine slope may be negative) -- equal
 * to the sweep's result whenever the float32 evaluation of the prologue is monotone (always for the affine / LeakyReLU prologues; Snake's
 * x + sin^2(a x) / a has derivative 1 + sin(2 a x) >= 0 and can round non-monotonically by an ulp where that derivative vanishes, which moves the
 * quantiser's scale by <= 1e-7 relative). */
int mi355_fake_quant_extrema_from_partials(const mi355_fake_quant_args* a, void* stream);
int mi355_fake_quant_u8(const mi355_fake_quant_args* a, void* stream);
/* Extrema pass alone for a conv that quantises in its prologue (mi355_conv_gemm_args.pre_fq): minmax[b] = {-min, max} of act(scale x + shift)
 * over utterance b's valid rows, joined with 0; y is not touched (may be NULL).  The prologue value is evaluated exactly as the conv prologues
 * evaluate it (fmaf, v_sin, v_rcp + Newton), not with the libm calls of mi355_fake_quant_u8. */
int mi355_fake_quant_extrema(const mi355_fake_quant_args* a, void* stream);

/* Residual vector quantisation, encode side (codec/models/mimi/modules/quantization.py:37-45, 84-96: per layer argmin_b (|e_b|^2 / 2 - x . e_b),
 * first minimum, then x -= e_idx in float32): the quantiser of Mimi.encode (mimi.py:146-153; CSM audio context, sesame.py:527-559) and of the
 * Qwen3-TTS speech tokenizer's encoder (speech_tokenizer.py:1037-1058).  x is the PROJECTED input (after input_proj). */
typedef struct {
  const float* x; int64_t rows; int32_t ldx; int32_t D;      /* [rows, D] float32 */
  const float* tables;    /* [n_layers, bins, D]  embedding = embedding_sum / max(cluster_usage, 1e-5) */
  const float* tables_t;  /* [n_layers, D, bins]  the same values transposed (the search reads these) */
  const float* c2;        /* [n_layers, bins]     |e|^2 / 2 */
  int32_t bins; int32_t n_layers;
  int32_t* codes; int32_t ld_codes;   /* [rows, ld_codes >= n_layers] */
  float* margins;                     /* nullable, same layout as codes: second-best score - best score of every decision */
} mi355_rvq_encode_args;
int mi355_rvq_encode(const mi355_rvq_encode_args* a, void* stream);

/* ECAPA-TDNN speaker encoder of Qwen3-TTS (tts/models/qwen3_tts/speaker_encoder.py), the row-wise pieces between its convolutions (which are
 * mi355_conv_gemm launches).  mi355_ecapa_rows: y[b, r, c] = f(x[b, s, c]) * sigmoid(gate[b, c]) + res[b, s, c] with s = reflect(r - pad) for
 * r in [0, T + 2 pad): reflect_pad_1d (:11-26: mirror without repeating the edge row; needs pad < T), the Res2Net input "chunk + previous
 * output" (:98-101), the tanh in front of the attention conv (:213), SqueezeExcitationBlock's x * sigmoid(.) (:138-141) + the block residual (:180). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;        /* [B, T, C] float32, any row stride (a channel slice of a wider buffer) */
  const float* gate; int32_t gate_ld;                    /* nullable [B, C] LOGITS of the gate */
  const float* res; int64_t res_bstride; int32_t ldr;    /* nullable [B, T, C] */
  int32_t pre_tanh;                                      /* 1: f = tanh, 0: identity */
  float* y; int64_t y_bstride; int32_t ldy;              /* [B, T + 2 pad, C] */
  int32_t B; int32_t T; int32_t C; int32_t pad;
} mi355_ecapa_rows_args;
int mi355_ecapa_rows(const mi355_ecapa_rows_args* a, void* stream);
/* mean[b, c] = mean_t x[b, t, c];  std[b, c] (nullable) = sqrt(biased variance + eps)  (speaker_encoder.py:133, 201-202) */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;
  int32_t B; int32_t T; int32_t C;
  float eps;
  float* mean; float* std; int32_t out_ld;               /* [B, out_ld >= C] each */
} mi355_time_moments_args;
int mi355_time_moments(const mi355_time_moments_args* a, void* stream);
/* AttentiveStatisticsPooling's tail (speaker_encoder.py:219-228): w = softmax over TIME of logits[b, :, c];  out[b, c] = sum_t w x;
 * out[b, C + c] = sqrt(max(sum_t w (x - mean)^2, eps)). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;        /* [B, T, C] */
  const float* logits; int64_t l_bstride; int32_t ldl;   /* [B, T, C] */
  int32_t B; int32_t T; int32_t C;
  float eps;
  float* out; int32_t out_ld;                            /* [B, out_ld >= 2 C] */
} mi355_attentive_pool_args;
int mi355_attentive_pool(const mi355_attentive_pool_args* a, void* stream);

/* Anti-aliased activation of BigVGAN (codec/models/bigvgan/resample.py:157-177 ``Activation1d`` with SnakeBeta, activation.py:27-51):
 * x [B, L, C] channels-last -> y [B, L, C]:  2x up-sampling (edge pad 5, depthwise transposed conv with the 12-tap Kaiser-sinc filter, x 2, trimmed
 * to 2L: resample.py:101-136), a = u + inv_beta[c] * sin^2(alpha[c] * u), 2x down-sampling (edge pad 5 / 6, the 12-tap low-pass at stride 2:
 * resample.py:49-98, 139-154).  Ratio 2 and 12 taps only (the only configuration the reference builds).  up_filter / down_filter: 12 floats each on
 * the device (they are module buffers of the checkpoint); alpha, inv_beta: [C] (exp() of the log-scale parameters and 1 / (beta + 1e-9), applied at load). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* up_filter; const float* down_filter;
  const float* alpha; const float* inv_beta;
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_aa_act_args;
int mi355_aa_activation(const mi355_aa_act_args* a, void* stream);

/* Polyphase FIR sample-rate conversion (mlx_audio/resample.py:29-47 ``resample_audio_array`` = scipy.signal.resample_poly(x, up, down, window=taps,
 * padtype="edge") with the kaiser_best taps of resample.py:15-26; also utils.py:541-578 ``resample_audio``):  x [rows, n_in] -> y [rows, n_out],
 *   y[r][n] = sum_k table[k][p] * x[r][clamp(q - k, 0, n_in - 1)],   t = (n + first) * down,  p = t % up,  q = t / up.
 * table [K, up] float64 on the device (tap-major) = the taps x up, shifted right by (down - half % down) zeros (half = (len(taps) - 1) / 2; resample_poly
 * centres its output this way), split by phase: table[k][p] = padded[p + k * up], zero where that runs past the end; first = (half + shift) / down =
 * the outputs resample_poly drops at the front; n_out = ceil(n_in * up / down).  Sums are float64 like scipy's.  Two kernels: four same-phase outputs per
 * thread with a float32 input window in LDS when that window fits 64 KB (every pair of the usual rates), else one output per thread with a float64
 * window of (255 * down / up + K + 1) samples; a conversion that fits neither (384 kHz -> 8 kHz) is refused. */
typedef struct {
  const float* x; int64_t x_bstride; int32_t n_in; int32_t rows;
  const double* table; int32_t up; int32_t down; int32_t K; int32_t first;
  float* y; int64_t y_bstride; int32_t n_out;
} mi355_resample_args;
int mi355_resample_poly(const mi355_resample_args* a, void* stream);

/* AdaIN + LeakyReLU + depthwise ConvTranspose1d(k3, s2) with the first output dropped
 * (AdainResBlk1d pool, istftnet.py:879-881,907-915): x [B, L, C] -> y [B, 2L, C]. */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* scale; const float* shift; int32_t pre_ld; float slope;
  const float* w;    /* [C, 3] folded depthwise weight */
  const float* bias; /* [C] */
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_pool_up2_args;
int mi355_adain_pool_up2(const mi355_pool_up2_args* a, void* stream);

/* interpolate1d (tts/models/interpolate.py:61-132): x [rows, W] -> y [rows, size] (rows = N * C of the reference's [N, C, W]).
 * mode 0 = nearest: source index floor(i * scale) with scale = float32(W / size); mode 1 = linear with torch semantics: source coordinate
 * i * scale + half_scale - 0.5 clamped at 0 (scale = float32(W / size), half_scale = float32(0.5 * W / size)), or i * scale with
 * scale = float32((W - 1) / (size - 1)) when align_corners; every step rounded to float32 like the MLX op sequence. */
typedef struct {
  const float* x; int64_t x_rstride; int32_t W; int64_t rows;
  int32_t size; int32_t mode; int32_t align_corners; float scale; float half_scale;
  float* y; int64_t y_rstride;
} mi355_interp1d_args;
int mi355_interpolate1d(const mi355_interp1d_args* a, void* stream);

/* Scalar strided conv (1 -> 1 channel, k3, stride 2, pad 1) writing one column of a wider buffer:
 * Decoder.F0_conv / N_conv (istftnet.py:973-974,983-984).  x [B, Lin], y[b, l, col]. */
int mi355_conv1d_c1_k3s2(const float* x, int32_t ldx_b, int32_t Lin, const int32_t* lens_in,
                         float w0, float w1, float w2, float bias, float* y, int64_t y_bstride, int32_t ldy, int32_t col,
                         int32_t Lout, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------
 * Harmonic source + STFT features + iSTFT head of the iSTFTNet Generator.
 * ------------------------------------------------------------------------------------------ */
/* SineGen + SourceModuleHnNSF (istftnet.py:548-709, 797-799): f0 [B, L2] at 2 values / frame ->
 * har_source [B, L2*up].  rand_ini [B, H] uniform, noise [B, L2*up, H] normal (explicit, so that
 * results are reproducible).  phase_ws: workspace [B, H, L2+1] floats. */
typedef struct {
  const float* f0; int32_t ld_f0; int32_t L2; const int32_t* lens2; int32_t B;
  int32_t up; int32_t H; float sr; float sine_amp; float noise_std; float voiced_thr;
  const float* rand_ini; const float* noise;
  const float* lin_w; float lin_b;  /* l_linear [H], bias */
  float* phase_ws;
  float* out; int32_t ld_out;
  float* quant_ws;  /* nullable [B, 2] scratch: when given, sine_wavs [L, H] goes through fake_quant_dynamic_u8 (per utterance) before l_linear
                     * (KittenTTS with activation_quant on m_source.l_linear, kitten_tts/istftnet.py:711-713) */
  int32_t coarse_f32; /* 1: coarse-grid length = ceil(L * float32(1 / up)) = L2 + 1 (KittenTTS hands interpolate() a float32 scale factor,
                       * kitten_tts/istftnet.py:572,595-599); 0: ceil(L * (1.0 / up)) in doubles (Kokoro, istftnet.py:567,590-594) */
} mi355_sine_source_args;
int mi355_sine_source(const mi355_sine_source_args* a, void* stream);

/* MLXSTFT.transform (istftnet.py:473-506) for small n_fft (<= 64): reflect-centred STFT with a
 * caller-supplied window, magnitude and atan2 phase, written channels-last:
 * x [B, L] -> y[b, f, 0:nb] = |X|, y[b, f, nb:2nb] = angle(X), f in [0, L/hop], nb = n_fft/2+1. */
typedef struct {
  const float* x; int32_t ldx; int32_t L; const int32_t* lens; int32_t B;
  int32_t n_fft; int32_t hop; const float* window;
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_stft_magphase_args;
int mi355_stft_magphase(const mi355_stft_magphase_args* a, void* stream);

/* Generator tail (istftnet.py:830-835 + MLXSTFT.inverse :508-541 + dsp.istft :436-513 with
 * normalized=True, center=True): x [B, Fr, 2nb] (conv_post output) -> spec = exp(x[..., :nb]),
 * phase = sin(x[..., nb:]), irfft, synthesis window, w^2 overlap-add as a gather,
 * trim n_fft/2 each side -> audio [B, (Fr-1)*hop]. */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t Fr; const int32_t* lens; int32_t B;
  int32_t n_fft; int32_t hop; const float* window;
  float* audio; int32_t ld_audio;
} mi355_istft_head_args;
int mi355_istft_head(const mi355_istft_head_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * mlx_audio.dsp (dsp.py): batched STFT / iSTFT for arbitrary n_fft (mixed-radix 2/3/4/5 +
 * generic-prime Stockham FFT staged in LDS) and the fused |X|^2 -> mel -> log front ends.
 * ------------------------------------------------------------------------------------------ */
/* dsp.stft (dsp.py:385-433): x [B, L] (already centre-padded by the caller or pad_mode applied
 * here: 0 none, 1 reflect, 2 constant) -> out complex64 [B, n_frames, n_fft/2+1] interleaved. */
typedef struct {
  const float* x; int32_t ldx; int32_t L; int32_t B;
  int32_t n_fft; int32_t hop; const float* window; /* [n_fft] already zero-padded */
  int32_t pad_mode; int32_t n_frames;
  float* out;  /* [B, n_frames, n_fft/2+1, 2] */
} mi355_stft_args;
int mi355_stft(const mi355_stft_args* a, void* stream);

/* Fused STFT -> power/magnitude -> mel -> log (whisper/audio.py:41-82; qwen3_tts.py:64-120).
 * mode 0 (whisper): p = |X|^2, y = log10(max(mel, 1e-10)) (global max clamp + (y+4)/4 by
 *   mi355_logmel_finish); mode 1 (qwen3): p = sqrt(|X|^2 + 1e-9), y = log(max(mel, 1e-5)); mode 2 (kaldi fbank): p = |X|^2,
 *   y = log(max(mel, 1e-8)); mode 3 (vocos, codec/models/vocos/mel.py:9-33): p = |X|, y = log(max(mel, 1e-5));
 *   mode 4 (NeMo FilterbankFeatures: stt/models/parakeet/audio.py:71-79, vad/models/sortformer/sortformer.py:95-97): p = |X|^2,
 *   y = log(mel + log_guard).
 * fb: [n_mels, n_fft/2+1] float32.  out [B, n_frames, n_mels]. */
typedef struct {
  const float* x; int32_t ldx; int32_t L; int32_t B;
  int32_t n_fft; int32_t hop; const float* window; int32_t pad_mode; int32_t n_frames;
  const float* fb; int32_t n_mels; int32_t mode;
  float* out;
  float* gmax;  /* [B] running max for mode 0 (must be -inf initialised by the call), nullable */
  float log_guard;  /* mode 4: the additive guard inside the log (NeMo's log_zero_guard_value, 2^-24) */
} mi355_logmel_args;
int mi355_logmel(const mi355_logmel_args* a, void* stream);
int mi355_logmel_finish(float* y, int64_t n_per_item, const float* gmax, int32_t B, void* stream);

/* Kaldi framing for dsp.compute_fbank_kaldi (dsp.py:821-975): frame f covers x[f*shift - pad, +win) with Kaldi's edge reflection
 * (snip_edges=True: pad = 0), + dither * noise (nullable), - frame mean, pre-emphasis inside the frame, * window, zero-padded to n_fft.
 * frames [n_frames, n_fft] then goes through mi355_logmel with mode 2 (|X|^2 -> mel -> log(max(., 1e-8))), hop = n_fft, window = ones. */
typedef struct {
  const float* x; int32_t L;
  int32_t win; int32_t shift; int32_t pad; int32_t n_fft; int32_t n_frames;
  const float* window;   /* [win] */
  const float* noise;    /* [n_frames, win] standard normal, nullable */
  float dither; float preemph;
  float* frames;         /* [n_frames, n_fft] */
} mi355_kaldi_frames_args;
int mi355_kaldi_frames(const mi355_kaldi_frames_args* a, void* stream);
/* Vocos ISTFTHead, first half (codec/models/vocos/vocos.py:126-134: mag, p = split(out(x)); mag = clip(exp(mag), max 1e2);
 * S = mag * (cos p + 1j sin p)): x [B, Fr, 2 nb] float32 (row pitch ldx) -> spec [B, Fr, nb] complex64 (interleaved re, im), which
 * mi355_istft then inverts (dsp.istft, plain-window overlap-add normalisation). */
int mi355_polar_spec(const float* x, int64_t x_bstride, int32_t ldx, int32_t Fr, int32_t nb, int32_t B, float clip, float* spec, void* stream);


/* dsp.istft / ISTFTCache.istft (dsp.py:436-513, 663-738): spec [B, n_frames, nb, 2] ->
 * frames irfft(n_fft) * window, overlap-add (gather form), / norm[t] (precomputed window
 * envelope), out[b, t - trim] for t in [trim, trim+out_len).  clamp: constrain_value_range. */
typedef struct {
  const float* spec; int32_t n_frames; int32_t B;
  int32_t n_fft; int32_t hop; const float* window; const float* norm; /* [ola_len] */
  int32_t norm_mode;  /* 0: out/norm always (ISTFTCache); 1: divide only where norm > 1e-10 (dsp.istft) */
  int32_t clamp; int32_t trim; int32_t out_len;
  float* frames_ws;   /* workspace [B, n_frames rounded up to even, n_fft] floats */
  float* out; int32_t ld_out;
} mi355_istft_args;
int mi355_istft(const mi355_istft_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scaled-dot-product attention with GQA, causal / sliding-window masks and ragged lengths, exact f32
 * (v_mfma_f32_32x32x2_f32 flash kernel for Tq > 8, a KV-cache streaming kernel for decode steps).
 * Replaces Whisper MultiHeadAttention.qkv_attention (stt/models/whisper/whisper.py:371-385: q@k, + mask,
 * softmax(precise=True), w@v) and mx.fast.scaled_dot_product_attention under the Qwen3-TTS talker / code predictor /
 * codec transformer (tts/models/qwen3_tts/talker.py:307, speech_tokenizer.py:150-186), the Mimi transformer
 * (codec/models/mimi/modules/transformer.py:96-113) and the CSM Llama blocks (tts/models/sesame/attention.py:150-175).
 * q [B, Tq, ldq] with head h at columns [h*dh, (h+1)*dh); k, v [B, Tk, ldk/ldv] with kv head g = h / (heads/kv_heads)
 * at [g*dh, (g+1)*dh) -- a KV cache is simply the buffer k/v point into.  Queries are the LAST lens_q[b] positions of
 * the lens_k[b] keys: key j is visible to query i iff j < lens_k[b], (causal) j <= i + lens_k[b] - lens_q[b],
 * (window > 0) j > i + lens_k[b] - lens_q[b] - window.  softmax(scale * q.k) over the visible keys, times v.
 * ------------------------------------------------------------------------------------------ */
enum { MI355_KV_F32 = 0, MI355_KV_BF16 = 1, MI355_KV_F16 = 2 };
typedef struct {
  const float* q; int64_t q_bstride; int32_t ldq;
  const float* k; int64_t k_bstride; int32_t ldk;
  const float* v; int64_t v_bstride; int32_t ldv;
  int32_t heads; int32_t kv_heads; int32_t dh;   /* dh: 64 or 128 */
  int32_t Tq; int32_t Tk;
  const int32_t* lens_q;  /* [B] nullable => Tq */
  const int32_t* lens_k;  /* [B] nullable => Tk */
  int32_t causal; int32_t window; float scale;
  int32_t B;
  int32_t mode;           /* 0 = auto (decode kernel when Tq <= 8), 1 = MFMA flash kernel, 2 = decode kernel */
  float* out; int64_t out_bstride; int32_t ldo;  /* [B, Tq, ldo], head h at [h*dh, (h+1)*dh) */
  const int32_t* k_start; /* [B] nullable: keys j < k_start[b] are left padding (BatchKVCache, lm/models/cache.py:502-560) */
  /* key-range split for decode steps over long key ranges (flash-decoding): partial (max, sum, out) records in split_ws
     [B * heads * Tq * 8 * (dh + 2)] floats, tickets in split_cnt [B * heads * Tq] int32 (zero before the first call; left zero by every call).
     nsplit: 0 = chosen by the library when the workspace is given, 1 = off, 2..8 = forced.  Null workspace = off. */
  float* split_ws; int32_t* split_cnt; int32_t nsplit;
  int64_t k_hstride; int64_t v_hstride; /* 0: kv head g at columns [g*dh, (g+1)*dh) of a row; else head-major planes: head g starts at
                                           g * k_hstride (rows of that head ldk apart, typically ldk = dh) */
  int32_t kv_dtype;       /* element type of k and v: MI355_KV_F32 (0: k / v are float*), MI355_KV_BF16, MI355_KV_F16 (16-bit: the checkpoint dtype the
                             reference keeps its caches in, whisper.py:360-361, lm/models/cache.py:104-176); every k / v stride is in ELEMENTS */
  /* Fused single-query decode step (Tq == 1, causal; talker.py:264-307 q_norm / k_norm / apply_rotary_pos_emb / cache update / sdpa in one launch):
     q holds the RAW projection output; new_k / new_v [B, kv_heads * dh] (items new_bstride floats apart) hold the raw k, v of the new position
     Tk - 1, which is NOT in the cache yet.  The kernel applies the per-head RMSNorm (q_norm_w / k_norm_w [dh], nullable) and the rotary
     embedding (rope_cos / rope_sin [rope_rows, dh / 2], nullable; rope_mode as in mi355_head_rope_args; position rope_pos - k_start[b]) to q and
     to the new k, attends over the cache rows [.., Tk - 1) plus the new pair, and writes the processed k and the v into cache row Tk - 1.
     With lens_k (slot caches of a continuous-batching session) item b's Tk is lens_k[b]: its new row and its rotary position are lens_k[b] - 1. */
  const float* new_k; const float* new_v; int64_t new_bstride;
  const float* q_norm_w; const float* k_norm_w; float norm_eps;
  const float* rope_cos; const float* rope_sin; int32_t rope_rows; int32_t rope_mode; int32_t rope_pos;
  /* the fused step inside the rows pipeline (steps of 9..64 sequences): in_kgroups > 1: q / new_k / new_v point into slab 0 of mi355_rows_gemm's
     partial sums and the kernel adds the in_kgroups slabs (in_kg_stride floats apart, fixed order) plus the projection bias (q_bias [heads dh],
     k_bias / v_bias [kv_heads dh], nullable) itself -- the q | k | v projection needs no separate row epilogue.  out_planes (nullable): the output
     rows additionally (out may then be null) leave as planes of planes_R rows in planes_dtype (MI355_W_BF16 / MI355_W_F16) for the o-proj GEMM;
     item b is row b.  Both need Tq == 1. */
  int32_t in_kgroups; int64_t in_kg_stride; const float* q_bias; const float* k_bias; const float* v_bias;
  const float* q_wscale; const float* k_wscale; const float* v_wscale;   /* nullable: per-column scales of the slab sums (fp8 images), indexed like the biases */
  uint16_t* out_planes; int32_t planes_R; int32_t planes_dtype;
} mi355_flash_attn_args;
int mi355_flash_attention(const mi355_flash_attn_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whisper decode step on device: SuppressBlank -> SuppressTokens -> ApplyTimestampRules -> GreedyDecoder.update
 * (stt/models/whisper/decoding.py:333-443, 302-330), one workgroup per sequence, no host round trip.
 * logits: raw decoder logits of the last position; tokens[b, 0:n] is the context so far (initial tokens included),
 * tokens[b, n] receives the selected token (eot once the sequence has ended), sum_logprobs[b] accumulates
 * log softmax(filtered)[token] while the sequence is alive.  gumbel (nullable): [B, ld] Gumbel(0,1) noise, the
 * explicit-noise form of categorical(logits / temperature) for temperature > 0; null = argmax.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* logits; int32_t ld; int32_t V; int32_t B;
  int32_t* tokens; int32_t tokens_ld; int32_t n;
  int32_t sample_begin;
  const float* suppress_mask;                   /* [V] 0 / -inf, nullable (SuppressTokens) */
  const int32_t* blank_ids; int32_t n_blank;    /* ids masked while n == sample_begin, nullable (SuppressBlank) */
  int32_t timestamp_rules;                      /* 1 = ApplyTimestampRules */
  int32_t timestamp_begin; int32_t eot; int32_t no_timestamps;  /* no_timestamps < 0: tokenizer has none */
  int32_t max_initial_timestamp_index;          /* < 0: None */
  const float* gumbel; float temperature;
  float* sum_logprobs;                          /* [B] */
  float* filtered;                              /* [B, ld] nullable: the filtered logits (parity tests) */
  const int32_t* forced_next;                   /* [B] nullable: teacher forcing -- append this token instead of the selection
                                                   (its log-prob is what gets accumulated) */
  /* optional: spread each row over 16 workgroups (two launches: partial statistics, then filter + selection with the last workgroup of a row
     merging).  split_ws: [B * 16 * 12] floats scratch, split_cnt: [B] int32, zero before the first call and left zero by every call.  Null = one
     workgroup per row. */
  float* split_ws; int32_t* split_cnt;
} mi355_whisper_step_args;
int mi355_whisper_greedy_step(const mi355_whisper_step_args* a, void* stream);
/* out[b] = softmax(logits[b, 0:V])[token]  (no_speech_prob, decoding.py:604-606). */
int mi355_softmax_prob_at(const float* logits, int32_t ld, int32_t V, int32_t B, int32_t token, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Skinny GEMM for decode steps: y[m, n] = epilogue(sum_k x[m, k] * w[n, k]), 1 <= M <= 8 rows.
 * Replaces nn.Linear / Embedding.as_linear at sequence length 1: Whisper TextDecoder (stt/models/whisper/whisper.py:347-416,
 * 498), Qwen3-TTS talker / code predictor (tts/models/qwen3_tts/talker.py:230-330, 503-764), CSM backbone / depth decoder
 * (lm/models/llama.py:46-198, tts/models/sesame/sesame.py:361-404).  HBM-bound: w is read once, row-major [N, ldw] in the
 * checkpoint's 16-bit type (mi355_pack_rowmajor16_host) or as an fp8 image (mi355_pack_rowmajor_fp8_host), products accumulate in fp32.
 * 1..4 rows (and K > 2048 / fp8 at any row count): fp32 FMAs on the exactly decoded weights.  5..8 rows with 16-bit weights and K <= 2048,
 * K % 64 == 0: v_mfma_f32_16x16x32 on the weights' own type with the input rows split into hi + lo images (bf16: ~16 mantissa bits of x, the
 * split the prefill GEMMs use; fp16: ~22) -- environment MI355_GEMV_MFMA=0 keeps the FMA kernel for A/B runs.
 * Epilogue: v = act(acc + bias[n]) * colscale[n] + res[m, n]; y = v * out_scale.  glu = 1: rows of w are interleaved
 * (gate_0, up_0, gate_1, up_1, ...) and y[m, n/2] = silu(acc_gate + b) * (acc_up + b)  (SwiGLU, talker.py TalkerMLP).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* x; int32_t ldx; int32_t M; int32_t K;
  const uint16_t* w; int32_t ldw; int32_t wdtype; int32_t N;   /* wdtype: MI355_W_BF16 / MI355_W_F16 / MI355_W_FP8 (then w holds bytes; ldw in elements) */
  const float* bias;       /* [N] nullable */
  int32_t post_act; float post_slope;
  const float* colscale;   /* [N] nullable */
  const float* res; int32_t ldr;   /* [M, ldr] nullable */
  float out_scale;         /* 0 => 1 */
  int32_t glu;
  float* y; int32_t ldy;
  /* fused input normalisation over K (the pre-LN of a transformer block, whisper.py:407-415 / talker.py:385-396):
     norm = 1: LayerNorm (x - mean) * rsqrt(var + eps) * norm_weight + norm_bias;  norm = 2: RMSNorm x * rsqrt(mean(x^2) + eps) * norm_weight */
  int32_t norm; const float* norm_weight; const float* norm_bias; float norm_eps;
  /* optional second destination: columns n >= split are written to y2[m, n - split] (q -> y, k|v -> the KV-cache slot) */
  float* y2; int32_t ldy2; int32_t split;
  /* MI355_W_FP8 only: per-output-row dequantisation scale [N] (required): acc[n] * wscale[n] precedes the bias */
  const float* wscale;
  int32_t norm_two_reads;  /* set by the library (MI355_GEMV_TWO_READS): compute the fused-norm statistics from a separate read of x; callers leave 0 */
  /* optional fused rotary embedding on the first rope_cols output columns (the q | k part of a fused q|k|v projection), interleaved pairs
     (2i, 2i + 1) inside heads of rope_dh channels -- nn.RoPE(traditional=True) / sesame/attention.py:41-105 at ONE position: rope_cos / rope_sin
     point at that position's table row [rope_dh / 2].  Plain epilogue only (no activation / colscale / residual / glu); 1..4 rows. */
  const float* rope_cos; const float* rope_sin; int32_t rope_dh; int32_t rope_cols;
  /* optional gathered input (M == 1 only): the input row is row (x_ids[0] + x_id_offset) of the table x points at (rows ldx floats apart) -- an
     embedding lookup fused into the projection that consumes it (sesame.py:392-396: embed the code just sampled, project it to the decoder width);
     the id is read on the device, so the sampler's output feeds the next GEMV without a host round trip or a separate gather launch */
  const int32_t* x_ids; int32_t x_id_offset;
  /* optional scratch for splitting K over workgroups (5..8 rows, K > 2048, no fused norm: the down projections): split_ws holds
     ceil(N / 16) * ceil(K / 2048) * 256 floats, split_cnt ceil(N / 16) int32 tickets (zero before the first call, left zero by every call).  Null =
     the single-workgroup-per-column-group kernels. */
  float* split_ws; int32_t* split_cnt;
  int32_t y2_dtype;   /* element type of y2: MI355_KV_F32 (0) / MI355_KV_BF16 / MI355_KV_F16 -- a 16-bit KV-cache slot (the reference's cache dtype);
                         y2 is then a uint16_t* in disguise and ldy2 counts 16-bit elements */
  int32_t w_policy;   /* cache policy of the weight stream (one-row kernels): 0 = default loads, 1 = non-temporal (`nt`): images that are streamed once
                         per step and should not displace what the step re-reads -- a 2.4 GB backbone in front of a 220 MB depth decoder that runs
                         31 times per frame and fits the 256 MB Infinity Cache (sesame.py:361-404) */
  /* optional attention PROLOGUE (M == 1, K = attn_heads * attn_dh <= 2048, attn_Tk <= 64 cached positions): x is the step's query row (rotated,
     head-major) and the row that enters the product is softmax(q K^T * attn_scale) V over the cache rows 0 .. attn_Tk - 1 (the step's own k | v
     already filed by the q|k|v launch) -- mx.fast.scaled_dot_product_attention + o_proj of lm/models/llama.py:80-96 at one position, one launch
     instead of two where the context is short (CSM's depth decoder: <= 32 positions, sesame.py:385-404).  attn_k / attn_v: row t of kv head g at
     element t * attn_ld + g * attn_dh, element type attn_kv_dtype (MI355_KV_*).  No fused norm / glu / rope / gathered input. */
  const void* attn_k; const void* attn_v; int32_t attn_ld; int32_t attn_Tk; int32_t attn_heads; int32_t attn_kv_heads; int32_t attn_dh;
  float attn_scale; int32_t attn_kv_dtype;
} mi355_gemv_args;
int mi355_gemv(const mi355_gemv_args* a, void* stream);
int mi355_pack_rowmajor16_host(const float* w_host, int64_t n, int32_t dtype, uint16_t* out_host);
/* fp8 weight image for the decode-step GEMVs (BASELINE config[4] names fp8: at <= 8 rows per step a Linear is a weight stream, so the 8-bit
 * image halves the bytes of a step).  Per output row: scale = 2^ceil(log2(max|w| / 448)) (a power of two: the dequantised value
 * q * scale is exactly representable in bf16, so the prefill GEMMs run the SAME quantised weights through the bf16 MFMA image), q = OCP e4m3fn
 * (round-to-nearest-even, max 448).  out_host: rows * cols bytes, scale_host: rows floats; cols % 16 == 0. */
int mi355_pack_rowmajor_fp8_host(const float* w_host, int64_t rows, int64_t cols, uint8_t* out_host, float* scale_host);

/* ------------------------------------------------------------------------------------------
 * Row-wise glue of the decoder-only transformer blocks (Qwen3-TTS talker / code predictor / codec transformer, CSM,
 * Mimi); see mlx_audio_amd/csrc/transformer.hip for the reference call sites of each.
 * ------------------------------------------------------------------------------------------ */
/* nn.RMSNorm over the channel axis: y = x * rsqrt(mean(x^2) + eps) * weight (talker.py:366-369, llama.py:100-104). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;
  int32_t C; int32_t L; const int32_t* lens; int32_t B;
  const float* weight;   /* [C] nullable */
  float eps;
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_rmsnorm_args;
int mi355_rmsnorm(const mi355_rmsnorm_args* a, void* stream);

/* Per-head RMSNorm (q_norm / k_norm, talker.py:264-266) fused with the rotary embedding, out of place: x [B, L, ldx] holds
 * `heads` heads of dh (64 or 128) channels from column 0; y may be a KV-cache slot.  Angles come from host tables
 * cos/sin [rope_rows, dh/2]; position of row l = (pos ? pos[b*pos_ld + l] : pos0 + l) - (pos_sub ? pos_sub[b] : 0).  rope_mode 0 = rotate-half pairs
 * (i, i + dh/2) (talker.py:14-36), 1 = interleaved pairs (2i, 2i+1) (nn.RoPE(traditional=True), sesame/attention.py:41-105). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx;
  int32_t heads; int32_t dh; int32_t L; const int32_t* lens; int32_t B;
  const float* norm_weight; float eps;          /* [dh] nullable: no norm */
  const float* cos_table; const float* sin_table; /* nullable: no rotation */
  const int32_t* pos; int32_t pos_ld; int32_t pos0;
  const int32_t* pos_sub;  /* [B] nullable: subtracted from the row's position (left padding of a BatchKVCache row: its first real token is position 0) */
  int32_t rope_rows;       /* rows of cos_table / sin_table; positions are checked (pos0 + L <= rope_rows) or, for device-side positions, clamped */
  int32_t rope_mode;
  float* y; int64_t y_bstride; int32_t ldy;
  /* optional second tensor handled by the same launch with the same positions (q heads -> y, k heads -> the KV-cache slot y2) */
  const float* x2; int64_t x2_bstride; int32_t ldx2; int32_t heads2; const float* norm_weight2;
  float* y2; int64_t y2_bstride; int32_t ldy2;
} mi355_head_rope_args;
int mi355_head_norm_rope(const mi355_head_rope_args* a, void* stream);

/* y[r, i] = silu(x[r, 2i]) * x[r, 2i+1]  (SwiGLU on interleaved gate / up columns, talker.py:312-330). */
int mi355_swiglu(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t I, void* stream);

/* y[b, l, :] = scale * (add[b, l, :] + sum_q table[slot_offset[q] + ids[b, l, q], :]); ids < 0 skip the slot.
 * Codec-embedding sums (qwen3_tts.py:985-1015, sesame.py:361-404) and RVQ decode (codec/models/mimi/modules/quantization.py:
 * 187-191, speech_tokenizer.py:786-800).  ids element (b, l, q) lives at ids + b*ids_bstride + l*ids_ld + q*ids_qstride. */
typedef struct {
  const float* table; int32_t ld_table;
  const int32_t* slot_offset;   /* [Q] row offset of slot q's table inside `table`, nullable => 0 */
  const int32_t* ids; int64_t ids_bstride; int32_t ids_ld; int32_t ids_qstride; int32_t Q;
  const float* add; int64_t add_bstride; int32_t add_ld;   /* nullable */
  float scale;                  /* 0 => 1 */
  int32_t C; int32_t L; const int32_t* lens; int32_t B;
  float* y; int64_t y_bstride; int32_t ldy;
} mi355_embed_sum_args;
int mi355_embed_sum(const mi355_embed_sum_args* a, void* stream);

/* Depthwise conv1d (transpose = 0: y[n] = b + sum_k w[c,k] x[n + k - pad]) or depthwise conv_transpose1d (transpose = 1:
 * y[n] = b + sum over t*stride + k - pad == n of w[c,k] x[t]), channels-last, zero outside [0, lens_in[b]).
 * ConvNeXt dwconv k7 of the Qwen3 codec decoder (speech_tokenizer.py ConvNeXtBlock), Mimi's depthwise upsampler (mimi.py:296-320),
 * SNAC's depthwise convs (codec/models/snac/layers.py:170-181, 209-233). */
typedef struct {
  const float* x; int64_t x_bstride; int32_t ldx; int32_t Lin; const int32_t* lens_in;
  const float* w;      /* [C, K] */
  const float* bias;   /* [C] nullable */
  int32_t C; int32_t K; int32_t pad; int32_t stride; int32_t transpose; int32_t B;
  float* y; int64_t y_bstride; int32_t ldy; int32_t Lout;
  /* transpose = 0 only: dilation (0 => 1; tap k reads x[n + k * dil - pad]) and an optional Snake prologue per channel,
     x -> x + pre_inv[c] * sin(pre_alpha[c] * x)^2 -- the depthwise k7 of SNAC's ResidualUnit (codec/models/snac/layers.py:209-233, 298-306) */
  int32_t dil; const float* pre_alpha; const float* pre_inv;
} mi355_dwconv_args;
int mi355_dwconv(const mi355_dwconv_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Token sampling for the codec-token LMs: suppress -> repetition penalty -> / temperature -> top-k -> top-p / min-p ->
 * categorical (Gumbel-max on caller-supplied noise).  Replaces Model._sample_token(_batch) (tts/models/qwen3_tts/qwen3_tts.py:
 * 805-925) and lm/sample_utils.py:130-281.  One workgroup per row, V <= 8192.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* logits; int32_t ld; int32_t V; int32_t B;
  const float* suppress_mask;     /* [V] 0 / -inf, nullable */
  const int32_t* history; int32_t hist_ld; int32_t n_hist; const int32_t* hist_len;  /* generated tokens [B, hist_ld]; per-row count
                                                                                        hist_len[b] (nullable => n_hist) */
  float repetition_penalty;       /* 1 = off */
  float temperature;              /* <= 0: arg-max of the penalised logits */
  int32_t top_k;                  /* <= 0 or >= V: off */
  float top_p; float min_p;       /* top_p outside (0, 1): off; min_p = 0: off */
  const float* gumbel;            /* [B, ld] Gumbel(0,1) noise, nullable => arg-max */
  const int32_t* done; int32_t done_token;  /* nullable: rows with done[b] != 0 emit done_token */
  int32_t* out; int32_t out_ld;   /* token of row b -> out[b * out_ld] */
  float* filtered;                /* [B, ld] nullable: logits after all filters (parity tests) */
} mi355_sample_args;
int mi355_sample(const mi355_sample_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Native decode-step runner: one call launches every kernel of a single-position step of a whole decoder stack (<= 64 sequences; 9..64 run
 * the rows pipeline -- mi355_rows_gemm / mi355_rows_finish -- on the tile images of the layer records),
 * replacing the per-op Python schedule of TalkerDecoderLayer / Qwen3TTSTalkerModel.__call__ (tts/models/qwen3_tts/talker.py:385-500),
 * CodePredictorModel (talker.py:615-690), LlamaModel (lm/models/llama.py:160-198) and Whisper's TextDecoder blocks with self- and
 * cross-attention (stt/models/whisper/whisper.py:405-416, 476-498).  All weights are row-major 16-bit images (mi355_pack_rowmajor16_host);
 * the KV cache of layer i is kv [B, kv_capacity, 2*kv_heads*dh] fp32 (k columns, then v), row `offset` is written by the step.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint16_t* wqkv; const float* bqkv;      /* [(heads + 2 kv_heads) dh, d_model]: q | k | v rows */
  const uint16_t* wo; const float* bo;          /* [d_model, heads dh] */
  const uint16_t* w_in; const float* b_in;      /* SwiGLU: [2 d_ff, d_model] with gate / up rows interleaved; else [d_ff, d_model] */
  const uint16_t* w_out; const float* b_out;    /* [d_model, d_ff] */
  const float* attn_norm_w; const float* attn_norm_b; const float* mlp_norm_w; const float* mlp_norm_b;
  const float* q_norm; const float* k_norm;     /* [dh] per-head RMSNorm weights, nullable */
  const float* ls1; const float* ls2;           /* [d_model] LayerScale, nullable */
  float* kv; int64_t kv_bstride; int32_t kv_capacity;   /* element type = mi355_stack_desc.kv_dtype (then kv is a uint16_t* in disguise); strides in elements */
  /* optional cross-attention block (Whisper decoder): q projection + pre-norm, K | V precomputed [B, cross_len, 2*kv_heads*dh] */
  const uint16_t* wcq; const float* bcq; const uint16_t* wco; const float* bco;
  const float* cross_norm_w; const float* cross_norm_b;
  const float* cross_k; const float* cross_v;   /* head-major [B, kv_heads, cross_len, dh] each (a head's keys contiguous) */
  int64_t cross_bstride; int64_t cross_hstride; int32_t cross_ld; int32_t cross_len;
  int32_t cross_kv_dtype;   /* MI355_KV_F32 / MI355_KV_BF16 / MI355_KV_F16: element type of cross_k / cross_v (strides in elements) */
  /* wdtype == MI355_W_FP8: per-row scales of the six images (mi355_pack_rowmajor_fp8_host), else null */
  const float* s_qkv; const float* s_o; const float* s_in; const float* s_out; const float* s_cq; const float* s_co;
  /* steps for 9..64 sequences: tile images (mi355_pack_tiles16_host) of the four projections; null = this layer only runs steps of <= 8 sequences */
  const uint16_t* wqkv_t; const uint16_t* wo_t; const uint16_t* w_in_t; const uint16_t* w_out_t;
  const uint16_t* wcq_t; const uint16_t* wco_t;   /* ... and of the cross-attention projections (Whisper decoder at 9..64 windows per step) */
} mi355_layer_desc;

typedef struct {
  int32_t n_layers; int32_t d_model; int32_t heads; int32_t kv_heads; int32_t dh; int32_t d_ff;
  int32_t norm;            /* 1 = LayerNorm, 2 = RMSNorm */
  float eps;
  int32_t glu;             /* 1 = SwiGLU MLP */
  int32_t act;             /* MLP activation when glu == 0 (MI355_ACT_GELU, MI355_ACT_GELU_TANH, ...) */
  int32_t wdtype;          /* MI355_W_BF16 / MI355_W_F16 / MI355_W_FP8 (every image of the stack, scales in the layer records) */
  int32_t causal; int32_t window;
  float attn_scale;        /* 0 => dh^-0.5 */
  int32_t rope_mode; const float* cos; const float* sin;   /* tables [rope_rows, dh/2], nullable: no rotary embedding */
  int32_t rope_rows;       /* table rows: a step at offset >= rope_rows is refused (sesame.py:817-820 raises for the same reason) */
  const int32_t* k_start;  /* [B] nullable: left padding of each sequence (lm/models/cache.py:502-560 BatchKVCache): keys before it are invisible,
                              RoPE positions count from it */
  const mi355_layer_desc* layers;                           /* HOST array of n_layers records */
  float* attn_split_ws; int32_t* attn_split_cnt;            /* nullable: workspace of mi355_flash_attn_args.split_* for B <= 8, Tq = 1 */
  float* gemv_split_ws; int32_t* gemv_split_cnt;            /* nullable: mi355_gemv_args.split_ws / split_cnt sized for the largest projection of the stack */
  const float* final_norm_w; const float* final_norm_b;     /* nullable */
  int32_t kv_dtype;        /* MI355_KV_F32 / MI355_KV_BF16 / MI355_KV_F16: element type of every layer's kv buffer (the reference keeps its caches in the
                              checkpoint dtype, lm/models/cache.py:104-176).  16-bit caches need the stores that exist: q|k|v GEMV with the rotary pairs in its
                              epilogue, the fused norm / rope attention step, or no rotary embedding at all */
  void* rows_ws; int64_t rows_ws_bytes;   /* steps for 9..64 sequences: planes + partial slabs, mi355_stack_rows_ws_bytes(d, B) bytes (nullable otherwise) */
  const int32_t* slot_lens_k;  /* [B] nullable (device): SLOT caches of a continuous-batching session -- sequence b holds slot_lens_k[b] - 1 cached positions
                                  and this step appends position slot_lens_k[b] - 1 to ITS row of every kv buffer (rotary position = that index); `offset`
                                  is then only the host's upper bound of the positions (capacity / table checks).  Needs the fused attention step
                                  (per-head norms and / or rotary embedding, causal). */
  int32_t w_policy;        /* mi355_gemv_args.w_policy of every projection of the stack (steps of 1 sequence) */
} mi355_stack_desc;

/* ------------------------------------------------------------------------------------------
 * Unidirectional LSTM over a whole sequence, any hidden size (EnCodec's two 512-wide layers, codec/models/encodec/encodec.py:89-167: the
 * reference's Metal `lstm` kernel + one matmul per step).  xproj [B, T, 4H] = x @ Wx^T + bias for all steps (the caller's GEMM), gate chunks in the
 * order i | f | g | o; wh: row-major 16-bit image [4H, H] (mi355_pack_rowmajor16_host); h, c [B, H] fp32 state (in: initial, normally zeros; out:
 * final); pre [B, 4H] scratch; out [B, T, H].  Two launches per time step (mi355_gemv with the x-projection row as residual, then the gates) --
 * or ONE (round 5, ABI 32): with gate_interleaved = 1 the rows of `wh` are ordered by hidden unit (row 4 j + g = gate g of unit j instead of the
 * reference's g H + j), so a 16-row tile of the step's GEMM holds all four gates of four units and the gates, the cell update and the stores of h
 * are that GEMM's epilogue; h then ping-pongs between `h` and the scratch `h2` (every workgroup reads the whole previous h), the final state ends
 * up in `h`.  Needs H % 64 == 0; `pre` is not used.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* xproj; int64_t xproj_bstride; int32_t ld_xproj;
  const uint16_t* wh; int32_t wdtype;   /* MI355_W_BF16 / MI355_W_F16 */
  float* h; float* c; float* pre;
  float* out; int64_t out_bstride; int32_t ld_out;
  int32_t B; int32_t T; int32_t H;
  int32_t gate_interleaved;   /* 0: wh rows in the reference's i | f | g | o blocks (two launches per step); 1: row 4 j + g (one launch per step) */
  float* h2;                  /* [B, H] scratch, gate_interleaved = 1 only */
} mi355_lstm_seq_args;
int mi355_lstm_seq(const mi355_lstm_seq_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decode steps for a batch of 9..64 sequences (BASELINE config[3]: 64 utterances; the reference's batched generation
 * tts/models/qwen3_tts/qwen3_tts.py:1651-2060 over talker.py:229-336, 385-500): nn.Linear at sequence length 1 as
 *   mi355_rows_gemm   (pure matrix-pipe GEMM: tile image of W x pre-split input planes -> fp32 partial slabs, one per K group)
 *   mi355_rows_finish (row epilogue: sum of the slabs in a fixed order, then the epilogue of mi355_gemv, optional normalisation of the
 *                      finished row, optional output as planes for the next mi355_rows_gemm; kgroups = 1: converter fp32 rows -> planes).
 * Planes: the input rows as hi + lo images of the weights' 16-bit type (x ~= hi + lo: ~16 mantissa bits for bf16, ~22 for fp16) in MFMA fragment
 * order: 16-byte piece ((((s * 2 + image) * 2 + half) * 4 + group) * R + row) holds x[row][64 s + 16 group + 8 half .. + 8), R = 16 / 32 / 64 rows;
 * 4 * R * K bytes for K columns.  Rows >= M of the planes are never read into a stored result.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint16_t* wt;       /* tile image of W [N, K] from mi355_pack_tiles16_host (MI355_W_FP8: the byte image of mi355_pack_tiles8_host) */
  int32_t wdtype;           /* MI355_W_BF16 / MI355_W_F16 (planes must hold the same type) / MI355_W_FP8 (OCP e4m3fn bytes, decoded to bf16 in registers: exact;
                               planes hold bf16; the per-row scales are applied by the row epilogue: mi355_rows_finish_args.wscale) */
  int32_t N; int32_t K;     /* K % 64 == 0 */
  const uint16_t* planes;   /* the input rows (written by mi355_rows_finish) */
  int32_t M; int32_t R;     /* rows in use / rows of the planes (16, 32 or 64) */
  float* part;              /* slab g at part + g * kg_stride: [M, ldp] fp32 partial sums over the g-th range of k steps */
  int32_t ldp; int64_t kg_stride;
  int32_t kgroups;          /* mi355_rows_kgroups(N, K) (or any count that leaves no K group empty) */
  /* kgroups == 1 only (a workgroup then holds complete sums): fused SwiGLU epilogue -- columns are (gate, up) pairs, planes_out receives
     silu(gate + bias) * (up + bias) [M, N / 2] as planes of R rows (N % 128 == 0), nothing is written to part (which may then be null) */
  uint16_t* glu_planes_out; const float* glu_bias;
  const float* wscale;      /* fused SwiGLU epilogue of an fp8 image only: per-row scales [N] (applied to the sums before the bias) */
} mi355_rows_gemm_args;
int mi355_rows_gemm(const mi355_rows_gemm_args* a, void* stream);
int32_t mi355_rows_kgroups(int32_t N, int32_t K);
/* fp32 [N, K] (host) -> tile image: element e of (tile t, k step s, group g, row i) = W[16 t + i][64 s + 16 g + e], stored
 * [ceil(N / 16)][K / 64][4][16][16]; rows past N are zero.  out: ceil(N / 16) * 16 * K 16-bit elements.  dtype: MI355_W_BF16 / MI355_W_F16 */
int mi355_pack_tiles16_host(const float* w_host, int64_t N, int64_t K, int32_t dtype, uint16_t* out_host);
/* uint8 [N, K] e4m3 codes (mi355_pack_rowmajor_fp8_host) -> the fp8 tile image, same order, one byte per element; out: ceil(N / 16) * 16 * K bytes */
int mi355_pack_tiles8_host(const uint8_t* codes_host, int64_t N, int64_t K, uint8_t* out_host);

typedef struct {
  const float* part; int32_t kgroups; int64_t kg_stride; int32_t ldp;   /* input: sum over g of part[g * kg_stride + m * ldp + n]; kgroups = 1: a plain fp32 matrix */
  int32_t M; int32_t N;
  /* epilogue of mi355_gemv: v = act(sum + bias[n]) * colscale[n] + res[m, n]; v *= out_scale;  glu = 1: columns are (gate, up) pairs and
     v[m, n / 2] = silu(gate + b) * (up + b) * out_scale */
  const float* bias; int32_t post_act; float post_slope; const float* colscale; const float* res; int32_t ldr; float out_scale; int32_t glu;
  const float* wscale;      /* nullable: per-column scale of the SUM, before the bias (the per-row dequantisation scales of an fp8 image) */
  float* y; int32_t ldy;    /* nullable: the finished row as fp32 (may alias res) */
  float* y2; int32_t ldy2; int32_t split; int32_t y2_dtype;   /* nullable: columns >= split go to y2[m, n - split] (the KV-cache slot, MI355_KV_*) */
  /* optional normalisation of the finished row over its N (glu: N / 2) outputs, feeding yn / planes: 1 = LayerNorm, 2 = RMSNorm */
  int32_t norm; const float* norm_weight; const float* norm_bias; float norm_eps;
  float* yn; int32_t ldyn;  /* nullable: the (normalised) row as fp32 (a stack's final norm) */
  uint16_t* planes; int32_t R; int32_t planes_dtype;   /* nullable: the (normalised) row as planes of R rows in MI355_W_BF16 / MI355_W_F16 (row length % 64 == 0) */
} mi355_rows_finish_args;
int mi355_rows_finish(const mi355_rows_finish_args* a, void* stream);

/* x [B, d_model] (updated in place: the residual stream), ws: workspace of B * (2 * heads * dh + d_ff + 2 * kv_heads * dh) floats, out (nullable) [B, d_model]
 * receives the final-normed hidden state when final_norm_w is set.  offset = rows already in the KV caches. */
int mi355_stack_decode_step(const mi355_stack_desc* d, float* x, int32_t B, int32_t offset, float* ws, float* out, void* stream);
/* bytes of mi355_stack_desc.rows_ws a step of B (9..64) sequences needs (planes of the three GEMM inputs + the largest set of partial slabs) */
int64_t mi355_stack_rows_ws_bytes(const mi355_stack_desc* d, int32_t B);

#ifdef __cplusplus
}
#endif
#endif /* MI355AUDIO_H */
