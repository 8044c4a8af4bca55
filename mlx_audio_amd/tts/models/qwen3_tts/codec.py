"""Qwen3-TTS speech-tokenizer decoder (codes -> 24 kHz waveform) on MI355X: host schedule over the HIP kernels.

Mirrors ``Qwen3TTSSpeechTokenizerDecoder.__call__`` / ``chunked_decode`` (``tts/models/qwen3_tts/speech_tokenizer.py:786-954``),
with the reference's op-by-op graph collapsed into:
  * split RVQ decode (1 semantic + 15 acoustic codebooks, :423-590): two ``embed_sum`` launches (sum of codebook rows, straight from
    the int codes) and two 1x1 projection GEMMs, the second accumulating onto the first;
  * ``pre_conv`` and every other causal conv (:32-80): conv_gemm with left padding (K-1)*dil -- no padded copy;
  * the 8-layer transformer (:163-420): ``lm.stack.TransformerStack`` (RMSNorm, rotate-half RoPE, flash attention, SwiGLU,
    LayerScale folded into the GEMM epilogues);
  * ConvNeXt upsamplers (:130-160): K == stride transposed conv as one GEMM with a polyphase store, depthwise k7 conv kernel,
    LayerNorm, pointwise GEMMs with GELU / gamma / residual in the epilogues;
  * the SnakeBeta decoder blocks (:593-700): SnakeBeta is the PROLOGUE of the conv that consumes it (alpha = exp(log alpha) and
    1 / (exp(log beta) + 1e-9) precomputed once at load), transposed convs (K = 2 stride, trim right) run polyphase, residual adds are
    epilogues, the final clip is a clamp of the 1-channel output.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .... import ops
from ....lm.stack import StackConfig, TransformerStack, make_lin
from ....ops import ACT_GELU, ACT_NONE, ACT_SNAKE, PackedConv, round_up
from .config import Qwen3TTSTokenizerDecoderConfig


def codec_stack_config(cfg: Qwen3TTSTokenizerDecoderConfig) -> StackConfig:
    return StackConfig(d_model=cfg.hidden_size, n_layers=cfg.num_hidden_layers, n_heads=cfg.num_attention_heads,
                       n_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, d_ff=cfg.intermediate_size, norm="rms",
                       norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_pos=cfg.max_position_embeddings,
                       attn_bias=cfg.attention_bias, mlp="swiglu", layer_scale=True, causal=True, window=0, final_norm=True)


def canonical_stack_weights(w: Dict[str, torch.Tensor], prefix: str, n_layers: int) -> Dict[str, torch.Tensor]:
    """``<prefix>layers.N.self_attn.q_proj`` ... -> the canonical names of ``lm.stack`` (speech_tokenizer.py:255-352 module paths)."""
    m = {"self_attn.q_proj": "wq", "self_attn.k_proj": "wk", "self_attn.v_proj": "wv", "self_attn.o_proj": "wo",
         "mlp.gate_proj": "w_gate", "mlp.up_proj": "w_up", "mlp.down_proj": "w_down", "input_layernorm": "attn_norm",
         "post_attention_layernorm": "mlp_norm"}
    out = {}
    for i in range(n_layers):
        for src, dst in m.items():
            for suf in ("weight", "bias"):
                k = f"{prefix}layers.{i}.{src}.{suf}"
                if k in w:
                    out[f"layers.{i}.{dst}.{suf}"] = w[k]
        for src, dst in (("self_attn_layer_scale.scale", "ls1"), ("mlp_layer_scale.scale", "ls2"), ("self_attn.q_norm.weight", "q_norm.weight"),
                         ("self_attn.k_norm.weight", "k_norm.weight")):
            k = f"{prefix}layers.{i}.{src}"
            if k in w:
                out[f"layers.{i}.{dst}"] = w[k]
    if prefix + "norm.weight" in w:
        out["final_norm.weight"] = w[prefix + "norm.weight"]
    return out


class _Snake:
    """exp(log alpha) and 1 / (exp(log beta) + 1e-9), zero-padded to a multiple of 32 channels (conv_gemm prologue operands)."""

    def __init__(self, alpha: torch.Tensor, beta: torch.Tensor, device):
        c = alpha.numel()
        cp = round_up(c, 32)
        a = torch.ones(cp)
        b = torch.zeros(cp)
        a[:c] = torch.exp(alpha.float())
        b[:c] = 1.0 / (torch.exp(beta.float()) + 1e-9)
        self.alpha, self.inv_beta = a.to(device), b.to(device)


class Qwen3CodecStream:
    """Carried state of ``streaming_step``: the pre-transformer's KV caches and the left-context rows of every causal conv / transposed conv
    (the reference keeps them inside its modules: ``CausalConv1d._buffer``, the transformer cache; speech_tokenizer.py:871-930)."""

    def __init__(self, caches, batch: int):
        self.caches, self.batch = caches, batch
        self.hist: Dict[str, torch.Tensor] = {}
        self.frames = 0


class Qwen3CodecDecoder:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: Qwen3TTSTokenizerDecoderConfig, device="cuda:0", precision: int = 2):
        ops.require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        dev = self.device
        w = {k: v.detach().to(torch.bfloat16).to(torch.float32).cpu() for k, v in weights.items() if v.is_floating_point()}
        self.total_upsample = 1
        for r in list(cfg.upsample_rates) + list(cfg.upsampling_ratios):
            self.total_upsample *= r

        def conv(name):
            return ops.pack_conv(w[name + ".weight"], w.get(name + ".bias"), dev)

        def convT(name, stride):
            return ops.pack_conv_transpose(w[name + ".weight"], w.get(name + ".bias"), stride, dev)

        # --- split RVQ: one stacked table per group + slot offsets
        ns = cfg.num_semantic_quantizers
        self.rvq = []
        for pfx, n in (("quantizer.rvq_first", ns), ("quantizer.rvq_rest", cfg.num_quantizers - ns)):
            if n == 0:
                continue
            tabs = [w[f"{pfx}.vq.layers.{i}.codebook.embed.weight"] for i in range(n)]
            table = torch.cat(tabs, 0).contiguous().to(dev)
            offs = torch.tensor([i * cfg.codebook_size for i in range(n)], dtype=torch.int32, device=dev)
            self.rvq.append((table, offs, conv(pfx + ".output_proj"), n))
        self.pre_conv = conv("pre_conv.conv")
        self.in_proj = make_lin(w["pre_transformer.input_proj.weight"], w.get("pre_transformer.input_proj.bias"), dev)
        self.out_proj = make_lin(w["pre_transformer.output_proj.weight"], w.get("pre_transformer.output_proj.bias"), dev)
        self.stack = TransformerStack(canonical_stack_weights(w, "pre_transformer.", cfg.num_hidden_layers), codec_stack_config(cfg), device=dev,
                                      precision=precision)
        self.ups = []
        for i, f in enumerate(cfg.upsampling_ratios):
            p = f"upsample.{i}.1"
            self.ups.append(dict(f=f, convT=convT(f"upsample.{i}.0.conv", f), dw_w=w[p + ".dwconv.conv.weight"][:, :, 0].contiguous().to(dev),
                                 dw_b=w[p + ".dwconv.conv.bias"].to(dev), ln_w=w[p + ".norm.weight"].to(dev), ln_b=w[p + ".norm.bias"].to(dev),
                                 pw1=conv(p + ".pwconv1"), pw2=conv(p + ".pwconv2"), gamma=w[p + ".gamma"].to(dev)))
        self.init_conv = conv("decoder.0.conv")
        self.blocks = []
        for bi, rate in enumerate(cfg.upsample_rates):
            p = f"decoder.{bi + 1}.block"
            units = []
            for ui, dil in enumerate((1, 3, 9)):
                u = f"{p}.{ui + 2}"
                units.append(dict(dil=dil, s1=_Snake(w[u + ".act1.alpha"], w[u + ".act1.beta"], dev), c1=conv(u + ".conv1.conv"),
                                  s2=_Snake(w[u + ".act2.alpha"], w[u + ".act2.beta"], dev), c2=conv(u + ".conv2.conv")))
            self.blocks.append(dict(rate=rate, snake=_Snake(w[p + ".0.alpha"], w[p + ".0.beta"], dev), up=convT(p + ".1.conv", rate),
                                    cout=cfg.decoder_dim // (2 ** (bi + 1)), units=units))
        n = len(cfg.upsample_rates)
        self.out_snake = _Snake(w[f"decoder.{n + 1}.alpha"], w[f"decoder.{n + 1}.beta"], dev)
        self.out_conv = conv(f"decoder.{n + 2}.conv")

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _snake_conv(self, x, sn: Optional[_Snake], pc: PackedConv, y, *, dil=1, res=None, up=None, lout=None):
        k = pc.k
        kw = dict(dil=dil, pad=(k - 1) * dil, res=res, precision=self.precision, up=up, lout=lout)
        if sn is not None:
            kw.update(pre_act=ACT_SNAKE, pre_alpha=sn.alpha, pre_inv_beta=sn.inv_beta)
        return ops.conv_gemm(x, pc, y, **kw)

    def _convT(self, x, sn: Optional[_Snake], pc: PackedConv, stride: int, cout: int):
        """Causal ConvTranspose1d (K = taps * stride) with the right trim: polyphase stride-1 conv with `taps` taps and stride*cout
        GEMM columns; GEMM row u, column r*cout + co -> out[u*stride + r]; rows >= Lin*stride are the trimmed tail."""
        B, Lin, _ = x.shape
        taps = pc.k
        y = self._f(B, Lin * stride, cout)
        up = dict(s=stride, p=0, cout=cout, lout=Lin * stride)
        kw = dict(pad=taps - 1, lout=Lin + taps - 1, up=up, precision=self.precision)
        if sn is not None:
            kw.update(pre_act=ACT_SNAKE, pre_alpha=sn.alpha, pre_inv_beta=sn.inv_beta)
        ops.conv_gemm(x, pc, y, **kw)
        return y

    def dequantize(self, codes: torch.Tensor) -> torch.Tensor:
        """codes int32 [B, Q, N] (device) -> [B, N, codebook_dim]."""
        cfg = self.cfg
        B, Q, N = codes.shape
        out = self._f(B, N, cfg.codebook_dim)
        q0 = 0
        for gi, (table, offs, proj, n) in enumerate(self.rvq):
            ids = codes[:, q0:q0 + n, :].permute(0, 2, 1)  # [B, N, n] view: embed_sum takes arbitrary id strides
            summed = self._f(B, N, table.shape[1])
            ops.embed_sum(table, ids, summed, slot_offset=offs)
            ops.conv_gemm(summed, proj, out, accumulate=gi > 0, precision=self.precision)
            q0 += n
        return out

    def __call__(self, codes: torch.Tensor, return_stages: bool = False):
        """codes int [B, num_quantizers, N] -> audio [B, 1, N * total_upsample] (speech_tokenizer.py:786-835)."""
        cfg = self.cfg
        if codes.shape[1] != cfg.num_quantizers:
            raise ValueError(f"Expected {cfg.num_quantizers} layers of codes, got {codes.shape[1]}")
        codes = codes.to(self.device, torch.int32).contiguous()
        B, _, N = codes.shape
        st = {}
        h = self.dequantize(codes)
        st["dequant"] = h
        x = self._f(B, N, cfg.latent_dim)
        self._snake_conv(h, None, self.pre_conv, x)
        st["pre_conv"] = x
        t = self._f(B, N, cfg.hidden_size)
        ops.conv_gemm(x, self.in_proj.pc, t, precision=self.precision)
        t = self.stack(t)
        h = self._f(B, N, cfg.latent_dim)
        ops.conv_gemm(t, self.out_proj.pc, h, precision=self.precision)
        st["transformer"] = h
        for up in self.ups:
            h = self._convT(h, None, up["convT"], up["f"], cfg.latent_dim)
            Bh, Lh, C = h.shape
            d = self._f(Bh, Lh, C)
            ops.dwconv(h, up["dw_w"], up["dw_b"], d, pad=6)
            ops.layernorm(d, d, weight=up["ln_w"], bias=up["ln_b"], eps=1e-6)
            m = self._f(Bh, Lh, 4 * C)
            ops.conv_gemm(d, up["pw1"], m, post_act=ACT_GELU, precision=self.precision)
            ops.conv_gemm(m, up["pw2"], h, colscale=up["gamma"], res=h, precision=self.precision)
        st["upsampled"] = h
        wav = self._f(B, h.shape[1], cfg.decoder_dim)
        self._snake_conv(h, None, self.init_conv, wav)
        for bi, blk in enumerate(self.blocks):
            wav = self._convT(wav, blk["snake"], blk["up"], blk["rate"], blk["cout"])
            tmp = torch.empty_like(wav)
            for u in blk["units"]:
                self._snake_conv(wav, u["s1"], u["c1"], tmp, dil=u["dil"])
                self._snake_conv(tmp, u["s2"], u["c2"], wav, res=wav)
            st[f"block{bi}"] = wav
        out = self._f(B, wav.shape[1], 1)
        self._snake_conv(wav, self.out_snake, self.out_conv, out)
        audio = out.transpose(1, 2).clamp_(-1.0, 1.0)
        return (audio, st) if return_stages else audio

    # ------------------------------------------------------------------ incremental decode (speech_tokenizer.py:882-930)
    def reset_streaming_state(self) -> None:
        self._stream = None

    def new_stream(self, batch: int = 1) -> "Qwen3CodecStream":
        return Qwen3CodecStream(self.stack.make_cache(), batch)

    def _ctx_conv(self, st: "Qwen3CodecStream", key: str, x, sn: Optional[_Snake], pc: PackedConv, y, *, dil=1, res=None):
        """``_snake_conv`` on a chunk: the (K - 1) * dil left-context rows come from the stream (zeros before the first chunk = the causal zero padding of
        the one-shot pass: SnakeBeta(0) = 0) and are refreshed from this chunk's input (``CausalConv1d.step``, speech_tokenizer.py)."""
        h = (pc.k - 1) * dil
        if h == 0:
            return self._snake_conv(x, sn, pc, y, dil=dil, res=res)
        B, L, C = x.shape
        ext = torch.empty((B, h + L, C), dtype=torch.float32, device=self.device)
        hist = st.hist.get(key)
        if hist is None:
            ext[:, :h].zero_()
        else:
            ext[:, :h].copy_(hist)
        ext[:, h:].copy_(x)
        kw = dict(dil=dil, pad=0, lout=L, res=res, precision=self.precision)
        if sn is not None:
            kw.update(pre_act=ACT_SNAKE, pre_alpha=sn.alpha, pre_inv_beta=sn.inv_beta)
        ops.conv_gemm(ext, pc, y, **kw)
        st.hist[key] = ext[:, L:, :].clone()
        return y

    def _ctx_convT(self, st: "Qwen3CodecStream", key: str, x, sn: Optional[_Snake], pc: PackedConv, stride: int, cout: int):
        """``_convT`` on a chunk: a transposed conv with ``taps`` = K / stride taps reads taps - 1 earlier input rows (none when K == stride)."""
        taps = pc.k
        if taps == 1:
            return self._convT(x, sn, pc, stride, cout)
        B, L, C = x.shape
        h = taps - 1
        ext = torch.zeros((B, h + L, C), dtype=torch.float32, device=self.device)
        if key in st.hist:
            ext[:, :h].copy_(st.hist[key])
        ext[:, h:].copy_(x)
        st.hist[key] = ext[:, L:, :].clone()
        y_ext = self._convT(ext, sn, pc, stride, cout)
        return y_ext[:, h * stride:, :]   # the first rows belong to earlier chunks (recomputed without THEIR history: dropped)

    def streaming_step(self, codes: torch.Tensor, st: Optional["Qwen3CodecStream"] = None, return_stages: bool = False):
        """codes int [B, num_quantizers, n]: the NEXT n frames of the stream -> their audio [B, 1, n * total_upsample].  The concatenation of successive
        calls is the one-shot decode of the concatenated codes (every operator is causal; tests/test_qwen3_codec_gpu.py).  ``st`` = a state from
        ``new_stream`` (default: the object-held one, reset by ``reset_streaming_state`` -- the reference's calling convention)."""
        cfg = self.cfg
        if st is None:
            st = getattr(self, "_stream", None)
            if st is None or st.batch != codes.shape[0]:
                st = self._stream = self.new_stream(codes.shape[0])
        if codes.shape[1] != cfg.num_quantizers:
            raise ValueError(f"Expected {cfg.num_quantizers} layers of codes, got {codes.shape[1]}")
        codes = codes.to(self.device, torch.int32).contiguous()
        B, _, N = codes.shape
        assert B == st.batch and N >= 1
        stg = {}
        h = self.dequantize(codes)
        stg["dequant"] = h
        x = self._f(B, N, cfg.latent_dim)
        self._ctx_conv(st, "pre_conv", h, None, self.pre_conv, x)
        stg["pre_conv"] = x
        t = self._f(B, N, cfg.hidden_size)
        ops.conv_gemm(x, self.in_proj.pc, t, precision=self.precision)
        t = self.stack(t, st.caches)    # appends the chunk's positions to the stream's KV caches
        h = self._f(B, N, cfg.latent_dim)
        ops.conv_gemm(t, self.out_proj.pc, h, precision=self.precision)
        stg["transformer"] = h
        for i, up in enumerate(self.ups):
            h = self._ctx_convT(st, f"upT{i}", h, None, up["convT"], up["f"], cfg.latent_dim)
            Bh, Lh, C = h.shape
            kd = up["dw_w"].shape[1] - 1   # depthwise causal conv, K = 7: six rows of history
            ext = torch.zeros((Bh, kd + Lh, C), dtype=torch.float32, device=self.device)
            if f"dw{i}" in st.hist:
                ext[:, :kd].copy_(st.hist[f"dw{i}"])
            ext[:, kd:].copy_(h)
            st.hist[f"dw{i}"] = ext[:, Lh:, :].clone()
            d = self._f(Bh, Lh, C)
            ops.dwconv(ext, up["dw_w"], up["dw_b"], d, pad=0)
            ops.layernorm(d, d, weight=up["ln_w"], bias=up["ln_b"], eps=1e-6)
            m = self._f(Bh, Lh, 4 * C)
            ops.conv_gemm(d, up["pw1"], m, post_act=ACT_GELU, precision=self.precision)
            h = h.contiguous()
            ops.conv_gemm(m, up["pw2"], h, colscale=up["gamma"], res=h, precision=self.precision)
        stg["upsampled"] = h
        wav = self._f(B, h.shape[1], cfg.decoder_dim)
        self._ctx_conv(st, "init", h, None, self.init_conv, wav)
        for bi, blk in enumerate(self.blocks):
            wav = self._ctx_convT(st, f"blkT{bi}", wav, blk["snake"], blk["up"], blk["rate"], blk["cout"]).contiguous()
            tmp = torch.empty_like(wav)
            for ui, u in enumerate(blk["units"]):
                self._ctx_conv(st, f"b{bi}u{ui}c1", wav, u["s1"], u["c1"], tmp, dil=u["dil"])
                self._ctx_conv(st, f"b{bi}u{ui}c2", tmp, u["s2"], u["c2"], wav, res=wav)
            stg[f"block{bi}"] = wav
        out = self._f(B, wav.shape[1], 1)
        self._ctx_conv(st, "out", wav, self.out_snake, self.out_conv, out)
        st.frames += N
        audio = out.transpose(1, 2).clamp_(-1.0, 1.0)
        return (audio, stg) if return_stages else audio

    def chunked_decode(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        """speech_tokenizer.py:930-954: chunks of ``chunk_size`` code frames with ``left_context_size`` frames of context."""
        wavs: List[torch.Tensor] = []
        start = 0
        while start < codes.shape[-1]:
            end = min(start + chunk_size, codes.shape[-1])
            ctx = left_context_size if start - left_context_size > 0 else start
            wav = self(codes[..., start - ctx:end])
            wavs.append(wav[..., ctx * self.total_upsample:])
            start = end
        return torch.cat(wavs, dim=-1)
