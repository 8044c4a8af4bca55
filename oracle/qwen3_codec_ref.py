"""PyTorch-CPU restatement of the Qwen3-TTS speech-tokenizer DECODER (TEST ORACLE, not product).

Follows ``tts/models/qwen3_tts/speech_tokenizer.py`` of the reference:
  * :32-104   CausalConv1d (left pad (K-1)*dil), CausalTransposeConv1d (trim right K - stride)
  * :107-127  SnakeBeta  x + 1/(exp(beta) + 1e-9) * sin^2(x * exp(alpha))
  * :130-160  ConvNeXtBlock (causal depthwise k7 -> LayerNorm(1e-6) -> Linear 4x -> GELU(erf) -> Linear -> gamma * x + residual)
  * :163-420  DecoderRMSNorm / LayerScale / rotate-half RoPE / DecoderAttention / DecoderMLP / DecoderTransformer
              (full causal mask: ``sliding_window`` is stored but never applied, :242, 400-404) -- via oracle.lm_ref.StackRef
  * :423-590  EuclideanCodebook / VectorQuantization / ResidualVectorQuantizer / SplitResidualVectorQuantizer.decode
  * :593-790  DecoderResidualUnit, DecoderBlockUpsample, DecoderBlock, DecoderInitialConv, DecoderOutputSnake / Conv
  * :786-880  Qwen3TTSSpeechTokenizerDecoder.__call__ and chunked_decode (:930-954)
Parameter names are the reference's module paths (``decoder.`` prefix dropped), conv weights in the MLX layout (C_out, K, C_in/groups);
ConvTranspose1d weights (C_out, K, C_in) as ``mx.conv_transpose1d`` takes them.
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files for
the Qwen3-TTS codec decoder (imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) on a seeded tiny checkpoint, and
tests/test_reference_fixtures_cpu.py holds this oracle to the result -- ``Qwen3TTSSpeechTokenizerDecoder.__call__`` and ``chunked_decode``: 9e-7 of the waveform.
The reference's own tests pin shapes / token-rule cases only (reproduced in tests/test_oracle_golden.py); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .lm_ref import StackConfig, StackRef

Tensor = torch.Tensor


def codec_stack_config(cfg) -> StackConfig:
    return StackConfig(d_model=cfg.hidden_size, n_layers=cfg.num_hidden_layers, n_heads=cfg.num_attention_heads,
                       n_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, d_ff=cfg.intermediate_size, norm="rms",
                       norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_pos=cfg.max_position_embeddings,
                       attn_bias=cfg.attention_bias, mlp="swiglu", layer_scale=True, causal=True, window=0, final_norm=True)


def canonical_stack_weights(w: Dict[str, Tensor], prefix: str, n_layers: int) -> Dict[str, Tensor]:
    """pre_transformer.layers.N.{self_attn.q_proj,...} -> canonical stack names (oracle/lm_ref.py header)."""
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
        out[f"layers.{i}.ls1"] = w[f"{prefix}layers.{i}.self_attn_layer_scale.scale"]
        out[f"layers.{i}.ls2"] = w[f"{prefix}layers.{i}.mlp_layer_scale.scale"]
    out["final_norm.weight"] = w[prefix + "norm.weight"]
    return out


class Qwen3CodecDecoderRef:
    def __init__(self, weights: Dict[str, Tensor], cfg, dtype=torch.float32, param_dtype=torch.bfloat16):
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: v.to(param_dtype).to(dtype) for k, v in weights.items()}
        self.stack = StackRef(canonical_stack_weights(weights, "pre_transformer.", cfg.num_hidden_layers), codec_stack_config(cfg), dtype, param_dtype)
        self.total_upsample = 1
        for r in list(cfg.upsample_rates) + list(cfg.upsampling_ratios):
            self.total_upsample *= r

    # NLC helpers -------------------------------------------------------------------------------------------
    def _conv(self, x: Tensor, name: str, dil: int = 1, groups: int = 1) -> Tensor:
        """CausalConv1d: x [B, L, C]."""
        w = self.w[name + ".weight"]
        k = w.shape[1]
        xp = F.pad(x.transpose(1, 2), ((k - 1) * dil, 0))
        return F.conv1d(xp, w.permute(0, 2, 1), self.w.get(name + ".bias"), dilation=dil, groups=groups).transpose(1, 2)

    def _convT(self, x: Tensor, name: str, stride: int) -> Tensor:
        """ConvTranspose1d (padding 0) then trim K - stride on the right."""
        w = self.w[name + ".weight"]  # (C_out, K, C_in)
        k = w.shape[1]
        y = F.conv_transpose1d(x.transpose(1, 2), w.permute(2, 0, 1), self.w.get(name + ".bias"), stride=stride)
        trim = k - stride
        if trim > 0:
            y = y[:, :, :-trim]
        return y.transpose(1, 2)

    def _snake(self, x: Tensor, name: str) -> Tensor:
        a, b = torch.exp(self.w[name + ".alpha"]), torch.exp(self.w[name + ".beta"])
        return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2

    def _lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    def dequantize(self, codes: Tensor) -> Tensor:
        """SplitResidualVectorQuantizer.decode: codes [B, Q, N] -> [B, N, codebook_dim]."""
        ns = self.cfg.num_semantic_quantizers

        def rvq(pfx, cs):
            q = 0
            for i in range(cs.shape[1]):
                q = q + self.w[f"{pfx}.vq.layers.{i}.codebook.embed.weight"][cs[:, i]]
            return F.conv1d(q.transpose(1, 2), self.w[pfx + ".output_proj.weight"].permute(0, 2, 1)).transpose(1, 2)

        out = rvq("quantizer.rvq_first", codes[:, :ns])
        if codes.shape[1] > ns:
            out = out + rvq("quantizer.rvq_rest", codes[:, ns:])
        return out

    def __call__(self, codes: Tensor, return_stages: bool = False):
        """codes int [B, num_quantizers, N] -> audio [B, 1, N * total_upsample] clipped to +-1."""
        cfg = self.cfg
        st = {}
        h = self.dequantize(codes).to(self.dtype)
        st["dequant"] = h
        h = self._conv(h, "pre_conv.conv")
        st["pre_conv"] = h
        h = self._lin(h, "pre_transformer.input_proj")
        h = self.stack(h)
        h = self._lin(h, "pre_transformer.output_proj")
        st["transformer"] = h
        for i, f in enumerate(cfg.upsampling_ratios):
            h = self._convT(h, f"upsample.{i}.0.conv", f)
            p = f"upsample.{i}.1"
            r = h
            y = self._conv(h, p + ".dwconv.conv", groups=h.shape[-1])
            y = F.layer_norm(y, (y.shape[-1],), self.w[p + ".norm.weight"], self.w[p + ".norm.bias"], 1e-6)
            y = self._lin(F.gelu(self._lin(y, p + ".pwconv1")), p + ".pwconv2")
            h = r + self.w[p + ".gamma"] * y
        st["upsampled"] = h
        wav = self._conv(h, "decoder.0.conv")
        for bi, rate in enumerate(cfg.upsample_rates):
            p = f"decoder.{bi + 1}.block"
            wav = self._snake(wav, p + ".0")
            wav = self._convT(wav, p + ".1.conv", rate)
            for ui, dil in enumerate((1, 3, 9)):
                u = f"{p}.{ui + 2}"
                r = wav
                y = self._conv(self._snake(wav, u + ".act1"), u + ".conv1.conv", dil=dil)
                y = self._conv(self._snake(y, u + ".act2"), u + ".conv2.conv")
                wav = y + r
            st[f"block{bi}"] = wav
        n = len(cfg.upsample_rates)
        wav = self._snake(wav, f"decoder.{n + 1}")
        wav = self._conv(wav, f"decoder.{n + 2}.conv")
        out = torch.clip(wav.transpose(1, 2), -1.0, 1.0)
        return (out, st) if return_stages else out

    def chunked_decode(self, codes: Tensor, chunk_size: int = 300, left_context_size: int = 25) -> Tensor:
        """speech_tokenizer.py:930-954."""
        wavs, start = [], 0
        while start < codes.shape[-1]:
            end = min(start + chunk_size, codes.shape[-1])
            ctx = left_context_size if start - left_context_size > 0 else start
            wav = self(codes[..., start - ctx:end])
            wavs.append(wav[..., ctx * self.total_upsample:])
            start = end
        return torch.cat(wavs, dim=-1)
