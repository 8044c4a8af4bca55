"""PyTorch-CPU restatement of KittenTTS's forward pass (TEST ORACLE, not product).

KittenTTS is Kokoro's StyleTTS2 / iSTFTNet stack with its own dimensions (``max_conv_dim``, ``decoder_out_dim``, ``asr_res_dim`` instead of
Kokoro's fixed 1024 / 512 / 64), a tanh-GELU ALBERT and -- because the checkpoint is converted from an int8 ONNX export -- *dynamic uint8
fake quantisation* of the inputs of the modules listed in ``activation_quant_modules``.  Follows:

  * ``tts/models/kitten_tts/kitten_tts.py:376-413``  Model.__call__ (token ids -> waveform), :121-174 KittenDecoder,
                                                     :177-330 KittenAlbert* (quant sites :108-110,:131-133,:255-268,:296-298), :291-299 flag rule
  * ``tts/models/kitten_tts/quant.py:4-24``          fake_quant_dynamic_u8 (oracle.kokoro_ref.fake_quant_dynamic_u8)
  * ``tts/models/kitten_tts/modules.py``             quant sites :19 (LinearNorm), :80 (AdaLayerNorm), :155/:178/:201/:224 (LSTM), :371/:380 (F0 / N proj)
  * ``tts/models/kitten_tts/istftnet.py``            quant sites :131 (ConvWeighted), :336 (AdaIN1d), :711 (l_linear), :815 (noise_convs);
                                                     per-index Snake parameters :379-384; no phase unwrap :524-527 (the identity for |phase| <= 1);
                                                     SineGen keeps ``upsample_scale`` as an mx.array :572, so its coarse phase grid has 2F + 1 points

Everything that is the same module as Kokoro's is ``oracle.kokoro_ref``'s function, called with a parameter view that carries the quantised-module
list (``P.quant`` implements the reference's flag rule); the quantisation hooks there are inert for Kokoro.

Parity status: **pinned to the reference's own modules**: ``tests/golden/make_reference_fixtures.py`` runs the reference's KittenTTS source
files (imported from /root/reference, unmodified) on a seeded synthetic checkpoint -- once without and once with the converter's
``activation_quant_modules`` -- over a numpy stand-in for MLX (``tests/golden/mlx_shim.py``), and ``tests/test_reference_fixtures_cpu.py`` holds this
oracle to those fixtures: durations exact, the quantisation flag rule identical on all 423 modules of the reference's module tree, intermediates
2e-7..2e-6 and waveform 3e-6 relative RMS without quantisation; with it, 1e-7 where no grid step flips and the flip-noise band otherwise.  The
reference's own tests only construct the model (tts/tests/test_models.py:416-492; reproduced in tests/test_kitten_cpu.py); MLX's kernels
themselves are not exercised by the stand-in.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import kokoro_ref as K
from .kokoro_ref import P, fake_quant_dynamic_u8, layer_norm, linear

Tensor = torch.Tensor


def gelu_tanh_onnx(x: Tensor) -> Tensor:
    """kitten_tts.py:244-263: the ONNX export's tanh GELU with its literal constants."""
    return 0.5 * x * (1.0 + torch.tanh(0.7978846 * (x + 0.044715 * (x ** 3))))


def kitten_albert(p: P, input_ids: Tensor, attn_mask01: Tensor, cfg: dict) -> Tensor:
    """KittenAlbert (kitten_tts.py:316-330) -> sequence_output [B, T, hidden]."""
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh = cfg["num_attention_heads"]
    B, T = input_ids.shape
    e = p.sub("embeddings")
    emb = e("word_embeddings.weight")[input_ids] + e("position_embeddings.weight")[torch.arange(T)][None] \
        + e("token_type_embeddings.weight")[torch.zeros_like(input_ids)]
    h = layer_norm(emb, e("LayerNorm.weight"), e("LayerNorm.bias"), eps)
    add_mask = (1.0 - attn_mask01.to(h.dtype))[:, None, None, :] * -10000.0
    enc = p.sub("encoder")
    h = linear(enc.sub("embedding_hidden_mapping_in"), enc.fq(h))
    groups = cfg.get("num_hidden_groups", 1)
    hd = h.shape[-1] // nh
    for i in range(cfg["num_hidden_layers"]):
        g = int(i / (cfg["num_hidden_layers"] / groups))
        grp = enc.sub(f"albert_layer_groups.{g}")
        for j in range(cfg.get("inner_group_num", 1)):
            lay = grp.sub(f"albert_layers.{j}")
            att = lay.sub("attention")

            def split(t):
                return t.view(B, T, nh, hd).permute(0, 2, 1, 3)
            hq = att.fq(h)
            q, k, v = split(linear(att.sub("query"), hq)), split(linear(att.sub("key"), hq)), split(linear(att.sub("value"), hq))
            sc = q @ k.transpose(-1, -2) / math.sqrt(hd) + add_mask
            ctx = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, nh * hd)
            a = layer_norm(linear(att.sub("dense"), att.fq(ctx)) + h, att("LayerNorm.weight"), att("LayerNorm.bias"), eps)
            f = gelu_tanh_onnx(linear(lay.sub("ffn"), lay.fq(a)))
            f = linear(lay.sub("ffn_output"), lay.fq(f))
            h = layer_norm(f + a, lay("full_layer_layer_norm.weight"), lay("full_layer_layer_norm.bias"), eps)
    return h


class KittenRef:
    """Oracle for ``Model.__call__`` (kitten_tts.py:376-413), batch 1 like the reference."""

    def __init__(self, weights: Dict[str, Tensor], config: dict, dtype=torch.float32, param_dtype=torch.float32,
                 quant_modules: Optional[Sequence[str]] = None):
        self.cfg = config
        if quant_modules is None:
            quant_modules = config.get("activation_quant_modules") or ()
        self.p = P(weights, "", dtype, param_dtype, quant_modules)
        self.dtype = dtype

    def durations(self, input_ids: Tensor, ref_s: Tensor, speed: float = 1.0):
        """Returns (pred_dur int32 [T], d [1, T, hid + style], raw duration float [T])."""
        cfg, p = self.cfg, self.p
        ids = input_ids.view(1, -1)
        s = ref_s.to(self.dtype)[:, 128:]
        bert_out = kitten_albert(p.sub("bert"), ids, torch.ones_like(ids), cfg["plbert"])
        be = p.sub("bert_encoder")
        d_en = linear(be, be.fq(bert_out)).transpose(1, 2)
        d = K.duration_encoder(p.sub("predictor.text_encoder"), d_en, s, cfg["n_layer"])
        x = K.bilstm(p.sub("predictor.lstm"), d)
        dp = p.sub("predictor.duration_proj")
        logits = linear(dp.sub("linear_layer"), dp.fq(x))
        dur = torch.sigmoid(logits).sum(dim=-1) / speed
        pred = torch.clamp(torch.round(dur), min=1).to(torch.int32)[0]  # kitten_tts.py:398: no upper clip
        return pred, d, dur[0]

    def forward(self, input_ids: Tensor, ref_s: Tensor, speed: float = 1.0, rand_ini: Optional[np.ndarray] = None,
                noise: Optional[np.ndarray] = None, pred_dur: Optional[Tensor] = None, noise_seed: int = 1234,
                return_intermediates=False, f0_override: Optional[Tensor] = None, n_override: Optional[Tensor] = None):
        """input_ids: LongTensor [T] INCLUDING the leading / trailing 0 tokens; ref_s [1, 256]."""
        cfg, p = self.cfg, self.p
        with torch.no_grad():
            ids = input_ids.view(1, -1)
            ref_s = ref_s.to(self.dtype)
            s_pred = ref_s[:, 128:]
            pd, d, raw = self.durations(input_ids, ref_s, speed)
            if pred_dur is None:
                pred_dur = pd
            idx = torch.repeat_interleave(torch.arange(ids.shape[1]), pred_dur.to(torch.long))
            Fr = idx.numel()
            en = d.transpose(1, 2)[:, :, idx]
            pr = p.sub("predictor")
            x = K.bilstm(pr.sub("shared"), en.transpose(1, 2))
            f0 = x.transpose(1, 2)
            nn_ = x.transpose(1, 2)
            for i in range(3):
                f0 = K.adain_resblk1d(pr.sub(f"F0.{i}"), f0, s_pred, upsample=pr.has(f"F0.{i}.pool.weight_v"))
                nn_ = K.adain_resblk1d(pr.sub(f"N.{i}"), nn_, s_pred, upsample=pr.has(f"N.{i}.pool.weight_v"))
            f0p, np_ = pr.sub("F0_proj"), pr.sub("N_proj")
            f0 = K.conv1d_mlx(f0p.fq(f0), f0p("weight"), f0p("bias"))[:, 0, :]
            nn_ = K.conv1d_mlx(np_.fq(nn_), np_("weight"), np_("bias"))[:, 0, :]
            if f0_override is not None:
                f0 = f0_override.to(self.dtype)
            if n_override is not None:
                nn_ = n_override.to(self.dtype)
            t_en = K.text_encoder(p.sub("text_encoder"), ids, cfg["n_layer"])
            asr = t_en[:, :, idx]
            if rand_ini is None or noise is None:
                rng = np.random.default_rng(noise_seed)
                up = int(np.prod(cfg["istftnet"]["upsample_rates"])) * cfg["istftnet"]["gen_istft_hop_size"]
                rand_ini = rng.uniform(size=(1, 9)).astype(np.float32)
                noise = rng.standard_normal((1, 2 * Fr * up, 9)).astype(np.float32)
            trace = {} if return_intermediates else None
            audio = K.decoder(p.sub("decoder"), asr, f0, nn_, ref_s[:, :128], cfg["istftnet"], rand_ini, noise, trace, coarse_f32=True)[0]
            if return_intermediates:
                return audio, pred_dur, dict(d=d, en=en, f0=f0, n=nn_, asr=asr, raw_dur=raw, **trace)
            return audio, pred_dur
