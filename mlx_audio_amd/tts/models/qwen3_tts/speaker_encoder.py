"""ECAPA-TDNN speaker encoder of Qwen3-TTS on the device (``tts/models/qwen3_tts/speaker_encoder.py`` of the reference): mel ``[B, T, 128]`` ->
x-vector ``[B, enc_dim]``.  The Base checkpoints carry it; ``Model.extract_speaker_embedding`` (qwen3_tts.py:285-324) feeds its output into the
codec prefix of every cloning prompt (plain ``ref_audio`` and in-context ``ref_audio`` + ``ref_text`` alike, :383-384, :743-746).

How the reference's modules map onto launches (activations stay time-major ``[B, T, C]`` float32, which is what ``nn.Conv1d`` of MLX takes -- the
reference's transposes to ``[B, C, T]`` between layers move nothing here):
  * ``TimeDelayNetBlock`` (:29-58)        -> ``mi355_ecapa_rows`` writes the reflect-padded copy (``reflect_pad_1d`` :11-26), ``mi355_conv_gemm`` does the
                                             dilated conv with the ReLU in its epilogue (k = 1 blocks: the conv alone)
  * ``Res2NetBlock`` (:61-105)            -> chunk i reads a 64-channel SLICE of the tdnn1 output and writes a slice of the concatenated output; the
                                             "chunk + previous output" sum rides on the padding pass (``res=``)
  * ``SqueezeExcitationBlock`` (:108-141) -> ``mi355_time_moments`` (the squeeze), two small Linear launches, then ONE ``mi355_ecapa_rows`` pass that
                                             applies sigmoid(gate), adds the block residual (:180) and writes straight into the block's slice of the
                                             multi-layer concatenation (:296) -- no ``x * se`` tensor, no concatenate
  * ``AttentiveStatisticsPooling`` (:183-229) -> moments + two broadcasts build [x | mean | std]; conv (ReLU) -> tanh pass -> conv -> ``mi355_attentive_pool``
                                             (softmax over time, weighted mean / std) -> ``fc`` (:271-277)
Weights are rounded to bf16 like every conv of this build (the published checkpoints are bf16); accumulation is float32 on hi + lo split activations.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .... import ops
from ....lm.stack import Lin, linear, make_lin
from ....ops import ACT_LEAKY, PackedConv
from .config import Qwen3TTSSpeakerEncoderConfig


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def tiny_speaker_encoder_config() -> Qwen3TTSSpeakerEncoderConfig:
    """The real structure (initial TDNN k 5, three SE-Res2Net blocks with dilations 2 / 3 / 4, MFA, attentive pooling, fc) at a fraction of the width."""
    return Qwen3TTSSpeakerEncoderConfig(mel_dim=16, enc_dim=32, enc_channels=[64, 64, 64, 64, 192], enc_kernel_sizes=[5, 3, 3, 3, 1],
                                        enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=16, enc_res2net_scale=4, enc_se_channels=16)


def make_speaker_encoder_weights(cfg: Qwen3TTSSpeakerEncoderConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded parameters under the reference's names (what its ``sanitize`` returns: no ``speaker_encoder.`` prefix, conv weights (out, K, in)),
    values bf16-representable."""
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin, gain=1.0):
        w[name + ".weight"] = _r16(torch.randn(cout, k, cin, generator=g) * (gain / math.sqrt(k * cin)))
        w[name + ".bias"] = _r16(torch.randn(cout, generator=g) * 0.05)

    ch, ks = cfg.enc_channels, cfg.enc_kernel_sizes
    conv("blocks.0.conv", ch[0], ks[0], cfg.mel_dim, gain=1.4)
    for i in range(1, len(ch) - 1):
        conv(f"blocks.{i}.tdnn1.conv", ch[i], 1, ch[i - 1], gain=1.4)
        sub = ch[i] // cfg.enc_res2net_scale
        for j in range(cfg.enc_res2net_scale - 1):
            conv(f"blocks.{i}.res2net_block.blocks.{j}.conv", sub, ks[i], sub, gain=1.4)
        conv(f"blocks.{i}.tdnn2.conv", ch[i], 1, ch[i], gain=1.4)
        conv(f"blocks.{i}.se_block.conv1", cfg.enc_se_channels, 1, ch[i], gain=2.0)
        conv(f"blocks.{i}.se_block.conv2", ch[i], 1, cfg.enc_se_channels, gain=2.0)
    conv("mfa.conv", ch[-1], ks[-1], ch[-1], gain=1.4)
    conv("asp.tdnn.conv", cfg.enc_attention_channels, 1, 3 * ch[-1], gain=2.0)
    conv("asp.conv", ch[-1], 1, cfg.enc_attention_channels, gain=4.0)   # wide logits: the softmax over time is far from uniform
    conv("fc", cfg.enc_dim, 1, 2 * ch[-1])
    return w


def make_mels(batch: int, frames: int, mel_dim: int, seed: int = 0) -> torch.Tensor:
    """Log-mel-like test input: smooth in time, around -4 with a spread of a few units."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, frames + 8, mel_dim, generator=g)
    x = torch.nn.functional.avg_pool1d(x.transpose(1, 2), 5, 1).transpose(1, 2)[:, :frames]
    return (x * 3.0 - 4.0 + torch.linspace(1.5, -1.5, mel_dim)).contiguous()


class Qwen3TTSSpeakerEncoder:
    def __init__(self, config: Qwen3TTSSpeakerEncoderConfig, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda", precision: int = 2):
        self.config = config
        self.channels = config.enc_channels
        self.device = torch.device(device)
        self.precision = precision
        ch = config.enc_channels
        # every SE-Res2Net block adds its input to its output (:180) and mfa reads their concatenation (:261-267, :296)
        if any(ch[i] != ch[i - 1] for i in range(1, len(ch) - 1)) or ch[-1] != sum(ch[1:-1]):
            raise ValueError(f"speaker encoder: enc_channels {ch} must be equal block widths followed by their sum")
        if any(w % config.enc_res2net_scale for w in ch[1:-1]):
            raise ValueError(f"speaker encoder: block widths {ch[1:-1]} must divide by enc_res2net_scale ({config.enc_res2net_scale})")
        self._ready = False
        if weights is not None:
            self.load_weights(weights)

    # ------------------------------------------------------------------ checkpoint handling
    @staticmethod
    def sanitize(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """``speaker_encoder.py:315-340``: keep ``speaker_encoder.*`` only, strip the prefix, PyTorch conv layout (out, in, K) -> (out, K, in) for every
        3-D ``.weight`` unless the shape heuristic says it already is."""
        from .qwen3_tts import check_array_shape_qwen3

        out = {}
        for k, v in weights.items():
            if not k.startswith("speaker_encoder."):
                continue
            nk = k.replace("speaker_encoder.", "")
            if nk.endswith(".weight") and v.dim() == 3:
                v = v if check_array_shape_qwen3(v) else v.permute(0, 2, 1).contiguous()
            out[nk] = v
        return out

    def load_weights(self, weights, strict: bool = True):
        """``weights``: names without the ``speaker_encoder.`` prefix, conv weights (out, K, in)."""
        ops.require_gpu()
        w = {k: _r16(v.detach().to(torch.float32).cpu()) for k, v in dict(weights).items()}
        dev = self.device
        c = self.config
        used = set()

        def conv(name) -> PackedConv:
            used.update((name + ".weight", name + ".bias"))
            try:
                return ops.pack_conv(w[name + ".weight"], w[name + ".bias"], dev)
            except KeyError as e:
                raise ValueError(f"speaker encoder checkpoint is missing parameter {e.args[0]}") from e

        def lin(name) -> Lin:
            used.update((name + ".weight", name + ".bias"))
            try:
                return make_lin(w[name + ".weight"][:, 0, :], w[name + ".bias"], dev)
            except KeyError as e:
                raise ValueError(f"speaker encoder checkpoint is missing parameter {e.args[0]}") from e

        ch = c.enc_channels
        self.tdnn0 = conv("blocks.0.conv")
        self.blocks = []
        for i in range(1, len(ch) - 1):
            self.blocks.append(dict(
                tdnn1=conv(f"blocks.{i}.tdnn1.conv"),
                res2=[conv(f"blocks.{i}.res2net_block.blocks.{j}.conv") for j in range(c.enc_res2net_scale - 1)],
                tdnn2=conv(f"blocks.{i}.tdnn2.conv"),
                se1=lin(f"blocks.{i}.se_block.conv1"), se2=lin(f"blocks.{i}.se_block.conv2"),
                k=c.enc_kernel_sizes[i], dil=c.enc_dilations[i], width=ch[i]))
        self.mfa = conv("mfa.conv")
        self.asp_tdnn = conv("asp.tdnn.conv")
        self.asp_conv = conv("asp.conv")
        self.fc = lin("fc")
        if strict:
            extra = sorted(set(w) - used)
            if extra:
                raise ValueError(f"speaker encoder: unexpected parameters {extra[:4]}")
        self._ready = True
        return self

    # ------------------------------------------------------------------ forward
    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _tdnn(self, x: torch.Tensor, pc: PackedConv, y: torch.Tensor, dil: int = 1, res: Optional[torch.Tensor] = None):
        """ReLU(conv(reflect_pad(x [+ res]))) into ``y`` (any channel slice)."""
        pad = (pc.k - 1) * dil // 2
        if pad > 0 or res is not None:
            B, T, C = x.shape
            xp = self._f(B, T + 2 * pad, C)
            ops.ecapa_rows(x, xp, pad=pad, res=res)
            x = xp
        ops.conv_gemm(x, pc, y, dil=dil, pad=0, post_act=ACT_LEAKY, post_slope=0.0, precision=self.precision)
        return y

    def _vec_linear(self, v: torch.Tensor, l: Lin, post_act: int = 0) -> torch.Tensor:
        """One row per utterance ``[B, K]`` -> ``[B, N]``: the B rows as ONE item (a GEMV for a single clip, the conv tile otherwise)."""
        B = v.shape[0]
        y = self._f(1, B, l.rm.n)
        linear(v[None], l, y, post_act=post_act, precision=self.precision)
        return y[0]

    def __call__(self, mels: torch.Tensor, stages: Optional[dict] = None) -> torch.Tensor:
        """mels ``[B, T, mel_dim]`` -> speaker embedding ``[B, enc_dim]`` (speaker_encoder.py:279-313)."""
        if not self._ready:
            raise RuntimeError("speaker encoder has no weights: call load_weights()")
        c = self.config
        x = torch.as_tensor(mels, dtype=torch.float32).to(self.device).contiguous()
        if x.dim() != 3 or x.shape[2] != c.mel_dim:
            raise ValueError(f"speaker encoder takes mels [batch, time, {c.mel_dim}], got {tuple(x.shape)}")
        B, T, _ = x.shape
        max_pad = max((k - 1) * d // 2 for k, d in zip(c.enc_kernel_sizes, c.enc_dilations))
        if T <= max_pad:
            raise ValueError(f"speaker encoder: {T} mel frames are too few for reflect padding of {max_pad}")
        ch = c.enc_channels
        h = self._tdnn(x, self.tdnn0, self._f(B, T, ch[0]), dil=c.enc_dilations[0])
        cat = self._f(B, T, ch[-1])            # the SE-Res2Net outputs side by side = the input of mfa
        col = 0
        scale = c.enc_res2net_scale
        for bi, blk in enumerate(self.blocks):
            W = blk["width"]
            sub = W // scale
            a = self._tdnn(h, blk["tdnn1"], self._f(B, T, W))
            r = self._f(B, T, W)
            r[:, :, :sub].copy_(a[:, :, :sub])                                  # chunk 0 passes through (:94-95)
            for i in range(1, scale):
                prev = r[:, :, (i - 1) * sub:i * sub] if i >= 2 else None       # chunk + previous output (:98-101)
                self._tdnn(a[:, :, i * sub:(i + 1) * sub], blk["res2"][i - 1], r[:, :, i * sub:(i + 1) * sub], dil=blk["dil"], res=prev)
            t2 = self._tdnn(r, blk["tdnn2"], self._f(B, T, W))
            mean, _ = ops.time_moments(t2, want_std=False)
            g = self._vec_linear(self._vec_linear(mean, blk["se1"], post_act=ACT_LEAKY), blk["se2"])   # gate LOGITS; the sigmoid is in the pass below
            out = cat[:, :, col:col + W]
            ops.ecapa_rows(t2, out, gate=g, res=h)                              # x * sigmoid(.) + block input (:141, :180)
            h = out
            col += W
            if stages is not None:
                stages[f"block{bi + 1}"] = out.clone()
        m = self._tdnn(cat, self.mfa, self._f(B, T, ch[-1]), dil=c.enc_dilations[-1])
        if stages is not None:
            stages["mfa"] = m.clone()
        C = ch[-1]
        mean, std = ops.time_moments(m, eps=1e-12)
        att_in = self._f(B, T, 3 * C)
        att_in[:, :, :C].copy_(m)
        ops.broadcast_rows(mean, att_in[:, :, C:2 * C])
        ops.broadcast_rows(std, att_in[:, :, 2 * C:])
        a1 = self._tdnn(att_in, self.asp_tdnn, self._f(B, T, c.enc_attention_channels))
        a2 = self._f(B, T, c.enc_attention_channels)
        ops.ecapa_rows(a1, a2, pre_tanh=True)
        logits = self._f(B, T, C)
        ops.conv_gemm(a2, self.asp_conv, logits, precision=self.precision)
        if stages is not None:
            stages["asp_logits"] = logits.clone()
        pooled = ops.attentive_pool(m, logits, eps=1e-12)
        if stages is not None:
            stages["pooled"] = pooled.clone()
        return self._vec_linear(pooled, self.fc)
