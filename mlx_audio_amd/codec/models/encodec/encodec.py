"""EnCodec (waveform -> codes -> waveform) on MI355X: host schedule over the HIP kernels.

Mirrors ``mlx_audio/codec/models/encodec/encodec.py`` (``EncodecConfig``, ``preprocess_audio``, ``Encodec.encode`` / ``_encode_frame`` / ``decode`` /
``_decode_frame`` / ``_linear_overlap_add`` / ``chunk_length`` / ``chunk_stride``, ``quantizer.encode`` / ``decode`` /
``get_num_quantizers_for_bandwidth``), with the reference's op-by-op graph collapsed into:
  * RVQ decode (encodec.py:533-547): a frame is ONE ``embed_sum`` launch over the stacked codebooks (sum of the codebook rows in codebook order);
  * every ``nn.ELU`` is the PROLOGUE of the conv that consumes it; the resnet block's skip and shortcut conv are the residual / accumulate
    operands of its second conv; convs are implicit GEMMs (``mi355_conv_gemm``), the transposed convs (K = 2 stride) run polyphase with a
    strided store that writes only the samples the reference keeps after its trim (encodec.py:282-292);
  * the reference pads explicitly before every conv (causal: everything on the left; reflect or zero, encodec.py:213-254); reflect padding is
    a gather of the first / last rows here (the padded rows are materialised, the ELU prologue commutes with the gather);
  * ``EncodecLSTM`` (encodec.py:137-167, 296-306): the x-projection of all time steps is ONE GEMM, the recurrence runs in the native per-step
    loop of ``mi355_lstm_seq`` (csrc/lstm_seq.hip: 2 MB of Wh per layer do not fit the persistent one-CU kernel of the Kokoro LSTMs) -- ONE launch per
    step since round 5 (``ops.pack_lstm_seq_wh`` orders the rows of Wh by hidden unit, so the gates are the epilogue of the step's GEMM); the
    reference's own Metal ``lstm`` kernel lives here (its gate order i | f | g | o and its sigmoid are reproduced).
Encode side (round 5; encodec.py:340-389, 445-533, 556-650), from the same kernels: the first conv runs FLATTENED over the (reflect-) padded samples
(K taps x audio channels = the "channels" of a one-tap conv); a resnet block is conv k3 (ELU prologue) -> shortcut conv -> conv k1 accumulating onto
it; the strided ``EncodecConv1d(K = 2 r, stride r)`` is a TWO-tap conv over the padded rows regrouped ``[rows / r, r * C]`` (the padding makes the
row count a whole number of strides by construction, encodec.py:202-210); the LSTM as in the decoder; ``quantizer.encode`` is ONE
``mi355_rvq_encode`` launch (residual kept on chip across the layers in use for the bandwidth).
``norm_type = "time_group_norm"`` (the 48 kHz model's GroupNorm after every conv) is not built.  Weights: float32 checkpoints are held as fp16 MFMA images, activations split fp16 hi + lo
(``precision = 4``); deviation from the float32 oracle asserted in ``tests/test_encodec_gpu.py``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .... import ops
from ....ops import ACT_ELU, ACT_NONE, PackedConv


@dataclass
class EncodecConfig:
    """``encodec.py:21-45`` (same fields and defaults)."""
    model_type: str = "encodec"
    audio_channels: int = 1
    num_filters: int = 32
    kernel_size: int = 7
    num_residual_layers: int = 1
    dilation_growth_rate: int = 2
    codebook_size: int = 1024
    codebook_dim: int = 128
    hidden_size: int = 128
    num_lstm_layers: int = 2
    residual_kernel_size: int = 3
    use_causal_conv: bool = True
    normalize: bool = False
    pad_mode: str = "reflect"
    norm_type: str = "weight_norm"
    last_kernel_size: int = 7
    trim_right_ratio: float = 1.0
    compress: int = 2
    upsampling_ratios: List[int] = None
    target_bandwidths: List[float] = None
    sampling_rate: int = 24000
    chunk_length_s: Optional[float] = None
    overlap: Optional[float] = None
    architectures: List[str] = None


def preprocess_audio(raw_audio, sampling_rate: int = 24000, chunk_length: Optional[int] = None, chunk_stride: Optional[int] = None):
    """``encodec.py:48-86``: list of [L] / [L, C] arrays -> (inputs [N, Lmax, C], masks [N, Lmax]) padded to a whole number of chunk strides."""
    if not isinstance(raw_audio, list):
        raw_audio = [raw_audio]
    raw_audio = [torch.as_tensor(x) for x in raw_audio]
    raw_audio = [x[..., None] if x.dim() == 1 else x for x in raw_audio]
    max_length = max(a.shape[0] for a in raw_audio)
    if chunk_length is not None:
        max_length += chunk_length - (max_length % chunk_stride)
    inputs, masks = [], []
    for x in raw_audio:
        length = x.shape[0]
        mask = torch.ones(length, dtype=torch.bool)
        diff = max_length - length
        if diff > 0:
            mask = torch.nn.functional.pad(mask, (0, diff))
            x = torch.nn.functional.pad(x, (0, 0, 0, diff))
        inputs.append(x)
        masks.append(mask)
    return torch.stack(inputs), torch.stack(masks)


def decoder_layer_names(c: dict) -> dict:
    """Module indices of ``EncodecDecoder.layers`` (encodec.py:391-437: the ``nn.ELU`` entries occupy list slots too)."""
    idx = 0
    names = dict(conv_in=f"decoder.layers.{idx}", lstm=f"decoder.layers.{idx + 1}", blocks=[])
    idx += 2
    for _ in c["upsampling_ratios"]:
        idx += 1
        blk = dict(up=f"decoder.layers.{idx}", res=[])
        idx += 1
        for _ in range(c["num_residual_layers"]):
            blk["res"].append(f"decoder.layers.{idx}")
            idx += 1
        names["blocks"].append(blk)
    idx += 1
    names["conv_out"] = f"decoder.layers.{idx}"
    return names


def encoder_layer_names(c: dict) -> dict:
    """Module indices of ``EncodecEncoder.layers`` (encodec.py:343-383: the ``nn.ELU`` entries occupy list slots too)."""
    idx = 1
    names = dict(conv_in="encoder.layers.0", blocks=[])
    for _ in c["upsampling_ratios"]:
        blk = dict(res=[])
        for _ in range(c["num_residual_layers"]):
            blk["res"].append(f"encoder.layers.{idx}")
            idx += 1
        idx += 1   # nn.ELU()
        blk["down"] = f"encoder.layers.{idx}"
        idx += 1
        names["blocks"].append(blk)
    names["lstm"] = f"encoder.layers.{idx}"
    names["conv_out"] = f"encoder.layers.{idx + 2}"
    return names


def _cfg_dict(config) -> dict:
    d = dict(EncodecConfig().__dict__)
    d.update(config if isinstance(config, dict) else config.__dict__)
    if d["upsampling_ratios"] is None:
        d["upsampling_ratios"] = [8, 5, 4, 2]
    if d["target_bandwidths"] is None:
        d["target_bandwidths"] = [1.5, 3.0, 6.0, 12.0, 24.0]
    d.setdefault("use_conv_shortcut", True)
    return d


def num_quantizers(c: dict) -> int:
    frame_rate = math.ceil(c["sampling_rate"] / int(np.prod(c["upsampling_ratios"])))
    return int(1000 * c["target_bandwidths"][-1] // (frame_rate * 10))


def make_encodec_weights(config, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 decode-side parameters of the shapes ``Encodec(config)`` allocates (reference module paths, MLX layouts)."""
    c = _cfg_dict(config)
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin, gain=1.0):
        w[name + ".conv.weight"] = (torch.rand(cout, k, cin, generator=g) * 2 - 1) * math.sqrt(3.0 / (cin * k)) * gain
        w[name + ".conv.bias"] = 0.02 * torch.randn(cout, generator=g)

    names = decoder_layer_names(c)
    scaling = int(2 ** len(c["upsampling_ratios"]))
    dim = scaling * c["num_filters"]
    conv(names["conv_in"], dim, c["kernel_size"], c["hidden_size"], gain=1.5)
    for l in range(c["num_lstm_layers"]):
        p = f"{names['lstm']}.lstm.{l}."
        w[p + "Wx"] = (torch.rand(4 * dim, dim, generator=g) * 2 - 1) / math.sqrt(dim)
        w[p + "Wh"] = (torch.rand(4 * dim, dim, generator=g) * 2 - 1) / math.sqrt(dim)
        w[p + "bias"] = 0.1 * torch.randn(4 * dim, generator=g)
    for blk, ratio in zip(names["blocks"], c["upsampling_ratios"]):
        cur = scaling * c["num_filters"]
        conv(blk["up"], cur // 2, 2 * ratio, cur, gain=1.4 * math.sqrt(ratio))
        for r in blk["res"]:
            hid = (cur // 2) // c["compress"]
            conv(r + ".block.1", hid, c["residual_kernel_size"], cur // 2, gain=1.3)
            conv(r + ".block.3", cur // 2, 1, hid, gain=0.7)
            if c["use_conv_shortcut"]:
                conv(r + ".shortcut", cur // 2, 1, cur // 2, gain=0.9)
        scaling //= 2
    conv(names["conv_out"], c["audio_channels"], c["last_kernel_size"], c["num_filters"], gain=0.5)
    for i in range(num_quantizers(c)):
        w[f"quantizer.layers.{i}.codebook.embed"] = torch.randn(c["codebook_size"], c["codebook_dim"], generator=g) / math.sqrt(i + 1.0)
    return w


def make_encodec_encoder_weights(config, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 ENCODE-side parameters (``encoder.*``; reference module paths, MLX layouts): merge with ``make_encodec_weights`` for a whole model."""
    c = _cfg_dict(config)
    g = torch.Generator().manual_seed(seed + 15485863)
    w: Dict[str, torch.Tensor] = {}

    def conv(name, cout, k, cin, gain=1.0):
        w[name + ".conv.weight"] = (torch.rand(cout, k, cin, generator=g) * 2 - 1) * math.sqrt(3.0 / (cin * k)) * gain
        w[name + ".conv.bias"] = 0.02 * torch.randn(cout, generator=g)

    names = encoder_layer_names(c)
    conv(names["conv_in"], c["num_filters"], c["kernel_size"], c["audio_channels"], gain=2.0)
    scaling = 1
    for blk, ratio in zip(names["blocks"], reversed(c["upsampling_ratios"])):
        cur = scaling * c["num_filters"]
        for r in blk["res"]:
            hid = cur // c["compress"]
            conv(r + ".block.1", hid, c["residual_kernel_size"], cur, gain=1.3)
            conv(r + ".block.3", cur, 1, hid, gain=0.7)
            if c["use_conv_shortcut"]:
                conv(r + ".shortcut", cur, 1, cur, gain=0.9)
        conv(blk["down"], cur * 2, 2 * ratio, cur, gain=1.2)
        scaling *= 2
    dim = scaling * c["num_filters"]
    for l in range(c["num_lstm_layers"]):
        p = f"{names['lstm']}.lstm.{l}."
        w[p + "Wx"] = (torch.rand(4 * dim, dim, generator=g) * 2 - 1) / math.sqrt(dim)
        w[p + "Wh"] = (torch.rand(4 * dim, dim, generator=g) * 2 - 1) / math.sqrt(dim)
        w[p + "bias"] = 0.1 * torch.randn(4 * dim, generator=g)
    conv(names["conv_out"], c["hidden_size"], c["last_kernel_size"], dim, gain=1.5)
    return w


class _Quantizer:
    """``EncodecResidualVectorQuantizer`` (encodec.py:486-547)."""

    def __init__(self, w: Dict[str, torch.Tensor], c: dict, device):
        self.codebook_size = c["codebook_size"]
        self.frame_rate = math.ceil(c["sampling_rate"] / int(np.prod(c["upsampling_ratios"])))
        self.num_quantizers = num_quantizers(c)
        tabs = [w[f"quantizer.layers.{i}.codebook.embed"].float() for i in range(self.num_quantizers)]
        self.table = torch.cat(tabs, 0).contiguous().to(device)
        self.offs = torch.tensor([i * self.codebook_size for i in range(self.num_quantizers)], dtype=torch.int32, device=device)
        self.device = device
        # the layouts mi355_rvq_encode reads: [nq, bins, D], its transpose, |e|^2 / 2 (float32 codebooks as the checkpoint holds them)
        t = torch.stack(tabs, 0).contiguous()
        self.search = (t.to(device), t.transpose(1, 2).contiguous().to(device), ((t * t).sum(-1) / 2).contiguous().to(device))

    def get_num_quantizers_for_bandwidth(self, bandwidth: Optional[float] = None) -> int:
        bw_per_q = math.log2(self.codebook_size) * self.frame_rate
        n = self.num_quantizers
        if bandwidth is not None and bandwidth > 0.0:
            n = int(max(1, math.floor(bandwidth * 1000 / bw_per_q)))
        return n

    def encode(self, embeddings, bandwidth: Optional[float] = None, return_margins: bool = False, force=None):
        """encodec.py:516-533: embeddings [B, T, codebook_dim] -> codes int64 [B, nq(bandwidth), T]; per layer the nearest codeword by
        ``-(|x|^2 - 2 x e^T + |e|^2)`` (first maximum), residual -= codeword.  ``return_margins`` adds the top-2 score gap of every decision.
        ``force`` = (mask bool [B, nq, T], codes int [B, nq, T]): parity-test hook -- the layers run as one launch EACH (the same kernel, the same float32
        subtraction between them) and where the mask is set the given code replaces the search result before the residual update: re-synchronises the
        residual chain with another build's at a knife edge.  With an all-false mask the result equals the one-launch path bit for bit (tested)."""
        n = min(self.get_num_quantizers_for_bandwidth(bandwidth), self.num_quantizers)   # ``self.layers[:num_quantizers]``: a slice stops at the last layer
        x = torch.as_tensor(embeddings, dtype=torch.float32).to(self.device).contiguous()
        B, T, D = x.shape
        tables, tables_t, c2 = self.search
        if force is not None:
            fm, fc = torch.as_tensor(force[0]).to(self.device), torch.as_tensor(force[1]).to(self.device, torch.int32)
            r = x.view(B * T, D).clone()
            cs, ms = [], []
            for i in range(n):
                ci, mi = ops.rvq_encode(r, tables[i:i + 1], tables_t[i:i + 1], c2[i:i + 1], margins=True)
                ci = torch.where(fm[:, i].reshape(B * T, 1), fc[:, i].reshape(B * T, 1), ci)
                r = r - tables[i][ci.view(-1).long()]
                cs.append(ci)
                ms.append(mi)
            codes = torch.cat(cs, 1).view(B, T, n).permute(0, 2, 1).contiguous().to(torch.int64)
            return (codes, torch.cat(ms, 1).view(B, T, n).permute(0, 2, 1)) if return_margins else codes
        out = ops.rvq_encode(x.view(B * T, D), tables[:n], tables_t[:n], c2[:n], margins=return_margins)
        c, m = out if return_margins else (out, None)
        codes = c.view(B, T, n).permute(0, 2, 1).contiguous().to(torch.int64)
        return (codes, m.view(B, T, n).permute(0, 2, 1)) if return_margins else codes

    def decode(self, codes) -> torch.Tensor:
        """codes int [B, nq, T] -> [B, T, codebook_dim]."""
        codes = torch.as_tensor(codes).to(self.device)
        B, n, T = codes.shape
        if n > self.num_quantizers:
            raise IndexError(f"decode: {n} codebooks given, the model has {self.num_quantizers}")
        if int(codes.min()) < 0 or int(codes.max()) >= self.codebook_size:
            raise IndexError("decode: code out of range")
        z = torch.empty((B, T, self.table.shape[1]), dtype=torch.float32, device=self.device)
        ops.embed_sum(self.table, codes.to(torch.int32).permute(0, 2, 1), z, slot_offset=self.offs[:n])
        return z


class Encodec:
    def __init__(self, config, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", seed: int = 0):
        """``config``: ``EncodecConfig`` or a dict of its fields; ``weights``: reference parameter names (omitted: random, like a freshly constructed model)."""
        ops.require_gpu()
        self.config = config if isinstance(config, EncodecConfig) else EncodecConfig(**{k: v for k, v in dict(config).items() if k in EncodecConfig.__dataclass_fields__})
        self.c = _cfg_dict(config)
        if self.c["norm_type"] != "weight_norm":
            raise NotImplementedError("norm_type 'time_group_norm' (GroupNorm after every conv, the 48 kHz model) is not built")
        if self.c["pad_mode"] not in ("reflect", "constant", "zero"):
            raise ValueError(f"pad_mode {self.c['pad_mode']!r}")
        self.device = torch.device(device)
        if weights is None:   # a freshly constructed reference model has both halves (the reference's own test encodes with one: codec/tests/test_encodec.py)
            weights = {**make_encodec_weights(self.c, seed), **make_encodec_encoder_weights(self.c, seed)}
        self.load_weights(weights)

    # ------------------------------------------------------------------ load
    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True):
        dev, c = self.device, self.c
        w = {k: torch.as_tensor(v).detach().float().cpu() for k, v in weights.items() if k.startswith(("decoder.", "quantizer.", "encoder."))}
        names = decoder_layer_names(c)

        def conv(name) -> PackedConv:
            return ops.pack_conv(w[name + ".conv.weight"], w.get(name + ".conv.bias"), dev, f16=True)

        self.quantizer = _Quantizer(w, c, dev)
        self.conv_in = conv(names["conv_in"])
        self.lstm = []
        for l in range(c["num_lstm_layers"]):
            p = f"{names['lstm']}.lstm.{l}."
            wx = w[p + "Wx"]
            self.lstm.append(dict(wx=ops.pack_conv(wx[:, None, :], w.get(p + "bias"), dev, f16=True), wh=ops.pack_lstm_seq_wh(w[p + "Wh"], dev, f16=True),
                                  H=wx.shape[0] // 4))
        self.blocks = []
        for blk, ratio in zip(names["blocks"], c["upsampling_ratios"]):
            up_w = w[blk["up"] + ".conv.weight"]
            res = []
            for r in blk["res"]:
                res.append(dict(c1=conv(r + ".block.1"), c2=conv(r + ".block.3"), sc=conv(r + ".shortcut") if c["use_conv_shortcut"] else None))
            self.blocks.append(dict(ratio=ratio, cout=up_w.shape[0], up=ops.pack_conv_transpose(up_w, w.get(blk["up"] + ".conv.bias"), ratio, dev, f16=True), res=res))
        self.conv_out = conv(names["conv_out"])
        self.enc = None
        if "encoder.layers.0.conv.weight" in w:   # the encode half (encodec.py:340-389)
            en = encoder_layer_names(c)
            w0 = w[en["conv_in"] + ".conv.weight"]   # [F, K, channels] -> one tap of K * channels "channels" (flattened conv over the padded samples)
            stem = ops.pack_conv(w0.reshape(w0.shape[0], 1, w0.shape[1] * w0.shape[2]).contiguous(), w.get(en["conv_in"] + ".conv.bias"), dev, f16=True)
            eblocks = []
            for blk, ratio in zip(en["blocks"], reversed(c["upsampling_ratios"])):
                res = []
                for j, r in enumerate(blk["res"]):
                    res.append(dict(c1=conv(r + ".block.1"), c2=conv(r + ".block.3"), sc=conv(r + ".shortcut") if c["use_conv_shortcut"] else None,
                                    dil=c["dilation_growth_rate"] ** j))
                wd = w[blk["down"] + ".conv.weight"]   # [cout, 2 r, cin]: tap j r + q -> (tap j, channel q cin + ch)
                cout, k, cin = wd.shape
                if k != 2 * ratio:
                    raise ValueError(f"{blk['down']}: {k} taps for stride {ratio} (the reference builds kernel_size = 2 * ratio)")
                eblocks.append(dict(ratio=ratio, cin=cin, res=res, down=ops.pack_conv(wd.reshape(cout, 2, ratio * cin).contiguous(), w.get(blk["down"] + ".conv.bias"), dev, f16=True)))
            elstm = []
            for l in range(c["num_lstm_layers"]):
                p = f"{en['lstm']}.lstm.{l}."
                wx = w[p + "Wx"]
                elstm.append(dict(wx=ops.pack_conv(wx[:, None, :], w.get(p + "bias"), dev, f16=True), wh=ops.pack_lstm_seq_wh(w[p + "Wh"], dev, f16=True), H=wx.shape[0] // 4))
            self.enc = dict(stem=stem, k0=w0.shape[1], blocks=eblocks, lstm=elstm, out=conv(en["conv_out"]))
        return self

    @classmethod
    def from_pretrained(cls, path_or_repo: str, device="cuda:0"):
        """encodec.py:710-737 for a LOCAL directory (``config.json`` + ``model.safetensors``; no hub here): (model, preprocess_audio partial)."""
        import functools
        import json
        from pathlib import Path

        from safetensors.torch import load_file

        path = Path(path_or_repo)
        if not path.exists():
            raise FileNotFoundError(f"{path_or_repo}: Encodec.from_pretrained needs a local directory (no hub access in this build)")
        with open(path / "config.json", "r") as f:
            config = json.load(f)
        config = {k: v for k, v in config.items() if k in EncodecConfig.__dataclass_fields__}
        model = cls(config, weights=load_file(str(path / "model.safetensors")), device=device)
        processor = functools.partial(preprocess_audio, sampling_rate=model.sampling_rate, chunk_length=model.chunk_length, chunk_stride=model.chunk_stride)
        return model, processor

    # ------------------------------------------------------------------ reference surface
    @property
    def channels(self):
        return self.c["audio_channels"]

    @property
    def sampling_rate(self):
        return self.c["sampling_rate"]

    @property
    def chunk_length(self):
        return None if self.c["chunk_length_s"] is None else int(self.c["chunk_length_s"] * self.c["sampling_rate"])

    @property
    def chunk_stride(self):
        if self.c["chunk_length_s"] is None or self.c["overlap"] is None:
            return None
        return max(1, int((1.0 - self.c["overlap"]) * self.chunk_length))

    # ------------------------------------------------------------------ encoder
    def _encoder(self, x: torch.Tensor, return_stages: bool = False):
        """x [B, L, audio_channels] -> embeddings [B, T, hidden_size] (encodec.py:340-389)."""
        if self.enc is None:
            raise ValueError("this Encodec was loaded without encoder weights (decode-only checkpoint)")
        c, e = self.c, self.enc
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()
        B, L, ch = x.shape
        if ch != c["audio_channels"]:
            raise ValueError(f"encoder: {ch} audio channels given, the model has {c['audio_channels']}")
        st = {}
        xp = self._padded(x, e["k0"], 1)                                  # [B, L + K - 1, ch]: the taps of a row are contiguous samples
        h = self._f(B, L, e["stem"].cout)
        ops.conv_gemm(xp, e["stem"], h, lout=L, flat=dict(ldx=ch, x_off=0, channels=ch), precision=4)
        st["conv_in"] = h
        for bi, blk in enumerate(e["blocks"]):
            r, C = blk["ratio"], blk["cin"]
            for rb in blk["res"]:
                hid = self._f(B, L, rb["c1"].cout)
                self._conv(h, rb["c1"], hid, dilation=rb["dil"], elu=True)
                out = self._f(B, L, C)
                if rb["sc"] is not None:
                    self._conv(h, rb["sc"], out)
                    self._conv(hid, rb["c2"], out, elu=True, accumulate=True)
                else:
                    self._conv(hid, rb["c2"], out, elu=True, res=h)
                h = out
            hp = self._padded(h, 2 * r, 1, stride=r)                        # a whole number of strides by construction
            assert hp.shape[1] % r == 0, (hp.shape, r)
            rows = hp.shape[1] // r
            L = rows - 1
            y = self._f(B, L, blk["down"].cout)
            ops.conv_gemm(hp.view(B, rows, r * C), blk["down"], y, pad=0, lout=L, pre_act=ACT_ELU, precision=4)   # K = 2 r, stride r as 2 taps of r rows
            h = y
            st[f"block{bi}"] = h
        y = h
        for l in e["lstm"]:
            xq = self._f(B, L, 4 * l["H"])
            ops.conv_gemm(y, l["wx"], xq, precision=4)
            out = self._f(B, L, l["H"])
            ops.lstm_seq(xq, l["wh"], out)
            y = out
        h = y + h
        st["lstm"] = h
        z = self._f(B, L, c["hidden_size"])
        self._conv(h, e["out"], z, elu=True)
        st["embeddings"] = z
        return (z, st) if return_stages else z

    def _encode_frame(self, input_values: torch.Tensor, bandwidth: float, padding_mask: torch.Tensor):
        """encodec.py:556-583: one chunk -> (codes [B, nq, T], scale [B, 1, 1] or None)."""
        c = self.c
        length = input_values.shape[1]
        duration = length / c["sampling_rate"]
        if c["chunk_length_s"] is not None and duration > 1e-5 + c["chunk_length_s"]:
            raise RuntimeError(f"Duration of frame ({duration}) is longer than chunk {c['chunk_length_s']}")
        scale = None
        if c["normalize"]:
            input_values = input_values * padding_mask[..., None].to(input_values.dtype)
            mono = input_values.sum(dim=2, keepdim=True) / input_values.shape[2]
            scale = torch.sqrt((mono * mono).mean(dim=1, keepdim=True)) + 1e-8
            input_values = input_values / scale
        return self.quantizer.encode(self._encoder(input_values), bandwidth), scale

    def encode(self, input_values, padding_mask=None, bandwidth: Optional[float] = None):
        """encodec.py:585-650: input_values [B, samples, channels] -> (codes int64 [n_chunks, B, nq, T], scales: list of [B, 1, 1] or None)."""
        c = self.c
        if bandwidth is None:
            bandwidth = c["target_bandwidths"][0]
        if bandwidth not in c["target_bandwidths"]:
            raise ValueError(f"This model doesn't support the bandwidth {bandwidth}. Select one of {c['target_bandwidths']}.")
        input_values = torch.as_tensor(input_values, dtype=torch.float32).to(self.device)
        _, input_length, channels = input_values.shape
        if channels < 1 or channels > 2:
            raise ValueError(f"Number of audio channels must be 1 or 2, but got {channels}")
        chunk_length = self.chunk_length
        if chunk_length is None:
            chunk_length = input_length
            stride = input_length
        else:
            stride = self.chunk_stride
        if padding_mask is None:
            padding_mask = torch.ones(input_values.shape[:2], dtype=torch.bool, device=self.device)
        padding_mask = torch.as_tensor(padding_mask).to(self.device)
        step = chunk_length - stride
        if (input_length % stride) != step:
            raise ValueError("The input length is not properly padded for batched chunked encoding. Make sure to pad the input correctly.")
        frames, scales = [], []
        for offset in range(0, input_length - step, stride):
            mask = padding_mask[:, offset:offset + chunk_length].bool()
            codes, scale = self._encode_frame(input_values[:, offset:offset + chunk_length], bandwidth, mask)
            frames.append(codes)
            scales.append(scale)
        return torch.stack(frames), scales

    # ------------------------------------------------------------------ decoder
    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _padded(self, x: torch.Tensor, kernel_size: int, dilation: int, stride: int = 1) -> torch.Tensor:
        """The explicit padding of ``EncodecConv1d`` (encodec.py:199-247): x [B, L, C] -> [B, L + (kernel_size - stride) + extra, C]."""
        c = self.c
        k_eff = (kernel_size - 1) * dilation + 1
        padding_total = kernel_size - stride
        L = x.shape[1]
        n_frames = int(math.ceil((L - k_eff + padding_total) / stride + 1)) - 1
        extra = n_frames * stride + k_eff - padding_total - L
        if c["use_causal_conv"]:
            pl, pr = padding_total, extra
        else:
            pr = padding_total // 2
            pl = padding_total - pr
            pr += extra
        if pl == 0 and pr == 0:
            return x
        if c["pad_mode"] == "reflect":
            parts = ([x[:, 1:pl + 1].flip(1)] if pl else []) + [x] + ([x[:, max(L - (pr + 1), 0):-1].flip(1)] if pr else [])
            return torch.cat(parts, dim=1).contiguous()
        return torch.nn.functional.pad(x, (0, 0, pl, pr)).contiguous()

    def _conv(self, x, pc: PackedConv, y, *, dilation: int = 1, elu: bool = False, res=None, accumulate: bool = False):
        xp = self._padded(x, pc.k, dilation)
        assert xp.shape[1] - (pc.k - 1) * dilation == y.shape[1], (xp.shape, y.shape, pc.k, dilation)
        return ops.conv_gemm(xp, pc, y, dil=dilation, pad=0, lout=y.shape[1], pre_act=ACT_ELU if elu else ACT_NONE, res=res, accumulate=accumulate, precision=4)

    def _decoder(self, z: torch.Tensor, return_stages: bool = False):
        """z [B, T, hidden_size] -> [B, T * prod(ratios), audio_channels] (encodec.py:391-444)."""
        c = self.c
        B, T, _ = z.shape
        st = {}
        dim = self.conv_in.cout
        h = self._f(B, T, dim)
        self._conv(z.contiguous(), self.conv_in, h)
        st["conv_in"] = h
        y = h
        for l in self.lstm:
            xp = self._f(B, T, 4 * l["H"])
            ops.conv_gemm(y, l["wx"], xp, precision=4)          # x @ Wx^T + bias for every step: one GEMM
            out = self._f(B, T, l["H"])
            ops.lstm_seq(xp, l["wh"], out)
            y = out
        h = y + h                                                # EncodecLSTM's skip connection (encodec.py:306)
        st["lstm"] = h
        for bi, blk in enumerate(self.blocks):
            s, cout, taps = blk["ratio"], blk["cout"], blk["up"].k
            Lin = h.shape[1]
            # full transposed conv has (Lin + 1) * s samples; the causal trim keeps the first Lin * s (trim_right_ratio = 1), else the middle
            padding_total = s
            pr = math.ceil(padding_total * c["trim_right_ratio"]) if c["use_causal_conv"] else padding_total // 2
            pl = padding_total - pr
            Lout = (Lin + 1) * s - padding_total
            y = self._f(B, Lout, cout)
            ops.conv_gemm(h, blk["up"], y, pad=taps - 1, lout=Lin + taps - 1, up=dict(s=s, p=pl, cout=cout, lout=Lout), pre_act=ACT_ELU, precision=4)
            for r in blk["res"]:
                hid = self._f(B, Lout, r["c1"].cout)
                self._conv(y, r["c1"], hid, dilation=1, elu=True)   # dilation_growth_rate ** j with j < num_residual_layers; j = 0 -> 1 (see load)
                out = self._f(B, Lout, cout)
                if r["sc"] is not None:
                    self._conv(y, r["sc"], out)                     # shortcut(residual) ...
                    self._conv(hid, r["c2"], out, elu=True, accumulate=True)   # ... + block(hidden_states)
                else:
                    self._conv(hid, r["c2"], out, elu=True, res=y)
                y = out
            h = y
            st[f"block{bi}"] = h
        out = self._f(B, h.shape[1], c["audio_channels"])
        self._conv(h, self.conv_out, out, elu=True)
        return (out, st) if return_stages else out

    def _decode_frame(self, codes, scale=None) -> torch.Tensor:
        out = self._decoder(self.quantizer.decode(codes))
        return out * torch.as_tensor(scale).to(out.device) if scale is not None else out

    @staticmethod
    def _linear_overlap_add(frames: List[torch.Tensor], stride: int) -> torch.Tensor:
        """``encodec.py:652-677``."""
        if len(frames) == 0:
            raise ValueError("`frames` cannot be an empty list.")
        N, fl, C = frames[0].shape
        dev = frames[0].device
        total = stride * (len(frames) - 1) + frames[-1].shape[1]
        tv = torch.linspace(0, 1, fl + 2, dtype=torch.float32, device=dev)[1:-1]
        weight = (0.5 - (tv - 0.5).abs())[:, None]
        sw = torch.zeros((total, 1), dtype=torch.float32, device=dev)
        out = torch.zeros((N, total, C), dtype=torch.float32, device=dev)
        off = 0
        for fr in frames:
            n = fr.shape[1]
            out[:, off:off + n] += weight[:n] * fr
            sw[off:off + n] += weight[:n]
            off += stride
        return out / sw

    def decode(self, audio_codes, audio_scales, padding_mask=None) -> torch.Tensor:
        """``encodec.py:740-777``: ``audio_codes`` int [n_chunks, B, nq, T] with chunking, [B, 1, nq, T] without; -> [B, samples, channels]."""
        audio_codes = torch.as_tensor(audio_codes)
        if self.chunk_length is None:
            if audio_codes.shape[1] != 1:
                raise ValueError(f"Expected one frame, got {len(audio_codes)}")
            audio = self._decode_frame(audio_codes[:, 0], audio_scales[0])
        else:
            audio = self._linear_overlap_add([self._decode_frame(f, s) for f, s in zip(audio_codes, audio_scales)], self.chunk_stride or 1)
        if padding_mask is not None and padding_mask.shape[1] < audio.shape[1]:
            audio = audio[:, :padding_mask.shape[1]]
        return audio
