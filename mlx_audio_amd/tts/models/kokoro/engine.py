"""Kokoro-82M forward pass on MI355X: the host-side schedule over the HIP kernels.

Mirrors ``Model.__call__`` of the reference (``tts/models/kokoro/kokoro.py:111-177``) and the modules
it calls (``modules.py``, ``istftnet.py``), but batched over utterances (ragged, padded to the longest)
and with the reference's op-by-op graph collapsed into fused kernels:

  * weight norm is folded once at load time (the reference recomputes it every call, istftnet.py:130);
  * all AdaIN / AdaLayerNorm style projections of the whole network are ONE GEMM per style vector;
  * AdaIN + Snake / LeakyReLU are the prologue of the consuming conv, bias / residual / scaling /
    resblock averaging its epilogue; channel concatenations are column offsets of wider buffers;
  * ConvTranspose1d up-samplers run as polyphase stride-1 GEMMs; the alignment "one-hot matmul" is a gather.

Everything here is plumbing (allocation, pointer arithmetic, launch order); all arithmetic on activations
happens in libmi355audio.so.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .... import ops
from ....ops import ACT_GELU, ACT_GELU_TANH, ACT_LEAKY, ACT_NONE, ACT_SNAKE, PackedConv, round_up


def fold_weight_norm(v: torch.Tensor, g: torch.Tensor, param_dtype: torch.dtype) -> torch.Tensor:
    """w = g * v / (||v|| + 1e-7) over all axes but 0, evaluated in the parameter dtype exactly like the
    reference's ``weight_norm`` (istftnet.py:53-93) does on an MLX array of that dtype; returns float32."""
    v_ = v.to(param_dtype)
    g_ = g.to(param_dtype)
    axes = tuple(range(1, v_.dim()))
    nrm = torch.sqrt((v_ * v_).sum(dim=axes, keepdim=True))
    w = (v_ / (nrm + torch.tensor(1e-7, dtype=param_dtype))) * g_
    return w.to(torch.float32)


@dataclass
class _AdaIN:
    off: int  # column offset of [gamma | beta] in the style-projection output
    c: int
    q: bool = False  # KittenTTS: fc sees the fake-quantised style vector (kitten_tts/istftnet.py:336, modules.py:80)


@dataclass
class _StyleProj:
    """Every style projection of one style vector, from the plain and (KittenTTS, when any AdaIN is flagged) the fake-quantised vector."""
    plain: torch.Tensor
    quant: Optional[torch.Tensor] = None


@dataclass
class _ResBlk1d:  # AdainResBlk1d
    conv1: PackedConv
    conv2: PackedConv
    norm1: _AdaIN
    norm2: _AdaIN
    conv1x1: Optional[PackedConv]
    pool_w: Optional[torch.Tensor]
    pool_b: Optional[torch.Tensor]
    din: int
    dout: int
    name: str = ""


@dataclass
class _ResBlock1:  # AdaINResBlock1
    convs1: List[PackedConv]
    convs2: List[PackedConv]
    adain1: List[_AdaIN]
    adain2: List[_AdaIN]
    alpha1: List[torch.Tensor]
    alpha2: List[torch.Tensor]
    k: int
    dils: Sequence[int]
    ch: int
    name: str = ""


@dataclass
class _LSTM:
    wx: PackedConv
    wh: torch.Tensor
    hid: int
    q: bool = False  # KittenTTS: fake-quantised input and per-step hidden vector (kitten_tts/modules.py:155,178)
    f16: bool = False  # recurrent weights held as IEEE half (precision 4)
    wh_f16: bool = False   # the image handed to the kernel is IEEE half: f16 checkpoints, or a bf16 checkpoint scaled exactly into half's normal range
    wh_scale: float = 0.0  # ... then: the power of two the recurrent sum is multiplied by


class _StyleBank:
    """Collects every ``fc(style)`` of one style vector into a single [sum(2C), style_dim] linear."""

    def __init__(self):
        self.ws: List[torch.Tensor] = []
        self.bs: List[torch.Tensor] = []
        self.n = 0
        self.any_q = False  # some AdaIN of this bank wants the fake-quantised style vector (KittenTTS)

    def add(self, w: torch.Tensor, b: torch.Tensor) -> int:
        off = self.n
        self.ws.append(w.float())
        self.bs.append(b.float())
        self.n += round_up(w.shape[0], 4)
        pad = self.n - off - w.shape[0]
        if pad:
            self.ws.append(torch.zeros(pad, w.shape[1]))
            self.bs.append(torch.zeros(pad))
        return off

    def pack(self, device, f16: bool = False) -> PackedConv:
        return ops.pack_conv(torch.cat(self.ws, 0), torch.cat(self.bs, 0), device, f16=f16)


@dataclass
class KokoroFront:
    """What crosses the front / back split of the forward pass, per utterance (lists of length B): token ids ``[T]``, style row ``[256]``,
    duration-encoder output ``d`` ``[T, hid + style]``, predicted durations ``[T]`` int32, and the frame counts (host ints)."""
    ids: List[torch.Tensor]
    ref_s: torch.Tensor
    d: List[torch.Tensor]
    dur: List[torch.Tensor]
    frames: List[int]
    speed: float = 1.0
    trace: Optional[dict] = None
    padded: Optional[dict] = None  # front()'s own padded device tensors: back() on the same, unmoved state skips the re-padding

    def select(self, which: Sequence[int]) -> "KokoroFront":
        which = list(which)
        return KokoroFront([self.ids[i] for i in which], self.ref_s[which], [self.d[i] for i in which], [self.dur[i] for i in which],
                           [self.frames[i] for i in which], self.speed, None)

    def pack(self, i: int) -> torch.Tensor:
        """Utterance ``i`` as one float32 vector: ``[T, ids (bit pattern), dur (bit pattern), ref_s, d]`` -- the wire format of a re-balance."""
        T = int(self.ids[i].numel())
        hdr = torch.tensor([T], dtype=torch.int32, device=self.d[i].device).view(torch.float32)
        return torch.cat([hdr, self.ids[i].to(torch.int32).view(torch.float32), self.dur[i].to(torch.int32).view(torch.float32),
                          self.ref_s[i].reshape(-1).to(torch.float32), self.d[i].reshape(-1).to(torch.float32)])

    @staticmethod
    def packed_size(T: int, style: int, width: int) -> int:
        return 1 + 2 * T + style + T * width

    @classmethod
    def unpack(cls, blobs: Sequence[torch.Tensor], frames: Sequence[int], style: int, width: int, speed: float = 1.0) -> "KokoroFront":
        ids, dur, ref, d = [], [], [], []
        for b in blobs:
            T = (int(b.numel()) - 1 - style) // (2 + width)  # from the blob length: no device read
            ids.append(b[1:1 + T].view(torch.int32))
            dur.append(b[1 + T:1 + 2 * T].view(torch.int32))
            ref.append(b[1 + 2 * T:1 + 2 * T + style])
            d.append(b[1 + 2 * T + style:].reshape(T, width))
        return cls(ids, torch.stack(ref, 0), d, dur, list(frames), speed, None)

    @classmethod
    def concat(cls, parts: Sequence["KokoroFront"]) -> "KokoroFront":
        parts = [p for p in parts if len(p.ids)]
        return cls(sum((p.ids for p in parts), []), torch.cat([p.ref_s for p in parts], 0), sum((p.d for p in parts), []),
                   sum((p.dur for p in parts), []), sum((list(p.frames) for p in parts), []), parts[0].speed, None)


class KokoroEngine:
    # what KittenTTS (tts/models/kitten_tts/engine.py) changes: decoder widths, ALBERT activation, Snake parameter names, the duration clip
    # and the list of modules whose inputs are fake-quantised
    ffn_act = ACT_GELU        # nn.GELU (modules.py:571)
    alpha_name = "alpha{w}.{i}"
    max_frames = 0            # clip(round(dur), 1, 100) (kokoro.py:145-147)
    coarse_f32 = False        # SineGen's coarse phase grid: ceil(L * (1 / up)) in doubles (istftnet.py:567,590-594)

    def _decoder_dims(self, config: dict):
        """(width of the decoder blocks, generator input width, asr_res width): fixed in Kokoro (istftnet.py:948-975)."""
        return 1024, 512, 64

    @staticmethod
    def default_precision(param_dtype, quant_modules: Sequence[str] = ()) -> int:
        """The mode an engine runs when the caller names none -- the SAME rule for ``load_model()``, ``smoke()``, the tests and ``bench.py``:
        bf16 checkpoints (Kokoro-82M-bf16, BASELINE config[1]) -> 6 (round 6: fp16 hi pass + block-scaled FP4 lo pass on the >= 7-tap vocoder convs,
        exact bf16 images elsewhere: 2.3e-4 of the peak / 76 dB on the canonical sentence against the 2e-3 / 50 dB bars -- a 9x margin -- and 6.6 % faster
        than mode 5 on the contract line, same box: profiles/r6_bench_contract_p6_call6.json; mode 5 = the e4m3 lo pass of rounds 4 - 5, 7.9e-5 / 89 dB,
        stays selectable, as does the exact bf16 hi + lo mode 2); float32 checkpoints -> 4 (fp16 images, fp16 hi + lo); fp16 checkpoints and engines with
        fake-quantised modules (KittenTTS: the quantising prologue has no MX form) -> 2."""
        if param_dtype == torch.float32:
            return 4
        return 6 if (param_dtype == torch.bfloat16 and not tuple(quant_modules)) else 2

    def __init__(self, weights: Dict[str, torch.Tensor], config: dict, device="cuda", param_dtype=torch.bfloat16,
                 precision: Optional[int] = None, quant_modules: Sequence[str] = ()):
        ops.require_gpu()
        self.cfg = config
        self.qmods = tuple(quant_modules)
        if precision is None:
            precision = self.default_precision(param_dtype, self.qmods)
        # precision 4: EVERY conv / linear weight as an fp16 image (11 significant bits instead of bf16's 8) with fp16 hi + lo activations -- the mode
        # for float32 checkpoints, whose values a bf16 image would round at 2^-9 (measured against the reference run on a float32 checkpoint:
        # 35-40 dB with bf16 images).  bf16 checkpoints (Kokoro-82M-bf16, BASELINE config[1]) are exact in the default mode 2.  The recurrent LSTM
        # weights follow: IEEE half in mode 4 (``mi355_lstm_args.wh_f16``), bf16 otherwise.
        self.all_f16 = precision == 4
        self.cdim, self.gdim, self.adim = self._decoder_dims(config)
        self.dev = torch.device(device)
        self.pdt = param_dtype
        self.precision = precision
        self.fuse_stats = True  # instance-norm statistics out of the producing conv's epilogue (False: separate pass)
        # KittenTTS activation quantisation: extrema pass + quantiser inside the consuming conv's prologue (False: materialise the quantised tensor
        # with mi355_fake_quant_u8 first -- the round-2 path, kept for A/B runs: MI355_FOLD_QUANT=0)
        self.fold_quant = os.environ.get("MI355_FOLD_QUANT", "1") != "0"
        # ... and the extrema of a quantised conv's input from the per-block partials the producing conv left, instead of one more read of the tensor (False: every
        # quantised conv sweeps its input; A/B knob MI355_EXT_PARTIALS=0)
        self.ext_partials = os.environ.get("MI355_EXT_PARTIALS", "1") != "0"
        self.w = weights
        ist = config["istftnet"]
        self.rates = [int(r) for r in ist["upsample_rates"]]
        self.kers = [int(k) for k in ist["upsample_kernel_sizes"]]
        self.n_fft, self.hop = int(ist["gen_istft_n_fft"]), int(ist["gen_istft_hop_size"])
        self.total_up = int(np.prod(self.rates)) * self.hop
        self.rk = list(ist["resblock_kernel_sizes"])
        self.rd = [list(d) for d in ist["resblock_dilation_sizes"]]
        self.hid = config["hidden_dim"]
        self.sty = config["style_dim"]
        self.n_layer = config["n_layer"]
        self.pb = config["plbert"]
        self.bank_dec = _StyleBank()
        self.bank_pred = _StyleBank()
        self._load()

    # ------------------------------------------------------------------ weight preparation (load time, CPU)
    def _t(self, name) -> torch.Tensor:
        return self.w[name].to(torch.float32)

    def _q(self, t: torch.Tensor) -> torch.Tensor:
        """Parameters live in the checkpoint dtype; compute sees exactly those values."""
        return t.to(self.pdt).to(torch.float32)

    def _lin(self, pre, bias=True) -> PackedConv:
        return ops.pack_conv(self._q(self._t(f"{pre}.weight")), self._q(self._t(f"{pre}.bias")) if bias else None, self.dev, f16=self.all_f16)

    def _wn(self, pre) -> torch.Tensor:
        return fold_weight_norm(self.w[f"{pre}.weight_v"], self.w[f"{pre}.weight_g"], self.pdt)

    def _convw(self, pre, bias=True) -> PackedConv:
        b = self._q(self._t(f"{pre}.bias")) if bias and f"{pre}.bias" in self.w else None
        w = self._wn(pre)
        # precision 5: MX images where the fp16 hi + e4m3 lo arithmetic is the faster one (>= 7 taps on the wave-specialised kernel); every other
        # conv keeps the default mode's bf16 image (exact weights, the cheapest prologue: the 3-tap convs are HBM / VALU bound)
        # (a conv whose input is fake-quantised -- KittenTTS -- keeps its bf16 image: the quantising prologue has no MX form, mi355_conv_gemm rejects it)
        # precision 6: the same convs on MX4 images (FP4 lo elements: the lo pass at 4x the 16-bit rate of the matrix pipe)
        mx = (self.precision in (5, 6) and pre.startswith("decoder.") and w.dim() == 3 and ops.mx_pays(w.shape[0], w.shape[1], w.shape[2])
              and not (self.qmods and self._isq(pre)))
        return ops.pack_conv(w, b, self.dev, f16=self._f16(pre), mx=(2 if self.precision == 6 else 1) if mx else 0)

    def _f16(self, pre: str) -> bool:
        """precision 3: the decoder / generator convs (97 % of the FLOPs) run the single-pass fp16 MFMA; the front end
        (PL-BERT, prosody predictor, text encoder: the bit-exact duration path and the F0 curve the harmonic source
        integrates) stays on the bf16 hi+lo split.  precision 5: the decoder / generator convs of >= 7 taps on the fp16 hi + block-scaled e4m3 lo
        pass (MX images, ``_convw``), everything else exactly as the default mode."""
        return self.all_f16 or (self.precision == 3 and pre.startswith("decoder."))

    def _dvec(self, t: torch.Tensor, pad_to: int = 0) -> torch.Tensor:
        t = self._q(t.reshape(-1).float())
        if pad_to and t.numel() < pad_to:
            t = torch.cat([t, torch.ones(pad_to - t.numel())])
        return t.contiguous().to(self.dev)

    def _isq(self, name: str) -> bool:
        """Does module ``name`` carry ``activation_quant``?  A module is flagged when a listed name is the module itself or lies below it
        (kitten_tts/kitten_tts.py:291-299); Kokoro lists none."""
        return any(q == name or q.startswith(name + ".") for q in self.qmods)

    def _adain(self, bank: _StyleBank, pre, c) -> _AdaIN:
        q = self._isq(pre)
        bank.any_q = bank.any_q or q
        return _AdaIN(bank.add(self._q(self._t(f"{pre}.fc.weight")), self._q(self._t(f"{pre}.fc.bias"))), c, q)

    def _lstm(self, pre) -> _LSTM:
        wx = torch.cat([self._q(self._t(f"{pre}.Wx_forward")), self._q(self._t(f"{pre}.Wx_backward"))], 0)
        b = torch.cat([self._q(self._t(f"{pre}.bias_ih_forward")) + self._q(self._t(f"{pre}.bias_hh_forward")),
                       self._q(self._t(f"{pre}.bias_ih_backward")) + self._q(self._t(f"{pre}.bias_hh_backward"))])
        whf, whb = self._q(self._t(f"{pre}.Wh_forward")), self._q(self._t(f"{pre}.Wh_backward"))
        # bf16 checkpoints: the recurrent weights as an exactly scaled IEEE-half image (v_fma_mix_f32 reads a half operand directly: half the VALU
        # work of the bf16 image's shift / mask + FMA, bit-identical sums; MI355_LSTM_BF16=1 keeps the bf16 image: A/B aid)
        scaled = None if (self.all_f16 or os.environ.get("MI355_LSTM_BF16")) else ops.pack_lstm_wh_scaled(whf, whb, self.dev)
        if scaled is not None:
            return _LSTM(ops.pack_conv(wx, b, self.dev, f16=self.all_f16), scaled[0], whf.shape[1], self._isq(pre), self.all_f16, True, scaled[1])
        return _LSTM(ops.pack_conv(wx, b, self.dev, f16=self.all_f16), ops.pack_lstm_wh(whf, whb, self.dev, f16=self.all_f16), whf.shape[1],
                     self._isq(pre), self.all_f16, self.all_f16, 0.0)

    def _resblk1d(self, bank, pre, din, dout) -> _ResBlk1d:
        up = f"{pre}.pool.weight_v" in self.w
        pool_w = pool_b = None
        if up:
            pool_w = self._wn(f"{pre}.pool")[:, :, 0].contiguous().to(self.dev)  # [C, 3]
            pool_b = self._q(self._t(f"{pre}.pool.bias")).contiguous().to(self.dev)
        return _ResBlk1d(self._convw(f"{pre}.conv1"), self._convw(f"{pre}.conv2"), self._adain(bank, f"{pre}.norm1", din),
                         self._adain(bank, f"{pre}.norm2", dout),
                         self._convw(f"{pre}.conv1x1", bias=False) if f"{pre}.conv1x1.weight_v" in self.w else None,
                         pool_w, pool_b, din, dout, pre)

    def _resblock1(self, bank, pre, ch, k, dils) -> _ResBlock1:
        cp = round_up(ch, 32)
        return _ResBlock1([self._convw(f"{pre}.convs1.{i}") for i in range(3)], [self._convw(f"{pre}.convs2.{i}") for i in range(3)],
                          [self._adain(bank, f"{pre}.adain1.{i}", ch) for i in range(3)],
                          [self._adain(bank, f"{pre}.adain2.{i}", ch) for i in range(3)],
                          [self._dvec(self._t(f"{pre}.{self.alpha_name.format(w=1, i=i)}"), cp) for i in range(3)],
                          [self._dvec(self._t(f"{pre}.{self.alpha_name.format(w=2, i=i)}"), cp) for i in range(3)], k, dils, ch, pre)

    def _load(self):
        d, hid, sty = self.dev, self.hid, self.sty
        # ---- PL-BERT
        e = "bert.embeddings"
        self.word_emb = self._q(self._t(f"{e}.word_embeddings.weight")).contiguous().to(d)
        self.pos_emb = self._q(self._t(f"{e}.position_embeddings.weight")).contiguous().to(d)
        self.type_row = self._q(self._t(f"{e}.token_type_embeddings.weight"))[0].contiguous().to(d)
        self.emb_ln = (self._dvec(self._t(f"{e}.LayerNorm.weight")), self._dvec(self._t(f"{e}.LayerNorm.bias")))
        self.map_in = self._lin("bert.encoder.embedding_hidden_mapping_in")
        lay = "bert.encoder.albert_layer_groups.0.albert_layers.0"
        qkv_w = torch.cat([self._q(self._t(f"{lay}.attention.{n}.weight")) for n in ("query", "key", "value")], 0)
        qkv_b = torch.cat([self._q(self._t(f"{lay}.attention.{n}.bias")) for n in ("query", "key", "value")], 0)
        self.qkv = ops.pack_conv(qkv_w, qkv_b, d, f16=self.all_f16)
        self.att_dense = self._lin(f"{lay}.attention.dense")
        self.att_ln = (self._dvec(self._t(f"{lay}.attention.LayerNorm.weight")), self._dvec(self._t(f"{lay}.attention.LayerNorm.bias")))
        self.ffn = self._lin(f"{lay}.ffn")
        self.ffn_out = self._lin(f"{lay}.ffn_output")
        self.full_ln = (self._dvec(self._t(f"{lay}.full_layer_layer_norm.weight")), self._dvec(self._t(f"{lay}.full_layer_layer_norm.bias")))
        self.bert_encoder = self._lin("bert_encoder")
        # ---- prosody predictor
        self.dur_lstms = [self._lstm(f"predictor.text_encoder.lstms.{2 * i}") for i in range(self.n_layer)]
        self.dur_adaln = [self._adain(self.bank_pred, f"predictor.text_encoder.lstms.{2 * i + 1}", hid) for i in range(self.n_layer)]
        self.pred_lstm = self._lstm("predictor.lstm")
        self.dur_proj = self._lin("predictor.duration_proj.linear_layer")
        self.shared = self._lstm("predictor.shared")
        self.f0_blocks, self.n_blocks = [], []
        dims = [(hid, hid), (hid, hid // 2), (hid // 2, hid // 2)]
        for i, (a, b) in enumerate(dims):
            self.f0_blocks.append(self._resblk1d(self.bank_pred, f"predictor.F0.{i}", a, b))
            self.n_blocks.append(self._resblk1d(self.bank_pred, f"predictor.N.{i}", a, b))
        self.f0_proj = ops.pack_conv(self._q(self._t("predictor.F0_proj.weight")), self._q(self._t("predictor.F0_proj.bias")), d, f16=self.all_f16)
        self.n_proj = ops.pack_conv(self._q(self._t("predictor.N_proj.weight")), self._q(self._t("predictor.N_proj.bias")), d, f16=self.all_f16)
        # ---- text encoder
        self.te_emb = self._q(self._t("text_encoder.embedding.weight")).contiguous().to(d)
        self.te_cnn = []
        for i in range(self.n_layer):
            self.te_cnn.append((self._convw(f"text_encoder.cnn.{i}.0"), self._dvec(self._t(f"text_encoder.cnn.{i}.1.weight")),
                                self._dvec(self._t(f"text_encoder.cnn.{i}.1.bias"))))
        self.te_k = self.w["text_encoder.cnn.0.0.weight_v"].shape[1]
        self.te_lstm = self._lstm("text_encoder.lstm")
        # ---- decoder
        cd, gd, ad = self.cdim, self.gdim, self.adim
        self.enc_blk = self._resblk1d(self.bank_dec, "decoder.encode", hid + 2, cd)
        self.dec_blks = [self._resblk1d(self.bank_dec, f"decoder.decode.{i}", cd + 2 + ad, cd if i < 3 else gd) for i in range(4)]
        f0w, nw = self._wn("decoder.F0_conv").reshape(-1), self._wn("decoder.N_conv").reshape(-1)
        self.f0_conv = ([float(v) for v in f0w], float(self._q(self._t("decoder.F0_conv.bias"))[0]))
        self.n_conv = ([float(v) for v in nw], float(self._q(self._t("decoder.N_conv.bias"))[0]))
        self.asr_res = self._convw("decoder.asr_res.0")
        g = "decoder.generator"
        self.src_w = self._dvec(self._t(f"{g}.m_source.l_linear.weight"))
        self.src_b = float(self._q(self._t(f"{g}.m_source.l_linear.bias"))[0])
        c0 = int(self.cfg["istftnet"]["upsample_initial_channel"])
        nk = len(self.rk)
        self.ups, self.noise_convs, self.noise_res, self.resblocks = [], [], [], []
        for i, (u, k) in enumerate(zip(self.rates, self.kers)):
            cout = c0 // (2 ** (i + 1))
            # stored (Cin, K, Cout); mx.conv_transpose1d receives weight.T = (Cout, K, Cin) (istftnet.py:161-166)
            w_t = self._wn(f"{g}.ups.{i}").permute(2, 1, 0).contiguous()
            self.ups.append(ops.pack_conv_transpose(w_t, self._q(self._t(f"{g}.ups.{i}.bias")), u, d, f16=self.precision == 3 or self.all_f16))
            ncw = self._q(self._t(f"{g}.noise_convs.{i}.weight"))  # (cout, K, n_fft+2)
            ncb = self._q(self._t(f"{g}.noise_convs.{i}.bias"))
            # the harmonic features live in rows of round_up(n_fft + 2, 4) floats (22 -> 24: 16-byte aligned rows, so the strided noise conv reads its
            # taps as float4 runs instead of scalars); the pad columns are zero on both sides
            nbp = ops.round_up(ncw.shape[2], 4)
            ncwp = torch.zeros((ncw.shape[0], ncw.shape[1], nbp), dtype=ncw.dtype)
            ncwp[:, :, : ncw.shape[2]] = ncw
            self.noise_convs.append(ops.pack_conv(ncwp.reshape(ncw.shape[0], 1, -1) if ncw.shape[1] > 1 else ncw, ncb, d, f16=self.all_f16))  # raw phase features: keep hi+lo
            last = i + 1 == len(self.rates)
            self.noise_res.append(self._resblock1(self.bank_dec, f"{g}.noise_res.{i}", cout, 11 if last else 7, (1, 3, 5)))
            for j in range(nk):
                self.resblocks.append(self._resblock1(self.bank_dec, f"{g}.resblocks.{i * nk + j}", cout, self.rk[j], tuple(self.rd[j])))
        self.conv_post = self._convw(f"{g}.conv_post")
        self.style_dec = self.bank_dec.pack(d, self.all_f16)
        self.style_pred = self.bank_pred.pack(d, self.all_f16)
        self.q_style_dec, self.q_style_pred = self.bank_dec.any_q, self.bank_pred.any_q
        # periodic Hann of MLXSTFT (istftnet.py:466) -- same float32 values as dsp.hanning(n, periodic=True)
        n = self.n_fft
        self.window = torch.tensor([0.5 * (1 - math.cos(2 * math.pi * i / n)) for i in range(n)], dtype=torch.float32, device=d)
        self.w = None  # drop the CPU copy

    # ------------------------------------------------------------------ building blocks (device)
    def _new(self, *shape, zero=False):
        return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=self.dev)

    def _gb(self, gb_all: _StyleProj, a: _AdaIN) -> torch.Tensor:
        return (gb_all.quant if a.q else gb_all.plain)[:, a.off: a.off + 2 * a.c]

    def _style_proj(self, s: torch.Tensor, pc: PackedConv, quant: bool) -> _StyleProj:
        """Every ``fc(style)`` of one style vector: one GEMM, and one more from the fake-quantised vector when an AdaIN wants that."""
        # ONE item of B rows (not B items of one row: that is B row tiles with a single valid row each -- 0.53 ms for the 37 396-column projection of
        # 64 utterances, profiles/r6_shape_table_b64_call18.txt)
        B = s.shape[0]
        plain = self._new(1, B, pc.cout)
        self._conv(s[None, :, :], pc, plain)
        q = None
        if quant:
            q = self._new(1, B, pc.cout)
            sq = ops.fake_quant_u8(s[:, None, :])   # per utterance
            self._conv(sq.as_strided((1, B, sq.shape[2]), (B * sq.stride(0), sq.stride(0), 1)), pc, q)
            q = q[0]
        return _StyleProj(plain[0], q)

    def _prec(self, pc) -> int:
        # the weight image decides: fp16-packed weights select 3 themselves (4 in mode 5), MX images 5; bf16 images of modes 3 / 5 (front end) run 2
        return 2 if (self.precision == 3 or (self.precision in (5, 6) and not pc.f16)) else self.precision

    def _conv(self, x, pc, y, **kw):
        kw.setdefault("precision", self._prec(pc))
        return ops.conv_gemm(x, pc, y, **kw)

    def _convq(self, name: str, x, pc, y, *, lens_in=None, pre=None, pre_act=ACT_NONE, pre_slope=0.0, pre_alpha=None, ext_in=None, ext_out=None, ret_ext=False, **kw):
        """``_conv`` of module ``name``.  When the module is flagged for activation quantisation (KittenTTS) its input -- INCLUDING the
        AdaIN / activation prologue the conv would have fused -- is materialised and fake-quantised first (kitten_tts/istftnet.py:131).

        ``ext_in``: per-block, per-channel (min, max) of ``x`` left by the conv that produced it (``ops.new_ext``): the extrema of the prologue's
        output then come from those (every prologue is monotone per channel) instead of one more read of ``x``.  ``ext_out``: a buffer this launch
        should leave ITS output's extrema in for the next quantised conv.  ``ret_ext``: return ``(y, ext_out or None)`` -- None where the launch cannot
        write them (the caller's next conv then sweeps)."""
        want = ret_ext
        done = None
        if self.qmods and self._isq(name):
            if self.fold_quant and "flat" not in kw:
                if ext_in is not None:
                    mm = ops.fake_quant_extrema_from_partials(ext_in, x.shape[1], lens=lens_in, pre=pre, pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha)
                else:   # one read-only extrema pass; the conv quantises inside its prologue (no materialised tensor)
                    mm = ops.fake_quant_extrema(x, lens=lens_in, pre=pre, pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha)
                if ext_out is not None and self.ext_partials and ops.conv_ext_supported(x, pc, B=x.shape[0], lout=kw.get("lout") or y.shape[1], dil=kw.get("dil", 1), pre_act=pre_act,
                                                                         post_act=kw.get("post_act", ACT_NONE), up=kw.get("up"), precision=self._prec(pc)):
                    kw["ext"] = done = ext_out
                self._conv(x, pc, y, lens_in=lens_in, pre=pre, pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha, pre_fq=mm, **kw)
                return (y, done) if want else y
            xq = ops.fake_quant_u8(x, lens=lens_in, pre=pre, pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha)
            self._conv(xq, pc, y, lens_in=lens_in, **kw)
            return (y, None) if want else y
        if pre is not None or pre_act != ACT_NONE:
            kw.update(pre=pre, pre_act=pre_act, pre_slope=pre_slope, pre_alpha=pre_alpha)
        self._conv(x, pc, y, lens_in=lens_in, **kw)
        return (y, None) if want else y

    def _dur_cap(self, bins: int, speed: float) -> int:
        """Upper bound of one predicted duration: the clip (Kokoro), else the sum of ``bins`` sigmoids over the speed (KittenTTS, unclipped)."""
        return 100 if self.max_frames == 0 else (self.max_frames if self.max_frames > 0 else max(100, int(math.ceil(bins / speed)) + 1))

    def _bilstm(self, l: _LSTM, x, out, lens):
        B, L = x.shape[0], x.shape[1]
        xp = self._new(B, L, 8 * l.hid)
        if l.q:
            x = ops.fake_quant_u8(x, lens=lens)
        self._conv(x, l.wx, xp, lens_in=lens, lens_out=lens, flatten=True)
        return ops.lstm_bidir(xp, l.wh, l.hid, out, lens=lens, quant_h=l.q, wh_f16=l.wh_f16, wh_scale=l.wh_scale)

    def _resblk1d_fwd(self, blk: _ResBlk1d, x, gb_all, out, lens, lens2=None):
        """x [B, L, din] -> out [B, L or 2L, dout]."""
        B, L = x.shape[0], x.shape[1]
        up = blk.pool_w is not None
        sc1, sh1 = ops.adain_coef(x, self._gb(gb_all, blk.norm1), lens)
        lo = lens2 if up else lens
        Lo = 2 * L if up else L
        c1 = self._new(B, Lo, blk.dout)
        st = ops.new_stats(B, Lo, blk.dout, self.dev) if self.fuse_stats else None
        nm = blk.name
        if up:
            pooled = self._new(B, Lo, round_up(blk.din, 32), zero=bool(self.qmods))[:, :, : blk.din]
            if self.qmods and self._isq(f"{nm}.pool"):  # the depthwise transposed conv is a ConvWeighted too: it sees fq(leaky(adain(x)))
                xq = ops.fake_quant_u8(x, lens=lens, pre=(sc1, sh1), pre_act=ACT_LEAKY, pre_slope=0.2)
                one, zero = torch.ones_like(sc1), torch.zeros_like(sh1)
                ops.adain_pool_up2(xq, one, zero, 1.0, blk.pool_w, blk.pool_b, pooled, lens)
            else:
                ops.adain_pool_up2(x, sc1, sh1, 0.2, blk.pool_w, blk.pool_b, pooled, lens)
            self._convq(f"{nm}.conv1", pooled, blk.conv1, c1, pad=1, lens_in=lo, lens_out=lo, stats=st)
        else:
            self._convq(f"{nm}.conv1", x, blk.conv1, c1, pad=1, lens_in=lo, lens_out=lo, pre=(sc1, sh1), pre_act=ACT_LEAKY, pre_slope=0.2, stats=st)
        if st is not None:
            sc2, sh2 = ops.adain_from_partials(st, Lo, self._gb(gb_all, blk.norm2), lo)
        else:
            sc2, sh2 = ops.adain_coef(c1, self._gb(gb_all, blk.norm2), lo)
        if blk.conv1x1 is not None:
            short = self._new(B, L, blk.dout)
            # the reference up-samples the shortcut before conv1x1; k = 1, so conv-then-repeat is the same values, and the fake-quantised
            # input has the same extrema either way (nearest repeat adds no new values)
            self._convq(f"{nm}.conv1x1", x, blk.conv1x1, short, lens_in=lens, lens_out=lens, flatten=True)
        else:
            short = x
        self._convq(f"{nm}.conv2", c1, blk.conv2, out, pad=1, lens_in=lo, lens_out=lo, pre=(sc2, sh2), pre_act=ACT_LEAKY, pre_slope=0.2,
                    res=short, res_shift=1 if up else 0, out_scale=1.0 / math.sqrt(2.0))
        return out

    def _resblock1_fwd(self, rb: _ResBlock1, x, gb_all, lens, out=None, accumulate=False, out_scale=1.0, x_stats=None, x_sums=None):
        """AdaINResBlock1.  ``out`` None: returns a fresh tensor; else the last conv writes (or adds) into ``out``.
        Instance-norm statistics of every intermediate tensor come out of the producing conv's epilogue (``stats=``);
        only the block input needs a separate pass, and callers that feed the same input to several blocks share it:
        ``x_sums`` = (float64 sums buffer, already_filled) -- the first block of a stage fills it, the other two only recompute their coefficients."""
        B, L, C = x.shape
        cur = x
        work = None
        tmp = self._new(B, L, C)
        fuse = self.fuse_stats  # never by length: a ragged batch and its single utterances must take the same path
        st_cur = x_stats
        st_tmp = ops.new_stats(B, L, C, self.dev) if fuse else None
        st_work = ops.new_stats(B, L, C, self.dev) if fuse else None
        # KittenTTS activation quantisation: a conv leaves the per-block extrema of its output for the quantised conv that consumes it (the block input
        # of the first round has no producing conv here: that one sweeps)
        use_ext = bool(self.qmods) and self.fold_quant and self.ext_partials
        ex_cur = None
        ex_tmp = ops.new_ext(B, L, C, self.dev) if use_ext else None
        ex_work = ops.new_ext(B, L, C, self.dev) if use_ext else None
        for i, dl in enumerate(rb.dils):
            if fuse and st_cur is not None:
                sc, sh = ops.adain_from_partials(st_cur, L, self._gb(gb_all, rb.adain1[i]), lens)
            elif i == 0 and x_sums is not None:
                sc, sh = ops.adain_coef(cur, self._gb(gb_all, rb.adain1[i]), lens, sums=x_sums[0], reuse=x_sums[1])
            else:
                sc, sh = ops.adain_coef(cur, self._gb(gb_all, rb.adain1[i]), lens)
            _, ex_t = self._convq(f"{rb.name}.convs1.{i}", cur, rb.convs1[i], tmp, dil=dl, pad=(rb.k * dl - dl) // 2, lens_in=lens, lens_out=lens, pre=(sc, sh),
                                  pre_act=ACT_SNAKE, pre_alpha=rb.alpha1[i], stats=st_tmp, ext_in=ex_cur, ext_out=ex_tmp, ret_ext=True)
            if fuse:
                sc, sh = ops.adain_from_partials(st_tmp, L, self._gb(gb_all, rb.adain2[i]), lens)
            else:
                sc, sh = ops.adain_coef(tmp, self._gb(gb_all, rb.adain2[i]), lens)
            last = i == len(rb.dils) - 1
            if last and out is not None:
                dst, acc, scale = out, accumulate, out_scale
            else:
                if work is None:
                    work = self._new(B, L, C)
                dst, acc, scale = work, False, 1.0
            _, ex_w = self._convq(f"{rb.name}.convs2.{i}", tmp, rb.convs2[i], dst, pad=(rb.k - 1) // 2, lens_in=lens, lens_out=lens, pre=(sc, sh),
                                  pre_act=ACT_SNAKE, pre_alpha=rb.alpha2[i], res=cur, accumulate=acc, out_scale=scale,
                                  stats=st_work if (fuse and not last) else None, ext_in=ex_t, ext_out=None if last else ex_work, ret_ext=True)
            cur = dst
            st_cur = st_work if (fuse and not last) else None
            ex_cur = None if last else ex_w
        return cur

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, input_ids: Sequence[torch.Tensor], ref_s: torch.Tensor, speed: float = 1.0,
                forced_durations: Optional[Sequence[torch.Tensor]] = None, rand_ini: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, noise_seed: int = 1234, return_intermediates: bool = False,
                overrides: Optional[Dict[str, torch.Tensor]] = None):
        """input_ids: list of B LongTensors (each INCLUDING the 0 BOS/EOS tokens); ref_s [B, 256] float32.

        Returns (audio list of 1-D float32 tensors on device, pred_dur list of int32 tensors).
        ``rand_ini`` [B, 9] / ``noise`` [B, Lmax, 9] are SineGen's random inputs (generated from
        ``noise_seed`` when omitted).  ``overrides`` may replace intermediate signals: ``f0`` / ``n``
        ([B, 2*Fmax] pitch / energy curves: external prosody control) and ``har`` ([B, frames, n_fft+2] harmonic
        STFT features).  The parity tests use them to teacher-force the vocoder, because SineGen integrates F0
        into a phase (chaotic in F0 rounding) and the reflect-padded edge frames have rounding-noise phases.

        = ``back(front(...))``: ``front`` is everything up to the predicted durations (PL-BERT, duration encoder / predictor, kokoro.py:118-152),
        ``back`` the frame-rate half (text encoder, alignment, F0 / N, decoder, generator).  The split point is the forward pass's one host
        sync (the frame counts size every later buffer); the utterance-sharding layer (mlx_audio_amd/shard.py) re-balances utterances across
        GPUs there, on the real frame counts."""
        st = self.front(input_ids, ref_s, speed=speed, forced_durations=forced_durations, keep_trace=return_intermediates)
        return self.back(st, rand_ini=rand_ini, noise=noise, noise_seed=noise_seed, return_intermediates=return_intermediates, overrides=overrides)

    def front(self, input_ids: Sequence[torch.Tensor], ref_s: torch.Tensor, speed: float = 1.0,
              forced_durations: Optional[Sequence[torch.Tensor]] = None, keep_trace: bool = False,
              ids_padded: Optional[torch.Tensor] = None, forced_padded: Optional[torch.Tensor] = None) -> "KokoroFront":
        """Token-rate half: returns the per-utterance state ``back`` needs (ids, style, duration-encoder output ``d``, durations, frames).
        ``ids_padded`` / ``forced_padded`` (int32 ``[B, >= Tmax]`` on the device, rows zero-padded): the batch as the caller already holds it
        (the shard channel's request block) -- skips the per-utterance re-padding (one small copy kernel per utterance)."""
        dev, hid, sty = self.dev, self.hid, self.sty
        B = len(input_ids)
        Ts = [int(t.numel()) for t in input_ids]
        Tm = max(Ts)
        assert Tm <= self.pb["max_position_embeddings"], (Tm, self.pb["max_position_embeddings"])
        ragged = B > 1
        if ids_padded is not None:
            assert ids_padded.is_cuda and ids_padded.dtype == torch.int32 and ids_padded.shape[0] == B and ids_padded.shape[1] >= Tm
            ids = ids_padded[:, :Tm].contiguous()
        elif all(t.is_cuda for t in input_ids):  # already resident (sharded serving path): pad on the device, no host sync
            ids = torch.nn.utils.rnn.pad_sequence([t.to(device=dev, dtype=torch.int32) for t in input_ids], batch_first=True)
        else:
            ids = torch.zeros((B, Tm), dtype=torch.int32)
            for b, t in enumerate(input_ids):
                ids[b, : Ts[b]] = t.to(torch.int32)
            ids = ids.to(dev)
        lens_t = torch.tensor(Ts, dtype=torch.int32, device=dev) if ragged else None
        ref_s = ref_s.to(device=dev, dtype=torch.float32).contiguous()
        if ref_s.dim() != 2 or ref_s.shape[0] != B or ref_s.shape[1] != 2 * sty:
            raise ValueError(f"ref_s must be [{B}, {2 * sty}] (one style row per utterance), got {tuple(ref_s.shape)}")
        s_dec = ref_s[:, :sty].contiguous()
        s_pred = ref_s[:, sty:].contiguous()
        # ---- every style projection of the network: two GEMMs
        gb_dec = self._style_proj(s_dec, self.style_dec, self.q_style_dec)
        gb_pred = self._style_proj(s_pred, self.style_pred, self.q_style_pred)

        # ---- PL-BERT (CustomAlbert, modules.py:626-655)
        H = self.pb["hidden_size"]
        heads = self.pb["num_attention_heads"]
        E = self.word_emb.shape[1]
        emb = self._new(B, Tm, E)
        ops.gather_rows(self.word_emb, ids, emb, pos_table=self.pos_emb, add_row=self.type_row, lens=lens_t)
        eps = float(self.pb.get("layer_norm_eps", 1e-12))
        ops.layernorm(emb, emb, weight=self.emb_ln[0], bias=self.emb_ln[1], eps=eps, lens=lens_t)
        h = self._new(B, Tm, H)
        enc_n, lay_n = "bert.encoder", "bert.encoder.albert_layer_groups.0.albert_layers.0"
        att_n = f"{lay_n}.attention"
        # (KittenTTS: the encoder / layer / attention MODULES quantise at fixed points of their own forward, kitten_tts.py:108-133,255-268,296-298;
        #  _convq keys on the module that owns the quantisation point, not on the linear layer)
        self._convq(enc_n, emb, self.map_in, h, lens_in=lens_t, lens_out=lens_t, flatten=True)
        qkv = self._new(B, Tm, 3 * H)
        ctx = self._new(B, Tm, H)
        tmp = self._new(B, Tm, H)
        att = self._new(B, Tm, H)
        inter = self._new(B, Tm, self.pb["intermediate_size"])
        for _ in range(self.pb["num_hidden_layers"]):
            self._convq(att_n, h, self.qkv, qkv, lens_in=lens_t, lens_out=lens_t, flatten=True)
            if H // heads in (64, 128):  # f32-MFMA flash kernel; padded keys are invisible (the reference's additive -10000 mask, modules.py:639)
                ops.flash_attention(qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:], ctx, heads=heads, dh=H // heads, lens_q=lens_t, lens_k=lens_t)
            else:
                ops.attention(qkv, heads, H // heads, ctx, lens=lens_t)
            self._convq(att_n, ctx, self.att_dense, tmp, lens_in=lens_t, lens_out=lens_t, res=h, flatten=True)
            ops.layernorm(tmp, att, weight=self.att_ln[0], bias=self.att_ln[1], eps=eps, lens=lens_t)
            self._convq(lay_n, att, self.ffn, inter, lens_in=lens_t, lens_out=lens_t, post_act=self.ffn_act, flatten=True)
            self._convq(lay_n, inter, self.ffn_out, tmp, lens_in=lens_t, lens_out=lens_t, res=att, flatten=True)
            ops.layernorm(tmp, h, weight=self.full_ln[0], bias=self.full_ln[1], eps=eps, lens=lens_t)

        # ---- DurationEncoder (modules.py:380-411): [d_en | style] ping-pong buffers
        da = self._new(B, Tm, hid + sty, zero=True)
        db = self._new(B, Tm, hid + sty, zero=True)
        ops.broadcast_rows(s_pred, da[:, :, hid:], lens=lens_t)
        ops.broadcast_rows(s_pred, db[:, :, hid:], lens=lens_t)
        self._convq("bert_encoder", h, self.bert_encoder, da[:, :, :hid], lens_in=lens_t, lens_out=lens_t, flatten=True)
        cur, nxt = da, db
        for i in range(self.n_layer):
            self._bilstm(self.dur_lstms[i], cur, nxt[:, :, :hid], lens_t)
            ops.layernorm(nxt[:, :, :hid], nxt[:, :, :hid], ada_gb=self._gb(gb_pred, self.dur_adaln[i]), eps=1e-5, lens=lens_t)
            cur, nxt = nxt, cur
        d = cur  # [B, Tm, hid+sty]
        xl = self._new(B, Tm, hid, zero=True)
        self._bilstm(self.pred_lstm, d, xl, lens_t)
        bins = self.dur_proj.cout
        logits = self._new(B, Tm, round_up(bins, 4))
        self._convq("predictor.duration_proj", xl, self.dur_proj, logits[:, :, :bins], lens_in=lens_t, lens_out=lens_t)
        forced = None
        if forced_padded is not None:
            assert forced_padded.is_cuda and forced_padded.dtype == torch.int32 and forced_padded.shape[0] == B and forced_padded.shape[1] >= Tm
            forced = forced_padded[:, :Tm].contiguous()
        elif forced_durations is not None:
            if all(fd.is_cuda for fd in forced_durations):  # resident: pad on the device (a host-side fill would read every row back)
                forced = torch.nn.utils.rnn.pad_sequence([fd.to(torch.int32) for fd in forced_durations], batch_first=True)
                if forced.shape[1] < Tm:
                    forced = torch.nn.functional.pad(forced, (0, Tm - forced.shape[1]))
                forced = forced.contiguous()
            else:
                forced = torch.zeros((B, Tm), dtype=torch.int32)
                for b, fd in enumerate(forced_durations):
                    forced[b, : Ts[b]] = fd.to(torch.int32)
                forced = forced.to(dev)
        idx_cap = Tm * self._dur_cap(bins, float(speed))
        dur, dur_raw, frames, idx = ops.duration_align(logits[:, :, :bins], Tm, B, float(speed), idx_cap, dev, lens=lens_t,
                                                       forced=forced, bins=bins, max_frames=self.max_frames)
        frames_h = frames.cpu()  # the one host sync of the forward pass (kokoro.py:149-152 syncs per phoneme)
        Fs = [int(v) for v in frames_h]
        return KokoroFront(ids=[ids[b, : Ts[b]] for b in range(B)], ref_s=ref_s, d=[d[b, : Ts[b]] for b in range(B)],
                           dur=[dur[b, : Ts[b]] for b in range(B)], frames=Fs, speed=float(speed),
                           trace=dict(dur_raw=dur_raw, bert=h) if keep_trace else None,
                           padded=dict(ids=ids, lens_t=lens_t, d=d, dur=dur, frames=frames, idx=idx, gb_dec=gb_dec, gb_pred=gb_pred))

    def back(self, st: "KokoroFront", rand_ini: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, noise_seed: int = 1234,
             return_intermediates: bool = False, overrides: Optional[Dict[str, torch.Tensor]] = None):
        """Frame-rate half on the utterances of ``st`` (which may have been produced by ``front`` on another GPU: only ids, style, ``d`` and the
        durations travel; the alignment indices are rebuilt here from the durations)."""
        dev, hid, sty = self.dev, self.hid, self.sty
        B = len(st.ids)
        Ts = [int(t.numel()) for t in st.ids]
        Tm = max(Ts)
        ragged = B > 1
        if st.padded is not None:
            pd = st.padded
            ids, lens_t, d, dur, frames, idx, gb_dec, gb_pred = (pd[k] for k in ("ids", "lens_t", "d", "dur", "frames", "idx", "gb_dec", "gb_pred"))
        else:
            ids = torch.nn.utils.rnn.pad_sequence([t.to(device=dev, dtype=torch.int32) for t in st.ids], batch_first=True)
            lens_t = torch.tensor(Ts, dtype=torch.int32, device=dev) if ragged else None
            ref_s = st.ref_s.to(device=dev, dtype=torch.float32).contiguous()
            s_dec = ref_s[:, :sty].contiguous()
            s_pred = ref_s[:, sty:].contiguous()
            gb_dec = self._style_proj(s_dec, self.style_dec, self.q_style_dec)
            gb_pred = self._style_proj(s_pred, self.style_pred, self.q_style_pred)
            d = torch.nn.utils.rnn.pad_sequence([t.to(dev) for t in st.d], batch_first=True).contiguous()
            forced = torch.nn.utils.rnn.pad_sequence([t.to(device=dev, dtype=torch.int32) for t in st.dur], batch_first=True).contiguous()
            cap = max(100, max((int(f) for f in st.frames), default=0) // max(Tm, 1) + 1)
            dur, _, frames, idx = ops.duration_align(None, Tm, B, st.speed, Tm * cap, dev, lens=lens_t, forced=forced)
        Fs = list(st.frames)
        dur_raw = st.trace["dur_raw"] if st.trace else None
        h = st.trace["bert"] if st.trace else None
        Fm = max(Fs)
        if Fm <= 0:
            return [self._new(1, zero=True) for _ in range(B)], [dur[b, : Ts[b]] for b in range(B)]
        idx = idx[:, :Fm]
        lens_f = frames if ragged else None
        lens_2f = frames * 2 if ragged else None

        # ---- text encoder (modules.py:21-68)
        x0 = self._new(B, Tm, hid)
        ops.gather_rows(self.te_emb, ids, x0, lens=lens_t)
        x1 = self._new(B, Tm, hid)
        for i, (pc, lw, lb) in enumerate(self.te_cnn):
            self._convq(f"text_encoder.cnn.{i}.0", x0, pc, x1, pad=(self.te_k - 1) // 2, lens_in=lens_t, lens_out=lens_t)
            ops.layernorm(x1, x0, weight=lw, bias=lb, eps=1e-5, lens=lens_t, post_act=ACT_LEAKY, post_slope=0.2)
        t_en = self._new(B, Tm, hid, zero=True)
        self._bilstm(self.te_lstm, x0, t_en, lens_t)

        # ---- alignment (kokoro.py:148-169): gathers instead of one-hot matmuls
        en = self._new(B, Fm, hid + sty)
        ops.gather_rows(d, idx, en, per_batch=True, lens=lens_f)
        c_in = hid + 2
        dec_in = self._new(B, Fm, round_up(c_in, 32), zero=True)
        ops.gather_rows(t_en, idx, dec_in[:, :, :hid], per_batch=True, lens=lens_f)
        asr = dec_in[:, :, :hid]

        # ---- F0 / N predictor (modules.py:355-377)
        xs = self._new(B, Fm, hid, zero=True)
        self._bilstm(self.shared, en, xs, lens_f)
        curves = []
        for blocks, proj, proj_n in ((self.f0_blocks, self.f0_proj, "predictor.F0_proj"), (self.n_blocks, self.n_proj, "predictor.N_proj")):
            y0 = self._new(B, Fm, blocks[0].dout)
            self._resblk1d_fwd(blocks[0], xs, gb_pred, y0, lens_f)
            y1 = self._new(B, 2 * Fm, blocks[1].dout)
            self._resblk1d_fwd(blocks[1], y0, gb_pred, y1, lens_f, lens_2f)
            y2 = self._new(B, 2 * Fm, blocks[2].dout)
            self._resblk1d_fwd(blocks[2], y1, gb_pred, y2, lens_2f)
            curve = self._new(B, 2 * Fm, 1, zero=True)
            self._convq(proj_n, y2, proj, curve, lens_in=lens_2f, lens_out=lens_2f)
            curves.append(curve[:, :, 0])
        f0_curve, n_curve = curves
        overrides = overrides or {}
        if "f0" in overrides:
            f0_curve = overrides["f0"].to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(f0_curve.shape) == (B, 2 * Fm)
        if "n" in overrides:
            n_curve = overrides["n"].to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(n_curve.shape) == (B, 2 * Fm)

        # ---- decoder (istftnet.py:981-997)
        f0_in, n_in = f0_curve, n_curve
        if self.qmods and self._isq("decoder.F0_conv"):
            f0_in = ops.fake_quant_u8(f0_curve[:, :, None], y=torch.zeros_like(f0_curve)[:, :, None], lens=lens_2f)[:, :, 0]
        if self.qmods and self._isq("decoder.N_conv"):
            n_in = ops.fake_quant_u8(n_curve[:, :, None], y=torch.zeros_like(n_curve)[:, :, None], lens=lens_2f)[:, :, 0]
        ops.conv1d_c1_k3s2(f0_in, self.f0_conv[0], self.f0_conv[1], dec_in, hid, lens_in=lens_2f)
        ops.conv1d_c1_k3s2(n_in, self.n_conv[0], self.n_conv[1], dec_in, hid + 1, lens_in=lens_2f)
        cd, gd, ad = self.cdim, self.gdim, self.adim
        c_cat = cd + ad + 2
        ca = self._new(B, Fm, round_up(c_cat, 32), zero=True)
        cb = self._new(B, Fm, round_up(c_cat, 32), zero=True)
        trace = {}
        self._resblk1d_fwd(self.enc_blk, dec_in[:, :, :c_in], gb_dec, ca[:, :, :cd], lens_f)
        if return_intermediates:
            trace.update(dec_in=dec_in[:, :, :c_in].clone(), enc=ca[:, :, :cd].clone())
        self._convq("decoder.asr_res.0", asr, self.asr_res, ca[:, :, cd:cd + ad], lens_in=lens_f, lens_out=lens_f, flatten=True)
        ca[:, :, cd + ad:c_cat].copy_(dec_in[:, :, hid:hid + 2])
        cb[:, :, cd:c_cat].copy_(ca[:, :, cd:c_cat])
        cur, nxt = ca, cb
        for i in range(3):
            self._resblk1d_fwd(self.dec_blks[i], cur[:, :, :c_cat], gb_dec, nxt[:, :, :cd], lens_f)
            cur, nxt = nxt, cur
            if return_intermediates:
                trace[f"dec{i}"] = cur[:, :, :cd].clone()
        xg = self._new(B, 2 * Fm, gd)
        self._resblk1d_fwd(self.dec_blks[3], cur[:, :, :c_cat], gb_dec, xg, lens_f, lens_2f)

        # ---- generator (istftnet.py:797-835)
        up = self.total_up
        L2 = 2 * Fm
        if rand_ini is None or noise is None:
            # SineGen's uniform initial phases and gaussian noise (istftnet.py:581,649), drawn on the device
            gen = torch.Generator(device=dev).manual_seed(noise_seed)
            rand_ini = torch.rand((B, 9), generator=gen, device=dev, dtype=torch.float32)
            noise = torch.randn((B, L2 * up, 9), generator=gen, device=dev, dtype=torch.float32)
        rand_ini = rand_ini.to(dev).contiguous()
        noise = noise.to(dev).contiguous()
        g_n = "decoder.generator"
        har_src = ops.sine_source(f0_curve, rand_ini, noise, self.src_w, self.src_b, up, lens2=lens_2f,
                                  quant=bool(self.qmods) and self._isq(f"{g_n}.m_source.l_linear"), coarse_f32=self.coarse_f32)
        nb2 = self.n_fft + 2
        nbp = ops.round_up(nb2, 4)
        n_har = L2 * up // self.hop + 1
        har = self._new(B, n_har, nbp, zero=True)[:, :, :nb2]   # rows of nbp floats, pad columns zero (see the noise convs' weight images)
        lens_samples = frames * (2 * up) if ragged else None
        ops.stft_magphase(har_src, self.n_fft, self.hop, self.window, har, lens=lens_samples)
        if "har" in overrides:
            har.copy_(overrides["har"].to(device=dev, dtype=torch.float32))
        lens_har = lens_samples // self.hop + 1 if ragged else None
        x = xg
        L = L2
        lens_x = lens_2f
        nk = len(self.rk)
        for i, (u, k) in enumerate(zip(self.rates, self.kers)):
            last = i + 1 == len(self.rates)
            cout = self.ups[i].cout // u
            Lo = L * u + (1 if last else 0)
            lens_o = (lens_x * u + (1 if last else 0)) if ragged else None
            # harmonic branch
            xsrc = self._new(B, Lo, cout)
            har_i = har
            if self.qmods and self._isq(f"{g_n}.noise_convs.{i}"):  # a copy in the same padded row layout: the strided noise conv reads its taps as one flat run
                har_i = ops.fake_quant_u8(har, y=self._new(B, n_har, nbp, zero=True)[:, :, :nb2], lens=lens_har)
            if not last:
                sf = int(np.prod(self.rates[i + 1:]))
                self._conv(har_i, self.noise_convs[i], xsrc, lens_in=lens_har, lens_out=lens_o,
                           flat=dict(ldx=sf * nbp, x_off=-((sf + 1) // 2) * nbp, channels=nbp))
            else:
                self._conv(har_i, self.noise_convs[i], xsrc, lens_in=lens_har, lens_out=lens_o)
            if return_intermediates:
                trace[f"nconv{i}"] = xsrc.clone()
            xsrc = self._resblock1_fwd(self.noise_res[i], xsrc, gb_dec, lens_o)
            if return_intermediates:
                trace[f"nres{i}"] = xsrc.clone()
            # up-sampler: leaky_relu(0.1) prologue, polyphase GEMM, "+ x_source" epilogue
            kp = k // u
            xu = self._new(B, Lo, cout)
            lens_gemm = (lens_x + (kp - 1)) if ragged else None
            lens_ct = (lens_x * u) if ragged else None
            self._convq(f"{g_n}.ups.{i}", x, self.ups[i], xu, pad=kp - 1, lout=L + kp - 1, lens_in=lens_x, lens_out=lens_gemm, pre_act=ACT_LEAKY,
                        pre_slope=0.1, res=xsrc, up=dict(s=u, p=(k - u) // 2, cout=cout, row_off=1 if last else 0, lout=L * u, lens=lens_ct))
            if last:
                xu[:, 0, :].copy_(xsrc[:, 0, :])  # the zero left-pad row of "reflection_pad" + x_source
            if return_intermediates:
                trace[f"xu{i}"] = xu.clone()
            acc = self._new(B, Lo, cout)
            xu_sums = torch.empty(B * cout * 2, dtype=torch.float64, device=dev)  # instance-norm statistics of xu: one pass for the nk resblocks
            for j in range(nk):
                self._resblock1_fwd(self.resblocks[i * nk + j], xu, gb_dec, lens_o, out=acc, accumulate=j > 0,
                                    out_scale=(1.0 / nk) if j == nk - 1 else 1.0, x_sums=(xu_sums, j > 0))
            x, L, lens_x = acc, Lo, lens_o
            if return_intermediates:
                trace[f"stage{i}"] = acc.clone()
        post = self._new(B, L, round_up(nb2, 4))
        self._convq(f"{g_n}.conv_post", x, self.conv_post, post[:, :, :nb2], pad=3, lens_in=lens_x, lens_out=lens_x, pre_act=ACT_LEAKY, pre_slope=0.01)
        audio = self._new(B, (L - 1) * self.hop, zero=True)
        ops.istft_head(post[:, :, :nb2], self.n_fft, self.hop, self.window, audio, lens=lens_x)
        outs = [audio[b, : Fs[b] * 2 * up] for b in range(B)]
        durs = [dur[b, : Ts[b]] for b in range(B)]
        if return_intermediates:
            return outs, durs, dict(d=d, en=en, f0=f0_curve, n=n_curve, asr=asr, t_en=t_en, dur_raw=dur_raw, har_src=har_src,
                                    har=har, bert=h, xg=xg, post=post[:, :, :nb2], dec3=xg, **trace)
        return outs, durs
