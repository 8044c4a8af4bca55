"""Whisper on MI355X: the host-side schedule over the HIP kernels (encoder, decoder with KV caches, decode loop).

Mirrors ``AudioEncoder`` / ``TextDecoder`` / ``ResidualAttentionBlock`` (``stt/models/whisper/whisper.py:338-498``) and
``DecodingTask._main_loop`` with its logit filters and ``GreedyDecoder`` (``decoding.py:302-443, 588-632``), with the
reference's op-by-op graph collapsed into:

  * conv stem: conv1 (+GELU) is one implicit-GEMM launch; the stride-2 conv2 is re-expressed on the host as a
    stride-1, 2-tap conv over pairs of frames (rows of 2*C channels) so it runs on the aligned MFMA path, with GELU and
    the sinusoidal position add in its epilogue;
  * every Linear is conv_gemm (fp16 weights; ``precision`` 4 = fp16 hi+lo activations, 3 = the reference's own fp16
    activation rounding) with bias / GELU / residual fused; q, k, v share one GEMM (K has no bias: zero entries);
  * attention is ``mi355_flash_attention`` (f32 MFMA flash kernel for the encoder and the prefill, the KV-streaming kernel
    for decode steps) reading K / V straight out of the caches the projection GEMMs wrote into;
  * decode steps (1 row per sequence) switch every Linear to the HBM-bound ``mi355_gemv`` on the row-major fp16 image, and
    the whole filter + arg-max + log-prob bookkeeping of a step is one kernel -- the loop runs without host round trips
    (completion is polled every ``poll`` steps; results do not depend on when the loop stops).

Everything here is plumbing (allocation, views, launch order); all arithmetic on activations is in libmi355audio.so.
"""
from __future__ import annotations

import ctypes
import os
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .... import _lib, ops
from ....ops import ACT_GELU, ACT_NONE, PackedConv, RowMajor16
from .synthetic import ModelDimensions


@dataclass
class _Lin:
    pc: PackedConv      # MFMA fragment order (prefill / encoder)
    rm: RowMajor16      # row-major fp16 (decode-step GEMV)


@dataclass
class _LN:
    w: torch.Tensor
    b: torch.Tensor


@dataclass
class _Block:
    attn_ln: _LN
    qkv: _Lin            # self-attention q | k | v
    q: _Lin              # decoder only: q alone (k | v go straight into the cache)
    kv: _Lin
    out: _Lin
    cross_ln: Optional[_LN]
    cq: Optional[_Lin]
    ckv: Optional[_Lin]
    cout: Optional[_Lin]
    mlp_ln: _LN
    mlp1: _Lin
    mlp2: _Lin


def sinusoids(length: int, channels: int, max_timescale: float = 10000) -> torch.Tensor:
    """whisper.py:329-335 (host, float32)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    st = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


class WhisperEngine:
    def __init__(self, weights: Dict[str, torch.Tensor], dims: ModelDimensions, device: str = "cuda:0", precision: int = 4,
                 kv_dtype: torch.dtype = torch.float16):
        ops.require_gpu()
        assert precision in (3, 4)
        assert kv_dtype in ops.KV_DTYPES
        self.kv_dtype = kv_dtype  # K | V caches: Model.load_weights passes the checkpoint's floating dtype (the reference's cache dtype); default fp16 = the published checkpoints
        self.dims = dims
        self.device = torch.device(device)
        self.precision = precision
        # encoder linears behind a LayerNorm read PRE-SPLIT activations (precision 4): the LayerNorm kernel leaves the fp16 hi | lo words, so the conversion
        # happens once per value instead of once per 128-column tile of the linear.  Same numbers, bit for bit; measured -3 .. -5 % on those launches
        # (profiles/r6_conv_big_gemm_fastw_b64_call17.txt).  A conv epilogue can leave the words too (y_split), but that costs the producing launch more
        # than the consumer gains (+30 %), so the GELU -> mlp2 and attention -> out hand-overs stay float32.  MI355_WHISPER_SPLIT=0: A/B knob
        self.split_acts = precision == 4 and os.environ.get("MI355_WHISPER_SPLIT", "1") != "0"
        self.native_decode = True  # single-token decoder steps run through mi355_stack_decode_step
        # windows per step from which the decoder runs on the rows pipeline (tile images x input planes; stack_step.cpp tall_step); below: the
        # one-row-per-wave / 5..8-row matrix-pipe GEMV kernels.  MI355_WHISPER_ROWS_MIN: A/B knob
        self.rows_min = int(os.environ.get("MI355_WHISPER_ROWS_MIN", "9"))
        self._rows_ws = None
        self._logit_planes = None
        self._tile_images = {}
        self.dh = dims.n_audio_state // dims.n_audio_head
        assert self.dh in (64, 128) and dims.n_text_state // dims.n_text_head == self.dh
        assert max(dims.n_audio_state, dims.n_text_state) <= 1024, "layernorm kernel holds <= 1024 channels per row"
        w = {k: v.detach().to(torch.float16).to(torch.float32).cpu() for k, v in weights.items()}  # checkpoint dtype: fp16
        dev = self.device

        def lin(wt, bias):
            return _Lin(ops.pack_conv(wt, bias, dev, f16=True), ops.pack_rowmajor16(wt, bias, dev, f16=True))

        def ln(name):
            return _LN(w[name + ".weight"].to(dev), w[name + ".bias"].to(dev))

        def block(pfx, n, cross):
            def att(a):
                wq, bq = w[f"{pfx}.{a}.query.weight"], w[f"{pfx}.{a}.query.bias"]
                wk = w[f"{pfx}.{a}.key.weight"]
                wv, bv = w[f"{pfx}.{a}.value.weight"], w[f"{pfx}.{a}.value.bias"]
                z = torch.zeros(n)
                return (lin(torch.cat([wq, wk, wv]), torch.cat([bq, z, bv])), lin(wq, bq), lin(torch.cat([wk, wv]), torch.cat([z, bv])),
                        lin(w[f"{pfx}.{a}.out.weight"], w[f"{pfx}.{a}.out.bias"]))

            qkv, q, kv, out = att("attn")
            cq = ckv = cout = cln = None
            if cross:
                _, cq, ckv, cout = att("cross_attn")
                cln = ln(f"{pfx}.cross_attn_ln")
            return _Block(ln(f"{pfx}.attn_ln"), qkv, q, kv, out, cln, cq, ckv, cout, ln(f"{pfx}.mlp_ln"),
                          lin(w[f"{pfx}.mlp1.weight"], w[f"{pfx}.mlp1.bias"]), lin(w[f"{pfx}.mlp2.weight"], w[f"{pfx}.mlp2.bias"]))

        na = dims.n_audio_state
        self.conv1 = ops.pack_conv(w["encoder.conv1.weight"], w["encoder.conv1.bias"], dev, f16=True)
        # conv2 (k3, stride 2, pad 1) on rows of frame pairs r = (x[2r], x[2r+1]): out[t] = W0 x[2t-1] + W1 x[2t] + W2 x[2t+1]
        #   = tap0 . row[t-1] (second half only) + tap1 . row[t]
        w2 = w["encoder.conv2.weight"]
        wp = torch.zeros(na, 2, 2 * na)
        wp[:, 0, na:] = w2[:, 0, :]
        wp[:, 1, :na] = w2[:, 1, :]
        wp[:, 1, na:] = w2[:, 2, :]
        self.conv2 = ops.pack_conv(wp, w["encoder.conv2.bias"], dev, f16=True)
        self.enc_pos = sinusoids(dims.n_audio_ctx, na).to(torch.float16).to(torch.float32).to(dev)[None]  # whisper.py:434 .astype(dtype)
        self.enc_blocks = [block(f"encoder.blocks.{i}", na, False) for i in range(dims.n_audio_layer)]
        self.ln_post = ln("encoder.ln_post")
        self.tok_emb = w["decoder.token_embedding.weight"].to(dev)
        self.pos_emb = w["decoder.positional_embedding"].to(dev)
        self.dec_blocks = [block(f"decoder.blocks.{i}", dims.n_text_state, True) for i in range(dims.n_text_layer)]
        self.ln = ln("decoder.ln")
        self.logits_lin = lin(w["decoder.token_embedding.weight"], None)  # tied: token_embedding.as_linear (whisper.py:498)

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _linear(self, x: torch.Tensor, l: _Lin, y: torch.Tensor, post_act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
                ln: Optional[_LN] = None, y2: Optional[torch.Tensor] = None):
        """y = act(LN(x) W^T + b) + res on [B, L, C] views; decode steps (L == 1, B <= 64) take the GEMV (1..8 rows: one row per wave / matrix-pipe
        kernels, 9..64 rows: gemm_rows.hip), which also fuses the pre-LayerNorm (``ln``) and can split its columns over two destinations
        (``y2``: q -> y, k | v -> the KV-cache slot)."""
        B, L, _ = x.shape
        if L == 1 and B <= 64:
            ops.gemv(x[:, 0, :], l.rm, y[:, 0, :], post_act=post_act, res=None if res is None else res[:, 0, :],
                     norm=None if ln is None else ("layer", ln.w, ln.b, 1e-5), y2=None if y2 is None else y2[:, 0, :])
        else:
            assert y2 is None
            if ln is not None:
                x = self._lnorm(x, ln)
            ops.conv_gemm(x, l.pc, y, post_act=post_act, res=res, precision=self.precision)
        return y

    def _lnorm(self, x, p: _LN):
        return ops.layernorm(x, torch.empty_like(x), weight=p.w, bias=p.b, eps=1e-5)

    # ------------------------------------------------------------------ encoder (whisper.py:438-448)
    def encode(self, mel: torch.Tensor, return_layers: bool = False):
        d = self.dims
        mel = mel.to(self.device, torch.float32).contiguous()
        B, Fr, nm = mel.shape
        assert nm == d.n_mels and Fr == 2 * d.n_audio_ctx, "incorrect audio shape"
        na, H, dh = d.n_audio_state, d.n_audio_head, self.dh
        T = d.n_audio_ctx
        y1 = self._f(B, Fr, na)
        ops.conv_gemm(mel, self.conv1, y1, pad=1, post_act=ACT_GELU, precision=self.precision)
        x = self._f(B, T, na)
        ops.conv_gemm(y1.view(B, T, 2 * na), self.conv2, x, pad=1, post_act=ACT_GELU, res=self.enc_pos.expand(B, T, na),
                      precision=self.precision)
        layers = [x.clone()] if return_layers else None
        qkv = self._f(B, T, 3 * na)
        kv16 = torch.empty((B, T, 2 * na), dtype=self.kv_dtype, device=self.device) if self.kv_dtype != torch.float32 else None
        att = self._f(B, T, na)
        mid = self._f(B, T, 4 * na)
        sp = 4 if self.split_acts else 0
        hbuf = self._f(B, T, na) if sp else None
        for blk in self.enc_blocks:
            if sp:
                self._encoder_block_split(blk, x, hbuf, qkv, kv16, att, mid, H, dh)
                if return_layers:
                    layers.append(x.clone())
                continue
            h = self._lnorm(x, blk.attn_ln)
            self._linear(h, blk.qkv, qkv)
            if self.kv_dtype == torch.float32:
                ops.flash_attention(qkv[:, :, 0:na], qkv[:, :, na:2 * na], qkv[:, :, 2 * na:], att, heads=H, dh=dh, scale=dh ** -0.5)
            else:  # K | V in the checkpoint's 16-bit type: both contractions on the 16-bit matrix pipe (flash_attn16_kernel)
                kv16.copy_(qkv[:, :, na:])
                ops.flash_attention(qkv[:, :, 0:na], kv16[:, :, :na], kv16[:, :, na:], att, heads=H, dh=dh, scale=dh ** -0.5)
            self._linear(att, blk.out, x, res=x)
            h = self._lnorm(x, blk.mlp_ln)
            self._linear(h, blk.mlp1, mid, post_act=ACT_GELU)
            self._linear(mid, blk.mlp2, x, res=x)
            if return_layers:
                layers.append(x.clone())
        out = self._lnorm(x, self.ln_post)
        return (out, layers) if return_layers else out

    def _encoder_block_split(self, blk, x, h, qkv, kv16, att, mid, H, dh):
        """One ResidualAttentionBlock of the encoder (whisper.py:397-420) with the two linears behind a LayerNorm reading pre-split activations (``split_acts``)."""
        na = x.shape[2]
        ops.layernorm(x, h, weight=blk.attn_ln.w, bias=blk.attn_ln.b, eps=1e-5, split=4)
        ops.conv_gemm(h, blk.qkv.pc, qkv, precision=4, x_split=True)
        if self.kv_dtype == torch.float32:
            ops.flash_attention(qkv[:, :, 0:na], qkv[:, :, na:2 * na], qkv[:, :, 2 * na:], att, heads=H, dh=dh, scale=dh ** -0.5)
        else:
            kv16.copy_(qkv[:, :, na:])
            ops.flash_attention(qkv[:, :, 0:na], kv16[:, :, :na], kv16[:, :, na:], att, heads=H, dh=dh, scale=dh ** -0.5)
        ops.conv_gemm(att, blk.out.pc, x, res=x, precision=4)
        ops.layernorm(x, h, weight=blk.mlp_ln.w, bias=blk.mlp_ln.b, eps=1e-5, split=4)
        ops.conv_gemm(h, blk.mlp1.pc, mid, post_act=ACT_GELU, precision=4, x_split=True)
        ops.conv_gemm(mid, blk.mlp2.pc, x, res=x, precision=4)

    # ------------------------------------------------------------------ decoder (whisper.py:476-498)
    def new_state(self, xa: torch.Tensor) -> dict:
        """Allocates the self-attention KV caches and fills the cross-attention K / V (computed once, whisper.py:361-365)."""
        d = self.dims
        B, T, nt = xa.shape[0], xa.shape[1], d.n_text_state
        st = dict(B=B, n=0, self_kv=[self._f(B, d.n_text_ctx, 2 * nt) for _ in self.dec_blocks], cross_k=[], cross_v=[])
        H, dh = d.n_text_head, self.dh
        for blk in self.dec_blocks:
            ckv = self._f(B, T, 2 * nt)
            ops.conv_gemm(xa, blk.ckv.pc, ckv, precision=self.precision)
            # head-major [B, H, T, dh] copies (layout only, once per window): every decode step streams all 1500 keys of each head, and
            # with heads packed inside 12 KB rows a head's keys are 6 KB apart -- they all land on the same one or two L2 channels
            # ... and in the checkpoint's 16-bit type, as the reference keeps them (whisper.py:360-361: k, v of the cross-attention are computed
            # once in the model dtype and cached): the decode step is bound by streaming these 2 x B x 1500 x n_state values per layer
            if self.kv_dtype != torch.float32 and dh % 8 == 0:   # one pass: float32 rows -> the two 16-bit head-major blocks
                blocks = ops.kv_head_major16(ckv, 2, H, dh, self.kv_dtype)
                st["cross_k"].append(blocks[0])
                st["cross_v"].append(blocks[1])
                continue
            st["cross_k"].append(ckv[:, :, :nt].reshape(B, T, H, dh).permute(0, 2, 1, 3).to(self.kv_dtype).contiguous())
            st["cross_v"].append(ckv[:, :, nt:].reshape(B, T, H, dh).permute(0, 2, 1, 3).to(self.kv_dtype).contiguous())
        return st

    def _native_desc(self, st: dict):
        """C descriptor of the decoder for mi355_stack_decode_step (self-attention, cross-attention over this window's K | V, GELU MLP).  Steps of
        ``rows_min`` .. 64 windows also name the tile images and the rows workspace (the rows pipeline of rows_pipe.hip)."""
        if "native" in st:
            return st["native"]
        tall = st["B"] >= self.rows_min
        d = self.dims
        LD, SD = _lib.STRUCTS["mi355_layer_desc"], _lib.STRUCTS["mi355_stack_desc"]
        arr = (LD * d.n_text_layer)()
        p = ops._ptr
        for i, blk in enumerate(self.dec_blocks):
            a = arr[i]
            a.wqkv, a.bqkv = p(blk.qkv.rm.w), p(blk.qkv.rm.bias)
            a.wo, a.bo = p(blk.out.rm.w), p(blk.out.rm.bias)
            a.w_in, a.b_in = p(blk.mlp1.rm.w), p(blk.mlp1.rm.bias)
            a.w_out, a.b_out = p(blk.mlp2.rm.w), p(blk.mlp2.rm.bias)
            a.attn_norm_w, a.attn_norm_b = p(blk.attn_ln.w), p(blk.attn_ln.b)
            a.mlp_norm_w, a.mlp_norm_b = p(blk.mlp_ln.w), p(blk.mlp_ln.b)
            kv = st["self_kv"][i]
            a.kv, a.kv_bstride, a.kv_capacity = kv.data_ptr(), kv.stride(0), kv.shape[1]
            a.wcq, a.bcq, a.wco, a.bco = p(blk.cq.rm.w), p(blk.cq.rm.bias), p(blk.cout.rm.w), p(blk.cout.rm.bias)
            a.cross_norm_w, a.cross_norm_b = p(blk.cross_ln.w), p(blk.cross_ln.b)
            ck, cv = st["cross_k"][i], st["cross_v"][i]
            a.cross_k, a.cross_v = ck.data_ptr(), cv.data_ptr()
            a.cross_bstride, a.cross_hstride, a.cross_ld, a.cross_len = ck.stride(0), ck.stride(1), ck.stride(2), ck.shape[2]
            a.cross_kv_dtype = ops.KV_DTYPES[ck.dtype]
            if tall:   # tile images for the rows pipeline (steps of 5..64 windows), built once per engine
                tl = lambda l: ops._ptr(self._tiles(l).w)
                a.wqkv_t, a.wo_t, a.w_in_t, a.w_out_t, a.wcq_t, a.wco_t = tl(blk.qkv), tl(blk.out), tl(blk.mlp1), tl(blk.mlp2), tl(blk.cq), tl(blk.cout)
        sd = SD()
        sd.n_layers, sd.d_model, sd.heads, sd.kv_heads, sd.dh, sd.d_ff = d.n_text_layer, d.n_text_state, d.n_text_head, d.n_text_head, self.dh, 4 * d.n_text_state
        sd.norm, sd.eps, sd.glu, sd.act, sd.wdtype, sd.causal, sd.window, sd.attn_scale = 1, 1e-5, 0, ACT_GELU, 1, 1, 0, 0.0
        sd.layers = ctypes.cast(arr, ctypes.c_void_p)
        sws, scnt = ops.attn_split_workspace(self.device, 8 * d.n_text_head, self.dh)  # key-split decode attention (long key ranges)
        sd.attn_split_ws, sd.attn_split_cnt = sws.data_ptr(), scnt.data_ptr()
        gws, gcnt = ops.gemv_split_workspace(self.device, d.n_text_state, 4 * d.n_text_state)   # mlp2 (K = 4 n_state): K split over workgroups
        sd.gemv_split_ws, sd.gemv_split_cnt = gws.data_ptr(), gcnt.data_ptr()
        sd.final_norm_w, sd.final_norm_b = p(self.ln.w), p(self.ln.b)
        if tall:
            if self._rows_ws is None:
                need = int(_lib.load().mi355_stack_rows_ws_bytes(ctypes.byref(sd), 64))
                assert need > 0
                self._rows_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            sd.rows_ws, sd.rows_ws_bytes = self._rows_ws.data_ptr(), self._rows_ws.numel()
        st["native"] = dict(arr=arr, desc=sd)
        return st["native"]

    def _tiles(self, l: _Lin):
        t = self._tile_images.get(id(l))
        if t is None:
            t = self._tile_images[id(l)] = ops.tiles16_from_rowmajor(l.rm)
        return t

    def decoder_step(self, tokens: torch.Tensor, st: dict) -> torch.Tensor:
        """tokens int32 [B, n] (device view) appended at offset st['n'] -> final-LN hidden states [B, n, n_text_state]."""
        d = self.dims
        B, n = tokens.shape
        off = st["n"]
        assert off + n <= d.n_text_ctx
        nt, H, dh = d.n_text_state, d.n_text_head, self.dh
        x = self._f(B, n, nt)
        ops.gather_rows(self.tok_emb, tokens, x, pos_table=self.pos_emb[off:off + n])
        if n == 1 and B <= 64 and self.native_decode:  # the whole 12-layer step from the native runner: one call instead of ~100 launches from Python
            nd = self._native_desc(st)
            ws = self._f(B * (2 * nt + 4 * nt + 2 * nt))
            out = self._f(B, 1, nt)
            rc = _lib.load().mi355_stack_decode_step(ctypes.byref(nd["desc"]), x.data_ptr(), B, off, ws.data_ptr(), out.data_ptr(), ops._stream())
            _lib.check(rc, "mi355_stack_decode_step")
            st["n"] = off + 1
            return out
        q = self._f(B, n, nt)
        att = self._f(B, n, nt)
        mid = self._f(B, n, 4 * nt)
        for i, blk in enumerate(self.dec_blocks):
            cache = st["self_kv"][i]
            if n == 1 and B <= 64:  # decode step: LayerNorm + q | k | v in one GEMV, k | v written straight into the cache slot
                self._linear(x, blk.qkv, q, ln=blk.attn_ln, y2=cache[:, off:off + 1, :])
            else:
                h = self._lnorm(x, blk.attn_ln)
                self._linear(h, blk.q, q)
                self._linear(h, blk.kv, cache[:, off:off + n, :])
            ops.flash_attention(q, cache[:, :off + n, 0:nt], cache[:, :off + n, nt:], att, heads=H, dh=dh, scale=dh ** -0.5, causal=True)
            self._linear(att, blk.out, x, res=x)
            self._linear(x, blk.cq, q, ln=blk.cross_ln)
            ops.flash_attention(q, st["cross_k"][i], st["cross_v"][i], att, heads=H, dh=dh, scale=dh ** -0.5, head_major=True)
            self._linear(att, blk.cout, x, res=x)
            self._linear(x, blk.mlp1, mid, post_act=ACT_GELU, ln=blk.mlp_ln)
            self._linear(mid, blk.mlp2, x, res=x)
        st["n"] = off + n
        return self._lnorm(x, self.ln)

    def logits(self, hidden: torch.Tensor) -> torch.Tensor:
        """hidden [B, n, nt] -> [B, n, V_padded] fp32 (columns >= n_vocab are padding)."""
        B, n, _ = hidden.shape
        vp = ops.round_up(self.dims.n_vocab, 4)
        out = self._f(B, n, vp)
        if n == 1 and self.rows_min <= B <= 64 and os.environ.get("MI355_WHISPER_LOGITS_ROWS", "1") != "0":
            # decode step of a tall batch: the 51 865 x 768 tied head on the rows pipeline (tile image x pre-split planes, ONE K group: the slab a
            # workgroup writes IS the logits -- no bias, no epilogue launch).  The one-launch gemm_rows kernel re-reads and re-splits all B rows in each of its
            # 3 242 workgroups: 86 us per step at 64 windows against ~30 us here (profiles/r6_kernel_stats_whisper_b64_call16.txt).  MI355_WHISPER_LOGITS_ROWS=0: A/B knob
            R = ops.rows_R(B)
            nt = self.dims.n_text_state
            if self._logit_planes is None or self._logit_planes[0] != R:
                self._logit_planes = (R, ops.rows_planes(R, nt, self.device))
            planes = self._logit_planes[1]
            ops.rows_finish(hidden[:, 0, :], B, nt, 1, planes=planes, R=R, f16=True)   # float rows -> fp16 hi + lo planes
            ops.rows_gemm(planes, self._tiles(self.logits_lin), out.view(1, B, vp), B, R, kgroups=1)
            return out
        self._linear(hidden, self.logits_lin, out[:, :, :self.dims.n_vocab])
        return out

    # ------------------------------------------------------------------ decode loop (decoding.py:588-632)
    def decode(self, mel: Optional[torch.Tensor], tok, *, sample_len: Optional[int] = None, without_timestamps: bool = False,
               suppress_blank: bool = True, suppress_tokens: Optional[Sequence[int]] = None, max_initial_timestamp: Optional[float] = 1.0,
               forced_tokens: Optional[torch.Tensor] = None, audio_features: Optional[torch.Tensor] = None, record: bool = False,
               poll: int = 16, fixed_steps: bool = False, initial_tokens: Optional[Sequence[int]] = None, temperature: float = 0.0,
               generator: Optional[torch.Generator] = None):
        """``initial_tokens``: the full initial sequence (``[sot_prev] + prompt + sot_sequence + prefix``, decoding.py:525-551) when the caller
        conditions on a prompt; ``temperature`` > 0 samples ``categorical(logits / T)`` (decoding.py:266-269) as an arg-max over
        ``logits / T + Gumbel(0, 1)`` with the noise drawn on the device from ``generator``."""
        d = self.dims
        xa = self.encode(mel) if audio_features is None else audio_features.to(self.device, torch.float32)
        B = xa.shape[0]
        dev = self.device
        sot_sequence = tok.sot_sequence_including_notimestamps if without_timestamps else tok.sot_sequence
        initial = list(sot_sequence) if initial_tokens is None else [int(t) for t in initial_tokens]
        sample_begin = len(initial)
        sot_index = initial.index(tok.sot)
        sample_len = sample_len or d.n_text_ctx // 2
        cap = d.n_text_ctx + 2
        tokens = torch.full((B, cap), tok.eot, dtype=torch.int32, device=dev)
        tokens[:, :sample_begin] = torch.tensor(initial, dtype=torch.int32, device=dev)
        sum_logprobs = torch.zeros(B, dtype=torch.float32, device=dev)
        smask = None
        if suppress_tokens is not None:
            m = np.zeros(d.n_vocab, np.float32)
            m[list(suppress_tokens)] = -np.inf
            smask = torch.from_numpy(m).to(dev)
        blank = None
        if suppress_blank:
            blank = torch.tensor(list(tok.blank_ids) + [tok.eot], dtype=torch.int32, device=dev)
        ts_rules = not without_timestamps
        max_idx = -1
        if ts_rules and max_initial_timestamp:
            max_idx = round(max_initial_timestamp / (30.0 / d.n_audio_ctx))
        forced = None if forced_tokens is None else forced_tokens.to(dev, torch.int32).t().contiguous()  # [steps, B]
        st = self.new_state(xa)
        step_ws = ops.whisper_step_workspace(dev, B)   # each row of the decode-rules step spread over 16 workgroups
        trace: List[dict] = []
        no_speech = None
        n = sample_begin
        steps = 0
        for i in range(sample_len):
            if n > d.n_text_ctx:
                break
            if i == 0:
                hid = self.decoder_step(tokens[:, :n], st)
                lg_all = self.logits(hid)
                no_speech = ops.softmax_prob_at(lg_all[:, sot_index, :], tok.no_speech, V=d.n_vocab)
                lg = lg_all[:, n - 1, :]
            else:
                hid = self.decoder_step(tokens[:, n - 1:n], st)
                lg = self.logits(hid)[:, 0, :]
            gumbel = None
            if temperature > 0:
                if lg.stride(0) != lg.shape[1]:
                    lg = lg.contiguous()  # the noise shares the logits' row stride: do not draw it for the whole prompt block
                u = torch.rand(lg.shape, dtype=torch.float32, device=dev, generator=generator).clamp_(1e-20, 1.0 - 1e-7)
                gumbel = u.log_().neg_().log_().neg_()
            filt = torch.empty((B, lg.stride(0)), dtype=torch.float32, device=dev) if record else None  # same row stride as lg
            ops.whisper_greedy_step(lg, tokens, n, sample_begin, sum_logprobs, V=d.n_vocab, suppress_mask=smask, blank_ids=blank,
                                    timestamp_rules=ts_rules, timestamp_begin=tok.timestamp_begin, eot=tok.eot,
                                    no_timestamps=-1 if tok.no_timestamps is None else tok.no_timestamps,
                                    max_initial_timestamp_index=max_idx, filtered=filt, gumbel=gumbel, temperature=float(temperature), split_ws=step_ws,
                                    forced_next=None if forced is None else forced[i])
            if record:
                trace.append(dict(raw=lg[:, :d.n_vocab].clone(), filtered=filt[:, :d.n_vocab]))
            n += 1
            steps += 1
            if forced is None and not fixed_steps and (steps % poll == 0):
                if bool((tokens[:, n - 1] == tok.eot).all()):  # the only host round trip of the loop
                    break
        toks = tokens[:, :n].to(torch.long)
        if forced is None and not fixed_steps:
            # the reference stops right after the first step at which every sequence has emitted EOT (decoding.py:626)
            t = toks.cpu()
            all_eot = (t[:, sample_begin:] == tok.eot).all(dim=0)
            if bool(all_eot.any()):
                first = int(torch.nonzero(all_eot)[0]) + sample_begin
                toks = toks[:, :first + 1]
        return dict(tokens=toks, sum_logprobs=sum_logprobs, no_speech_probs=no_speech, trace=trace, sample_begin=sample_begin,
                    audio_features=xa, steps=steps)
