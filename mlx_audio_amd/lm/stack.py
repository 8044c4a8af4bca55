"""Decoder-only transformer stack on MI355X: the host-side schedule over the HIP kernels, shared by every stack on the hot path.

One parametrised engine replaces four separately written reference stacks (file:line relative to /root/reference/mlx_audio):
  * Qwen3-TTS talker and code predictor   tts/models/qwen3_tts/talker.py:230-400, 503-690
  * Qwen3-TTS codec transformer           tts/models/qwen3_tts/speech_tokenizer.py:150-420
  * Mimi transformer                      codec/models/mimi/modules/transformer.py:60-200
  * CSM Llama backbone / depth decoder    lm/models/llama.py:46-198, tts/models/sesame/attention.py:11-175

Per layer (prefill, L > 1 rows per sequence):      norm -> [q | kv] GEMMs (kv lands in its KV-cache slot) -> per-head RMSNorm + RoPE in place
  -> flash attention over the cache -> o-proj GEMM with LayerScale + residual in the epilogue -> norm -> gate|up GEMM -> SwiGLU
  -> down GEMM with LayerScale + residual.   Decode step (1 row per sequence, <= 64 sequences): every GEMM becomes the weight-streaming GEMV on
  the row-major bf16 image (1..8 rows: FMA / MFMA GEMV kernels, 9..64 rows: the matrix-pipe kernel of gemm_rows.hip), SwiGLU is fused into the gate|up GEMV, attention is the KV-streaming kernel.

``KVCache`` mirrors lm/models/cache.py:104-176 (capacity grows in steps of 256, in-place slice update, ``offset``, ``trim``) with one
[B, capacity, 2 * n_kv * dh] fp32 buffer per layer (k | v side by side: one GEMM writes both).
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib, ops
from ..ops import ACT_GELU, ACT_GELU_TANH, ACT_NONE, PackedConv, RowMajor16


@dataclass
class StackConfig:
    d_model: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    d_ff: int
    norm: str = "rms"            # "rms" | "layer"
    norm_eps: float = 1e-6
    qk_norm: bool = False
    rope_theta: Optional[float] = 10000.0
    rope_interleaved: bool = False
    rope_llama3_factor: Optional[float] = None
    max_pos: int = 4096
    attn_bias: bool = False
    mlp: str = "swiglu"          # "swiglu" | "gelu" | "gelu_tanh"
    mlp_bias: bool = False
    layer_scale: bool = False
    causal: bool = True
    window: int = 0
    final_norm: bool = True


def rope_tables(cfg: StackConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host cos / sin tables [max_pos, head_dim / 2] float32: inv_freq = 1 / theta^(2i/d) (talker.py:85, speech_tokenizer.py:198,
    nn.RoPE base), optionally Llama-3 scaled (sesame/attention.py:53-66), angle = pos * inv_freq in float32, then cos / sin."""
    d = cfg.head_dim
    freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    if cfg.rope_llama3_factor is not None:
        low_f, high_f, old_ctx = 1, 4, 8192
        wavelen = 2.0 * math.pi / freqs
        smooth = torch.clip((old_ctx / wavelen - low_f) / (high_f - low_f), 0.0, 1.0)
        scaled = freqs / cfg.rope_llama3_factor
        blended = (1.0 - smooth) * scaled + smooth * freqs
        freqs = torch.where(wavelen < old_ctx / high_f, freqs, torch.where(wavelen > old_ctx / low_f, scaled, blended))
    ang = torch.arange(cfg.max_pos, dtype=torch.float32)[:, None] * freqs[None, :]
    return torch.cos(ang), torch.sin(ang)


class KVCache:
    """lm/models/cache.py:104-176 on the device; ``kv`` is [B, capacity, 2 * n_kv * dh] (k columns first, then v)."""
    step = 256

    def __init__(self, n_kv_heads: int, head_dim: int, device, dtype: torch.dtype = torch.float32):
        self.width = 2 * n_kv_heads * head_dim
        self.n_kv_heads, self.head_dim = n_kv_heads, head_dim
        self.device = device
        self.dtype = dtype   # float32, or the checkpoint's 16-bit type like the reference's caches (lm/models/cache.py:104-176)
        self.kv: Optional[torch.Tensor] = None
        self.offset = 0

    def reserve(self, batch: int, n_new: int) -> torch.Tensor:
        """Makes room for ``n_new`` more rows and returns the slot view [B, n_new, width] they must be written to."""
        prev = self.offset
        if self.kv is None or prev + n_new > self.kv.shape[1]:
            n_steps = (self.step + n_new - 1) // self.step
            new = torch.zeros((batch, n_steps * self.step, self.width), dtype=self.dtype, device=self.device)
            if self.kv is not None:
                old = self.kv[:, :prev] if prev % self.step != 0 else self.kv
                self.kv = torch.cat([old, new], dim=1)
            else:
                self.kv = new
        self.offset = prev + n_new
        return self.kv[:, prev:self.offset, :]

    @property
    def keys(self):
        return None if self.kv is None else self.kv[:, :self.offset, : self.width // 2]

    @property
    def values(self):
        return None if self.kv is None else self.kv[:, :self.offset, self.width // 2:]

    def size(self) -> int:
        return self.offset

    def is_trimmable(self) -> bool:
        return True

    def trim(self, n: int) -> int:
        n = min(self.offset, n)
        self.offset -= n
        return n

    def reset(self):
        self.offset = 0

    def empty(self) -> bool:
        return self.kv is None

    @property
    def nbytes(self) -> int:
        return 0 if self.kv is None else self.kv.numel() * self.kv.element_size()


@dataclass
class Lin:
    pc: PackedConv
    rm: RowMajor16
    tl: Optional["ops.Tiles16"] = None   # tile image for steps of 9..64 sequences (rows_pipe.hip), built on first use

    def tiles(self) -> "ops.Tiles16":
        if self.tl is None:
            self.tl = ops.tiles16_from_rowmajor(self.rm)
        return self.tl


def make_lin(w: torch.Tensor, bias: Optional[torch.Tensor], device, f16: bool = False, fp8: bool = False, gemv_only: bool = False) -> Lin:
    """MFMA image (prefill) + row-major GEMV image (decode steps) of one Linear.  ``fp8``: the GEMV image is OCP e4m3fn with per-row
    power-of-two scales (half the bytes of a decode step) and the MFMA image is built from the SAME dequantised values, which bf16 holds
    exactly -- prefill and decode see one set of weights."""
    if fp8:
        rm, wq = ops.pack_rowmajor_fp8(w, bias, device)
        return Lin(None if gemv_only else ops.pack_conv(wq, bias, device), rm)
    return Lin(None if gemv_only else ops.pack_conv(w, bias, device, f16=f16), ops.pack_rowmajor16(w, bias, device, f16=f16))


def effective_weights(weights: Dict[str, torch.Tensor], cfg: "StackConfig", weight_format: str = "bf16", prefix: str = "") -> Dict[str, torch.Tensor]:
    """The float32 weights a ``TransformerStack`` built with ``weight_format`` actually computes with (canonical names, prefix stripped):
    bf16-rounded, and for "fp8" every Linear replaced by its dequantised fp8 image (rows quantised exactly as the engine quantises them: q | k | v
    and gate | up rows are quantised per row, so fusing them into one image changes nothing).  Parity tests hand these to the oracle."""
    w = {k[len(prefix):]: v.detach().to(torch.bfloat16).to(torch.float32).cpu() for k, v in weights.items() if k.startswith(prefix)}
    if weight_format == "fp8":
        for k in list(w):
            if k.endswith(".weight") and w[k].dim() == 2 and k.split(".")[-2] in ("wq", "wk", "wv", "wo", "w_gate", "w_up", "w_down", "w1", "w2"):
                w[k] = ops.dequantize_rows_fp8(*ops.quantize_rows_fp8(w[k]))
    return w


MAX_DECODE_ROWS = 64   # mi355_gemv: 1..8 rows on the GEMV kernels, 9..64 rows on the matrix-pipe kernel of gemm_rows.hip


def decode_rows(l: "Lin") -> int:
    """Most sequences a single-position step through ``l`` may carry on the weight-streaming path: 64 when K % 64 == 0 (tile images), else 8."""
    return MAX_DECODE_ROWS if l.rm.k % 64 == 0 else 8


def is_decode(x: torch.Tensor, l: Optional["Lin"] = None) -> bool:
    """1 row per sequence and few enough sequences for the weight-streaming GEMV path (``decode_rows``; 8 when no image is named)."""
    return x.shape[1] == 1 and x.shape[0] <= (8 if l is None else decode_rows(l))


ROWS_PIPE = True   # single-position Linear layers of ROWS_MIN..64 sequences outside the native runner: rows pipeline (False: mi355_gemv's kernels)
ROWS_MIN = int(os.environ.get("MI355_ROWS_MIN", "5"))   # fewest sequences per step the rows pipeline takes (5..8 were the matrix-pipe GEMV's; 9 = the old split)
_ROWS_WS: Dict[tuple, torch.Tensor] = {}


def _rows_ws(device, kind: str, nbytes: int) -> torch.Tensor:
    """Grow-only scratch of the rows pipeline per (device, stream, kind): launches on one stream are ordered, so one buffer per kind serves all calls."""
    key = (str(device), ops._stream(), kind)
    cur = _ROWS_WS.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ROWS_WS[key] = cur
    return cur


def linear_rows(x: torch.Tensor, l: Lin, y: torch.Tensor, *, post_act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
                colscale: Optional[torch.Tensor] = None, glu: bool = False, norm: Optional[tuple] = None, y2: Optional[torch.Tensor] = None):
    """``linear`` for 9..64 single-position rows through the rows pipeline: converter (fused input norm) -> mi355_rows_gemm -> row epilogue."""
    B = x.shape[0]
    tl = l.tiles()
    R = ops.rows_R(B)
    f16 = l.rm.f16
    planes = _rows_ws(x.device, "planes", 4 * R * tl.k)
    kg = ops.rows_kgroups(tl.n, tl.k)
    ld = ops.round_up(tl.n, 8)
    part = _rows_ws(x.device, "part", 4 * kg * B * ld).view(torch.float32)[: kg * B * ld].view(kg, B, ld)
    ops.rows_finish(x[:, 0, :], B, tl.k, norm=norm, planes=planes, R=R, f16=f16 and tl.scale is None)
    ops.rows_gemm(planes, tl, part, B, R, kgroups=kg)
    ops.rows_finish(part, B, tl.n, kg, bias=l.rm.bias, post_act=post_act, res=None if res is None else res[:, 0, :], colscale=colscale, glu=glu,
                    wscale=tl.scale, y=y[:, 0, :], y2=None if y2 is None else y2[:, 0, :])
    return y


def linear(x: torch.Tensor, l: Lin, y: torch.Tensor, *, post_act: int = ACT_NONE, res: Optional[torch.Tensor] = None,
           colscale: Optional[torch.Tensor] = None, glu: bool = False, precision: int = 2, norm: Optional[tuple] = None,
           y2: Optional[torch.Tensor] = None, w_policy: int = 0):
    """y = act(norm(x) W^T + b) * colscale + res on [B, L, C] views; 1-row-per-sequence inputs with B <= 8 take the GEMV
    (``norm`` / ``y2`` -- fused input normalisation and split destination -- exist on that path only)."""
    if is_decode(x, l) and x.shape[0] >= min(ROWS_MIN, 9) and ROWS_PIPE and l.rm.k % 64 == 0 and not (glu and (l.rm.n % 16)):
        return linear_rows(x, l, y, post_act=post_act, res=res, colscale=colscale, glu=glu, norm=norm, y2=y2)
    if is_decode(x, l):
        ops.gemv(x[:, 0, :], l.rm, y[:, 0, :], post_act=post_act, res=None if res is None else res[:, 0, :], colscale=colscale, glu=glu,
                 norm=norm, y2=None if y2 is None else y2[:, 0, :], w_policy=w_policy)
    else:
        assert not glu and norm is None and y2 is None
        ops.conv_gemm(x, l.pc, y, post_act=post_act, res=res, colscale=colscale, precision=precision)
    return y


@dataclass
class _Layer:
    attn_norm: Tuple[torch.Tensor, Optional[torch.Tensor]]
    wq: Lin
    wkv: Lin
    wqkv: Lin            # decode step: one GEMV writes q to its buffer and k | v into the KV-cache slot
    wo: Lin
    q_norm: Optional[torch.Tensor]
    k_norm: Optional[torch.Tensor]
    mlp_norm: Tuple[torch.Tensor, Optional[torch.Tensor]]
    w_in: Lin            # gate|up interleaved (SwiGLU) or w1
    w_out: Lin           # down or w2
    ls1: Optional[torch.Tensor]
    ls2: Optional[torch.Tensor]


class TransformerStack:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: StackConfig, device="cuda:0", precision: int = 2, prefix: str = "",
                 weight_format: str = "bf16", kv_dtype: torch.dtype = torch.float32):
        ops.require_gpu()
        assert kv_dtype in ops.KV_DTYPES
        self.kv_dtype = kv_dtype  # element type of the KV caches make_cache() builds (bfloat16 = the reference's cache dtype for bf16 checkpoints)
        assert cfg.head_dim in (64, 128) and cfg.d_model % 4 == 0
        assert weight_format in ("bf16", "fp8"), weight_format
        fp8 = weight_format == "fp8"
        assert not fp8 or (cfg.d_model % 16 == 0 and cfg.d_ff % 16 == 0 and (cfg.n_heads * cfg.head_dim) % 16 == 0)
        self.weight_format = weight_format
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        self.native_decode = True  # single-position steps go through mi355_stack_decode_step (False: the per-op Python schedule)
        self.rows_pipe = True      # native steps of 9..64 sequences run the rows pipeline (rows_pipe.hip); False: mi355_gemv's 9..64-row kernel
        # sequences per single-position step on the weight-streaming path: 9..64 need 16-bit images and widths that are multiples of 64
        self.max_decode_rows = MAX_DECODE_ROWS if (cfg.d_model % 64 == 0 and cfg.d_ff % 64 == 0 and (cfg.n_heads * cfg.head_dim) % 64 == 0) else 8
        dev = self.device
        w = {k[len(prefix):]: v.detach().to(torch.bfloat16).to(torch.float32).cpu() for k, v in weights.items() if k.startswith(prefix)}

        def vec(name):
            t = w.get(name)
            return None if t is None else t.to(dev)

        def lin(name, extra_rows=None):
            return make_lin(w[name + ".weight"], w.get(name + ".bias"), dev, fp8=fp8)

        self.layers: List[_Layer] = []
        for i in range(cfg.n_layers):
            p = f"layers.{i}."
            wk, wv = w[p + "wk.weight"], w[p + "wv.weight"]
            bk, bv = w.get(p + "wk.bias"), w.get(p + "wv.bias")
            kvb = None if bk is None and bv is None else torch.cat([bk if bk is not None else torch.zeros(wk.shape[0]),
                                                                     bv if bv is not None else torch.zeros(wv.shape[0])])
            if cfg.mlp == "swiglu":
                wg, wu = w[p + "w_gate.weight"], w[p + "w_up.weight"]
                w_in_w = torch.stack([wg, wu], dim=1).reshape(2 * wg.shape[0], wg.shape[1])  # gate_0, up_0, gate_1, up_1, ...
                bg, bu = w.get(p + "w_gate.bias"), w.get(p + "w_up.bias")
                w_in_b = None if bg is None else torch.stack([bg, bu], dim=1).reshape(-1)
                w_in = make_lin(w_in_w, w_in_b, dev, fp8=fp8)
                w_out = lin(p + "w_down")
            else:
                w_in, w_out = lin(p + "w1"), lin(p + "w2")
            wq_w, bq = w[p + "wq.weight"], w.get(p + "wq.bias")
            qkvb = None if kvb is None and bq is None else torch.cat([bq if bq is not None else torch.zeros(wq_w.shape[0]),
                                                                       kvb if kvb is not None else torch.zeros(wk.shape[0] + wv.shape[0])])
            wqkv = make_lin(torch.cat([wq_w, wk, wv]), qkvb, dev, fp8=fp8, gemv_only=True)
            self.layers.append(_Layer((vec(p + "attn_norm.weight"), vec(p + "attn_norm.bias")), lin(p + "wq"),
                                      make_lin(torch.cat([wk, wv]), kvb, dev, fp8=fp8), wqkv, lin(p + "wo"), vec(p + "q_norm.weight"), vec(p + "k_norm.weight"),
                                      (vec(p + "mlp_norm.weight"), vec(p + "mlp_norm.bias")), w_in, w_out, vec(p + "ls1"), vec(p + "ls2")))
        self.final_norm = (vec("final_norm.weight"), vec("final_norm.bias")) if cfg.final_norm else None
        if cfg.rope_theta is not None:
            cos, sin = rope_tables(cfg)
            self.cos, self.sin = cos.contiguous().to(dev), sin.contiguous().to(dev)
        else:
            self.cos = self.sin = None

    def make_cache(self) -> List[KVCache]:
        return [KVCache(self.cfg.n_kv_heads, self.cfg.head_dim, self.device, self.kv_dtype) for _ in range(self.cfg.n_layers)]

    # ------------------------------------------------------------------ native decode step (mi355_stack_decode_step)
    def _native_desc(self, cache: List[KVCache], k_start: Optional[torch.Tensor] = None, tall: bool = False, slot_lens_k: Optional[torch.Tensor] = None):
        """Builds (or refreshes after a cache re-allocation) the C descriptor of this stack for the given caches.  The key carries everything the
        descriptor stores about a cache: the caching allocator may hand a later, differently sized cache the same addresses.  ``tall``: the step
        carries 9..64 sequences -- the descriptor then also names the tile images and the rows workspace (built on first use, kept)."""
        tall = self.rows_pipe and (tall or getattr(self, "_rows_ws", None) is not None)
        key = tuple((c.kv.data_ptr(), c.kv.shape[0], c.kv.shape[1], c.kv.stride(0)) for c in cache) + (None if k_start is None else k_start.data_ptr(), tall,
                                                                                                           None if slot_lens_k is None else slot_lens_k.data_ptr())
        st = getattr(self, "_native", None)
        if st is not None and st["key"] == key:
            return st
        c = self.cfg
        LD, SD = _lib.STRUCTS["mi355_layer_desc"], _lib.STRUCTS["mi355_stack_desc"]
        arr = (LD * c.n_layers)()
        p = ops._ptr
        for i, (lyr, kvc) in enumerate(zip(self.layers, cache)):
            a = arr[i]
            a.wqkv, a.bqkv = p(lyr.wqkv.rm.w), p(lyr.wqkv.rm.bias)
            a.wo, a.bo = p(lyr.wo.rm.w), p(lyr.wo.rm.bias)
            a.w_in, a.b_in = p(lyr.w_in.rm.w), p(lyr.w_in.rm.bias)
            a.w_out, a.b_out = p(lyr.w_out.rm.w), p(lyr.w_out.rm.bias)
            a.attn_norm_w, a.attn_norm_b = p(lyr.attn_norm[0]), p(lyr.attn_norm[1])
            a.mlp_norm_w, a.mlp_norm_b = p(lyr.mlp_norm[0]), p(lyr.mlp_norm[1])
            a.q_norm, a.k_norm, a.ls1, a.ls2 = p(lyr.q_norm), p(lyr.k_norm), p(lyr.ls1), p(lyr.ls2)
            a.kv, a.kv_bstride, a.kv_capacity = kvc.kv.data_ptr(), kvc.kv.stride(0), kvc.kv.shape[1]
            a.s_qkv, a.s_o, a.s_in, a.s_out = p(lyr.wqkv.rm.scale), p(lyr.wo.rm.scale), p(lyr.w_in.rm.scale), p(lyr.w_out.rm.scale)
            if tall:
                a.wqkv_t, a.wo_t, a.w_in_t, a.w_out_t = p(lyr.wqkv.tiles().w), p(lyr.wo.tiles().w), p(lyr.w_in.tiles().w), p(lyr.w_out.tiles().w)
        d = SD()
        d.n_layers, d.d_model, d.heads, d.kv_heads, d.dh, d.d_ff = c.n_layers, c.d_model, c.n_heads, c.n_kv_heads, c.head_dim, c.d_ff
        d.norm, d.eps, d.glu = (1 if c.norm == "layer" else 2), c.norm_eps, int(c.mlp == "swiglu")
        d.act = {"gelu": ACT_GELU, "gelu_tanh": ACT_GELU_TANH}.get(c.mlp, ACT_NONE)
        d.wdtype, d.causal, d.window, d.attn_scale = self.layers[0].wqkv.rm.wdtype, int(c.causal), c.window, 0.0
        d.rope_mode, d.cos, d.sin = int(c.rope_interleaved), p(self.cos), p(self.sin)
        d.rope_rows = 0 if self.cos is None else self.cos.shape[0]
        d.k_start = p(k_start)
        d.slot_lens_k = p(slot_lens_k)
        d.kv_dtype = ops.KV_DTYPES[cache[0].kv.dtype]
        d.w_policy = int(getattr(self, "w_policy", 0))   # 1: the stack's weight images stream past the caches (non-temporal loads) in one-sequence steps
        d.layers = ctypes.cast(arr, ctypes.c_void_p)
        sws, scnt = ops.attn_split_workspace(self.device, 8 * c.n_heads, c.head_dim)  # key-split decode attention (long key ranges)
        d.attn_split_ws, d.attn_split_cnt = sws.data_ptr(), scnt.data_ptr()
        gws, gcnt = ops.gemv_split_workspace(self.device, c.d_model, c.d_ff)   # K split over workgroups for the down projection (5..8 rows)
        d.gemv_split_ws, d.gemv_split_cnt = gws.data_ptr(), gcnt.data_ptr()
        if self.final_norm is not None:
            d.final_norm_w, d.final_norm_b = p(self.final_norm[0]), p(self.final_norm[1])
        if tall:
            if getattr(self, "_rows_ws", None) is None:
                need = int(_lib.load().mi355_stack_rows_ws_bytes(ctypes.byref(d), MAX_DECODE_ROWS))
                assert need > 0
                self._rows_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            d.rows_ws, d.rows_ws_bytes = self._rows_ws.data_ptr(), self._rows_ws.numel()
        self._native = dict(key=key, arr=arr, desc=d, k_start=k_start, slot_lens_k=slot_lens_k)
        return self._native

    def _check_positions(self, offset: int, n_new: int):
        """RoPE tables have ``max_pos`` rows (CSM: 2048, talker: <= 8192): running past them must raise, not read garbage (sesame.py:817-820)."""
        if self.cos is not None and offset + n_new > self.cos.shape[0]:
            raise ValueError(f"sequence position {offset + n_new - 1} is past the {self.cos.shape[0]}-row rotary tables (max_pos) of this stack")

    def decode_step(self, x: torch.Tensor, cache: List[KVCache], k_start: Optional[torch.Tensor] = None, defer_final_norm: bool = False,
                    slot_lens_k: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One single-position step for B <= ``max_decode_rows`` sequences through the native runner: x [B, 1, d_model] (updated in place); returns the
        final-normed hidden state [B, 1, d_model] (or x itself when the stack has no final norm).  ``k_start`` int32 [B]: left padding.
        ``defer_final_norm``: return the UN-normalised residual stream; the caller fuses the final norm into the GEMV that consumes it
        (``final_norm_arg()`` is the ``norm=`` tuple for ``linear``): one launch less per step."""
        c = self.cfg
        B = x.shape[0]
        off = cache[0].offset
        self._check_positions(off, 1)
        for kvc in cache:
            kvc.reserve(B, 1)
        if slot_lens_k is not None:   # slot caches (continuous batching): int32 [B] device, item b appends position slot_lens_k[b] - 1 to its own row
            assert slot_lens_k.dtype == torch.int32 and slot_lens_k.is_cuda and slot_lens_k.numel() >= B and k_start is None
        st = self._native_desc(cache, k_start, tall=B >= min(ROWS_MIN, 9) and self.rows_pipe and self.max_decode_rows > 8, slot_lens_k=slot_lens_k)
        ws = torch.empty(B * (2 * c.n_heads * c.head_dim + c.d_ff + 2 * c.n_kv_heads * c.head_dim), dtype=torch.float32, device=self.device)
        out = torch.empty_like(x) if (self.final_norm is not None and not defer_final_norm) else None
        lib = _lib.load()
        rc = lib.mi355_stack_decode_step(ctypes.byref(st["desc"]), x.data_ptr(), B, off, ws.data_ptr(), None if out is None else out.data_ptr(),
                                         ops._stream())
        _lib.check(rc, "mi355_stack_decode_step")
        return out if out is not None else x

    def final_norm_arg(self) -> Optional[tuple]:
        """``norm=`` argument of ``linear`` / ``ops.gemv`` that applies this stack's final norm inside the consuming GEMV."""
        if self.final_norm is None:
            return None
        return ("layer" if self.cfg.norm == "layer" else "rms", self.final_norm[0], self.final_norm[1], self.cfg.norm_eps)

    def _norm(self, x: torch.Tensor, p) -> torch.Tensor:
        y = torch.empty_like(x)
        if self.cfg.norm == "layer":
            return ops.layernorm(x, y, weight=p[0], bias=p[1], eps=self.cfg.norm_eps)
        return ops.rmsnorm(x, y, p[0], eps=self.cfg.norm_eps)

    def __call__(self, x: torch.Tensor, cache: Optional[List[KVCache]] = None, return_layers: bool = False, k_start: Optional[torch.Tensor] = None,
                 defer_final_norm: bool = False):
        """x [B, L, d_model] fp32 on the device (modified in place and returned, normalised when ``final_norm``).
        ``k_start`` int32 [B] (device): row b of the batch is LEFT-padded by k_start[b] positions (BatchKVCache, lm/models/cache.py:502-560;
        the reference's batched talker builds the same thing from its attention mask, talker.py:443-470): its keys before k_start[b] are
        invisible and its rotary positions count from k_start[b].  Outputs at padding positions are meaningless."""
        c = self.cfg
        B, L, D = x.shape
        assert D == c.d_model and x.is_contiguous()
        if cache is None:
            cache = self.make_cache()
        if k_start is not None:
            assert k_start.dtype == torch.int32 and k_start.shape == (B,) and k_start.is_cuda
        decode = L == 1 and B <= self.max_decode_rows
        if decode and not return_layers and self.native_decode:
            return self.decode_step(x, cache, k_start, defer_final_norm)
        assert not defer_final_norm, "defer_final_norm exists on the single-position decode path only"
        self._check_positions(cache[0].offset, L)
        H, G, dh = c.n_heads, c.n_kv_heads, c.head_dim
        dev = self.device
        q = torch.empty((B, L, H * dh), dtype=torch.float32, device=dev)
        att = torch.empty((B, L, H * dh), dtype=torch.float32, device=dev)
        ff_w = c.d_ff
        mid = torch.empty((B, L, ff_w), dtype=torch.float32, device=dev)
        gu = None
        if c.mlp == "swiglu" and not decode:
            gu = torch.empty((B, L, 2 * c.d_ff), dtype=torch.float32, device=dev)
        act = {"gelu": ACT_GELU, "gelu_tanh": ACT_GELU_TANH}.get(c.mlp, ACT_NONE)
        layers = []
        nmode = "layer" if c.norm == "layer" else "rms"
        for lyr, kvc in zip(self.layers, cache):
            off = kvc.offset
            cache_slot = kvc.reserve(B, L)
            # a 16-bit cache receives its rows through a float32 scratch block: projection and norm / rope run in float32, ONE rounding into the cache
            slot = cache_slot if cache_slot.dtype == torch.float32 else torch.empty(cache_slot.shape, dtype=torch.float32, device=dev)
            if decode:  # pre-norm fused into the one q | k | v GEMV
                linear(x, lyr.wqkv, q, norm=(nmode, lyr.attn_norm[0], lyr.attn_norm[1], c.norm_eps), y2=slot)
            else:
                h = self._norm(x, lyr.attn_norm)
                linear(h, lyr.wq, q, precision=self.precision)
                linear(h, lyr.wkv, slot, precision=self.precision)
            if lyr.q_norm is not None or self.cos is not None:
                ks = slot[:, :, : G * dh]
                ops.head_norm_rope(q, q, heads=H, dh=dh, norm_weight=lyr.q_norm, eps=c.norm_eps, cos=self.cos, sin=self.sin, pos0=off,
                                   interleaved=c.rope_interleaved, second=(ks, ks, G, lyr.k_norm), pos_sub=k_start)  # q and k heads in one launch
            if slot is not cache_slot:
                cache_slot.copy_(slot)
            ops.flash_attention(q, kvc.keys, kvc.values, att, heads=H, kv_heads=G, dh=dh, causal=c.causal, window=c.window, k_start=k_start)
            linear(att, lyr.wo, x, res=x, colscale=lyr.ls1, precision=self.precision)
            if decode:  # pre-norm (+ SwiGLU) fused into the up-projection GEMV
                linear(x, lyr.w_in, mid, glu=c.mlp == "swiglu", post_act=ACT_NONE if c.mlp == "swiglu" else act,
                       norm=(nmode, lyr.mlp_norm[0], lyr.mlp_norm[1], c.norm_eps))
            else:
                h = self._norm(x, lyr.mlp_norm)
                if c.mlp == "swiglu":
                    linear(h, lyr.w_in, gu, precision=self.precision)
                    ops.swiglu(gu, mid)
                else:
                    linear(h, lyr.w_in, mid, post_act=act, precision=self.precision)
            linear(mid, lyr.w_out, x, res=x, colscale=lyr.ls2, precision=self.precision)
            if return_layers:
                layers.append(x.clone())
        out = self._norm(x, self.final_norm) if self.final_norm is not None else x
        return (out, layers) if return_layers else out
