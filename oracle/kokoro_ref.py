"""PyTorch-CPU restatement of Kokoro-82M's forward pass (TEST ORACLE, not product).

Follows, block by block:
  * ``tts/models/kokoro/kokoro.py:111-177``   Model.__call__ (tokens -> waveform)
  * ``tts/models/kokoro/modules.py``          CustomAlbert :434-655, TextEncoder :21-68,
                                              AdaLayerNorm :71-90, LSTM :93-285,
                                              ProsodyPredictor :288-377, DurationEncoder :380-411
  * ``tts/models/kokoro/istftnet.py``         weight_norm/ConvWeighted :53-170, AdaIN1d :326-338,
                                              AdaINResBlock1 :341-396, MLXSTFT :453-545,
                                              SineGen/SourceModuleHnNSF :548-709, Generator :725-835,
                                              AdainResBlk1d :853-933, Decoder :936-997
  * ``tts/models/interpolate.py``             (via oracle.interp_ref)
  * ``dsp.py``                                (via oracle.dsp_ref)

Layout: the reference keeps activations NCL between blocks and swaps to NLC around
every MLX conv; here everything is NCL and ``torch.nn.functional`` convs are used
with the MLX weight layout ``(C_out, K, C_in/groups)`` permuted on the fly.

Precision model: parameters are bf16-representable
values (the checkpoint is bf16 and ``weight_norm`` runs in the parameter dtype),
activations are float32 (the reference promotes to fp32 as soon as an fp32
operand -- voice pack, ``mx.zeros`` state, window -- enters).  ``dtype=torch.float64``
gives a higher-precision "truth" used to measure both this oracle's and the HIP
path's rounding error.

Stochastic inputs (SineGen's uniform initial phase and gaussian noise,
istftnet.py:581,649) are explicit arguments so results are reproducible.

End-to-end parity status: **pinned to the reference's own modules** (round 2): the reference ships no golden Kokoro
output, so ``tests/golden/make_reference_fixtures.py`` imports the reference's Kokoro source files from where they lie and
runs them, unmodified, on a seeded synthetic checkpoint over a numpy stand-in for MLX (``tests/golden/mlx_shim.py``: MLX itself
is not installable here); ``tests/test_reference_fixtures_cpu.py`` requires this oracle to reproduce those fixtures (durations
exact, harmonic source bit-exact, intermediates 1e-6..5e-6, waveform 3.2e-5 max-abs / 106 dB with the reference's F0 / N
injected).  What that cannot cover is MLX's own kernels (the stand-in follows their documented semantics).  Pinned pieces from
the reference's known-answer tests: weight-normed transposed conv, MLXSTFT round trip, interpolate, stft/istft/mel
(tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import dsp_ref, interp_ref

Tensor = torch.Tensor


# ------------------------------------------------------------------ weight norm / convs
def weight_norm_param_dtype(v: Tensor, g: Tensor, param_dtype=torch.bfloat16) -> Tensor:
    """istftnet.py:53-93 with dim=0, evaluated in the *parameter* dtype like MLX does.

    Every op (v*v, sum, sqrt, +1e-7, /, *g) rounds to ``param_dtype``; the result
    is returned as float32 holding param_dtype-representable values.
    """
    v_ = v.to(param_dtype)
    g_ = g.to(param_dtype)
    axes = tuple(range(1, v_.dim()))
    sq = v_ * v_
    nrm = torch.sqrt(sq.sum(dim=axes, keepdim=True))
    eps = torch.tensor(1e-7, dtype=param_dtype)
    w = (v_ / (nrm + eps)) * g_
    return w.to(torch.float32)


def conv1d_mlx(x: Tensor, w: Tensor, b: Optional[Tensor], stride=1, padding=0, dilation=1, groups=1):
    """``mx.conv1d`` on NCL ``x`` with MLX weight ``(C_out, K, C_in/groups)``."""
    return F.conv1d(x, w.permute(0, 2, 1).to(x.dtype), None if b is None else b.to(x.dtype),
                    stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose1d_mlx(x: Tensor, w: Tensor, b: Optional[Tensor], stride=1, padding=0, groups=1):
    """``mx.conv_transpose1d`` on NCL ``x``; MLX weight ``(C_out, K, C_in/groups)``.

    out[co, t*stride + k - padding] += x[ci, t] * w[co, k, ci]  (the scatter form the
    reference pins in tts/tests/test_istftnet_fidelity.py:18-31).
    """
    if groups == 1:
        wt = w.permute(2, 0, 1)  # torch wants (C_in, C_out, K)
    else:  # depthwise: (C, K, 1) -> (C_in=C, C_out/groups=1, K)
        wt = w.permute(0, 2, 1)
    y = F.conv_transpose1d(x, wt.to(x.dtype), None, stride=stride, padding=padding, groups=groups)
    if b is not None:
        y = y + b.to(x.dtype).view(1, -1, 1)
    return y


FQ_MARGINS: Optional[list] = None  # a test may bind a list here: every fake-quant call then records its smallest rounding margin
FQ_JITTER: Optional[tuple] = None  # (relative amplitude, torch.Generator): tests perturb every quantiser input by that much gaussian noise to
#                                    measure how far ordinary rounding noise of that size moves the network's outputs (grid steps flip)


def fake_quant_dynamic_u8(x: Tensor) -> Tensor:
    """Per-tensor dynamic uint8 fake quantisation (kitten_tts/quant.py:4-20), float32 like the reference (``x.astype(mx.float32)``)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    if FQ_JITTER is not None:
        amp, gen = FQ_JITTER
        xf = xf * (1.0 + amp * torch.randn(xf.shape, generator=gen))
    zero = torch.zeros((), dtype=torch.float32)
    x_min = torch.minimum(xf.min(), zero)
    x_max = torch.maximum(xf.max(), zero)
    scale = (x_max - x_min) / 255.0
    if float(scale) == 0.0:
        return torch.zeros_like(x)
    zp = torch.clamp(torch.round(-x_min / scale), 0.0, 255.0)
    pos = xf / scale + zp
    if FQ_MARGINS is not None:  # distance of the closest element from a rounding boundary, in grid steps (tests: margin rule)
        FQ_MARGINS.append(float(((pos - torch.floor(pos)) - 0.5).abs().min()))
    q = torch.clamp(torch.round(pos), 0.0, 255.0)
    return ((q - zp) * scale).to(dt)


class P:
    """Flat parameter dictionary with prefix navigation (MLX post-``sanitize`` names)."""

    def __init__(self, weights: Dict[str, Tensor], prefix: str = "", dtype=torch.float32,
                 param_dtype=torch.bfloat16, quant_modules=()):
        self.w, self.prefix, self.dtype, self.param_dtype = weights, prefix, dtype, param_dtype
        self.quant_modules = tuple(quant_modules)

    def sub(self, name) -> "P":
        return P(self.w, f"{self.prefix}{name}.", self.dtype, self.param_dtype, self.quant_modules)

    @property
    def quant(self) -> bool:
        """KittenTTS only: does the module at this prefix carry ``activation_quant``?  The reference flags a module when a listed name is
        the module itself or lies below it (kitten_tts.py:291-299); Kokoro never lists any."""
        name = self.prefix[:-1]
        return bool(name) and any(q == name or q.startswith(name + ".") for q in self.quant_modules)

    def fq(self, x: Tensor) -> Tensor:
        return fake_quant_dynamic_u8(x) if self.quant else x

    def has(self, name) -> bool:
        return f"{self.prefix}{name}" in self.w

    def __call__(self, name) -> Tensor:
        return self.w[f"{self.prefix}{name}"].to(torch.float32).to(self.dtype)

    def raw(self, name) -> Tensor:
        return self.w[f"{self.prefix}{name}"]

    def wn(self) -> Tensor:
        """Folded weight of a ConvWeighted at this prefix."""
        return weight_norm_param_dtype(self.raw("weight_v"), self.raw("weight_g"), self.param_dtype).to(self.dtype)

    def bias(self):
        return self("bias") if self.has("bias") else None


def conv_weighted(p: P, x: Tensor, transpose: bool = False, **kw) -> Tensor:
    """ConvWeighted.__call__ (istftnet.py:128-170) on NCL input."""
    w = p.wn()
    x = p.fq(x)  # kitten_tts/istftnet.py:131
    groups = kw.get("groups", 1)
    if transpose:
        if groups == 1:
            # x channels != weight.shape[-1] -> full axis reversal (istftnet.py:161-166)
            w = w.permute(2, 1, 0)
        return conv_transpose1d_mlx(x, w, p.bias(), stride=kw.get("stride", 1), padding=kw.get("padding", 0), groups=groups)
    return conv1d_mlx(x, w, p.bias(), **kw)


def linear(p: P, x: Tensor) -> Tensor:
    y = x @ p("weight").t()
    if p.has("bias"):
        y = y + p("bias")
    return y


def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float) -> Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, keepdim=True, unbiased=False)
    y = (x - mean) * torch.rsqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def leaky_relu(x: Tensor, slope: float) -> Tensor:
    return torch.where(x > 0, x, x * slope)


# ------------------------------------------------------------------ AdaIN blocks
def adain1d(p: P, x: Tensor, s: Tensor) -> Tensor:
    """AdaIN1d (istftnet.py:326-338): (1+gamma)*InstanceNorm(x)+beta, biased var, eps 1e-5."""
    h = linear(p.sub("fc"), p.fq(s)).unsqueeze(2)  # kitten_tts/istftnet.py:336
    gamma, beta = h.chunk(2, dim=1)
    mean = x.mean(dim=2, keepdim=True)
    var = x.var(dim=2, keepdim=True, unbiased=False)
    xn = (x - mean) / torch.sqrt(var + 1e-5)
    return (1 + gamma) * xn + beta


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    return x + (1 / alpha) * (torch.sin(alpha * x) ** 2)


def _alpha(p: P, which: int, i: int) -> Tensor:
    """Snake parameters: a list in Kokoro (``alpha1.0``, istftnet.py:374), per-index attributes in KittenTTS (``alpha1_0``, kitten_tts/istftnet.py:379-384)."""
    return p(f"alpha{which}.{i}") if p.has(f"alpha{which}.{i}") else p(f"alpha{which}_{i}")


def adain_resblock1(p: P, x: Tensor, s: Tensor, kernel: int, dilations=(1, 3, 5)) -> Tensor:
    """AdaINResBlock1 (istftnet.py:341-396)."""
    for i, d in enumerate(dilations):
        xt = adain1d(p.sub(f"adain1.{i}"), x, s)
        xt = snake(xt, _alpha(p, 1, i))
        xt = conv_weighted(p.sub(f"convs1.{i}"), xt, padding=(kernel * d - d) // 2, dilation=d)
        xt = adain1d(p.sub(f"adain2.{i}"), xt, s)
        xt = snake(xt, _alpha(p, 2, i))
        xt = conv_weighted(p.sub(f"convs2.{i}"), xt, padding=(kernel - 1) // 2, dilation=1)
        x = xt + x
    return x


def adain_resblk1d(p: P, x: Tensor, s: Tensor, upsample: bool) -> Tensor:
    """AdainResBlk1d (istftnet.py:853-933)."""
    r = leaky_relu(adain1d(p.sub("norm1"), x, s), 0.2)
    if upsample:
        c = x.shape[1]
        r = conv_weighted(p.sub("pool"), r, transpose=True, stride=2, padding=0, groups=c)[:, :, 1:]
    r = conv_weighted(p.sub("conv1"), r, padding=1)
    r = leaky_relu(adain1d(p.sub("norm2"), r, s), 0.2)
    r = conv_weighted(p.sub("conv2"), r, padding=1)
    sc = x
    if upsample:
        sc = sc.repeat_interleave(2, dim=2)  # nn.Upsample(scale_factor=2, nearest)
    if p.has("conv1x1.weight_v"):
        sc = conv_weighted(p.sub("conv1x1"), sc, padding=0)
    return (r + sc) / math.sqrt(2.0)


# ------------------------------------------------------------------ LSTM / ALBERT / encoders
def bilstm(p: P, x: Tensor) -> Tensor:
    """LSTM (modules.py:93-285): x [B, L, In] -> [B, L, 2H]; gate order i,f,g,o."""
    outs = []
    for direction in ("forward", "backward"):
        wx, wh = p(f"Wx_{direction}"), p(f"Wh_{direction}")
        bias = p(f"bias_ih_{direction}") + p(f"bias_hh_{direction}")
        xp = p.fq(x) @ wx.t() + bias  # kitten_tts/modules.py:155,201
        bsz, L, _ = x.shape
        hdim = wh.shape[1]
        h = x.new_zeros(bsz, hdim)
        c = x.new_zeros(bsz, hdim)
        seq = [None] * L
        order = range(L) if direction == "forward" else range(L - 1, -1, -1)
        for t in order:
            gates = xp[:, t, :] + p.fq(h) @ wh.t()  # kitten_tts/modules.py:178,224 (the emitted h stays unquantised)
            i, f, g, o = gates.chunk(4, dim=-1)
            i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
            c = f * c + i * g
            h = o * torch.tanh(c)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
    return torch.cat(outs, dim=-1)


def albert(p: P, input_ids: Tensor, attn_mask01: Tensor, cfg: dict) -> Tensor:
    """CustomAlbert (modules.py:434-655) -> sequence_output [B, T, hidden]."""
    eps = cfg.get("layer_norm_eps", 1e-12)
    nh = cfg["num_attention_heads"]
    B, T = input_ids.shape
    e = p.sub("embeddings")
    pos = torch.arange(T)
    emb = e("word_embeddings.weight")[input_ids] + e("position_embeddings.weight")[pos][None] \
        + e("token_type_embeddings.weight")[torch.zeros_like(input_ids)]
    h = layer_norm(emb, e("LayerNorm.weight"), e("LayerNorm.bias"), eps)
    add_mask = (1.0 - attn_mask01.to(h.dtype))[:, None, None, :] * -10000.0
    enc = p.sub("encoder")
    h = linear(enc.sub("embedding_hidden_mapping_in"), h)
    lay = enc.sub("albert_layer_groups.0.albert_layers.0")
    att = lay.sub("attention")
    hd = h.shape[-1] // nh
    for _ in range(cfg["num_hidden_layers"]):
        def split(t):
            return t.view(B, T, nh, hd).permute(0, 2, 1, 3)
        q, k, v = split(linear(att.sub("query"), h)), split(linear(att.sub("key"), h)), split(linear(att.sub("value"), h))
        sc = q @ k.transpose(-1, -2) / math.sqrt(hd) + add_mask
        ctx = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, nh * hd)
        a = layer_norm(linear(att.sub("dense"), ctx) + h, att("LayerNorm.weight"), att("LayerNorm.bias"), eps)
        f = linear(lay.sub("ffn"), a)
        f = F.gelu(f)  # nn.GELU() == exact erf form
        f = linear(lay.sub("ffn_output"), f)
        h = layer_norm(f + a, lay("full_layer_layer_norm.weight"), lay("full_layer_layer_norm.bias"), eps)
    return h


def text_encoder(p: P, input_ids: Tensor, n_layer: int) -> Tensor:
    """TextEncoder (modules.py:21-68), single un-padded utterance -> [B, C, T]."""
    x = p("embedding.weight")[input_ids].transpose(1, 2)
    for i in range(n_layer):
        c = p.sub(f"cnn.{i}")
        k = c.raw("0.weight_v").shape[1]
        x = conv_weighted(c.sub("0"), x, padding=(k - 1) // 2)
        x = layer_norm(x.transpose(1, 2), c("1.weight"), c("1.bias"), 1e-5).transpose(1, 2)
        x = leaky_relu(x, 0.2)
    return bilstm(p.sub("lstm"), x.transpose(1, 2)).transpose(1, 2)


def ada_layer_norm(p: P, x: Tensor, s: Tensor) -> Tensor:
    """AdaLayerNorm (modules.py:71-90): x [B, T, C], s [B, style]."""
    h = linear(p.sub("fc"), p.fq(s))  # kitten_tts/modules.py:80
    gamma, beta = h.chunk(2, dim=1)
    xn = layer_norm(x, None, None, 1e-5)
    return (1 + gamma[:, None, :]) * xn + beta[:, None, :]


def duration_encoder(p: P, d_en: Tensor, s: Tensor, n_layer: int) -> Tensor:
    """DurationEncoder (modules.py:380-411): d_en [B, C, T] -> [B, T, C+style]."""
    B, C, T = d_en.shape
    sty = s[:, None, :].expand(B, T, s.shape[-1])
    x = torch.cat([d_en.transpose(1, 2), sty], dim=-1)  # [B, T, C+S]
    for i in range(n_layer):
        x = bilstm(p.sub(f"lstms.{2 * i}"), x)
        x = ada_layer_norm(p.sub(f"lstms.{2 * i + 1}"), x, s)
        x = torch.cat([x, sty], dim=-1)
    return x


# ------------------------------------------------------------------ source module / STFT head
def sine_source(p: P, f0_frames: Tensor, rand_ini: np.ndarray, noise: np.ndarray,
                upsample: int = 300, sr: int = 24000, harmonics: int = 9,
                sine_amp: float = 0.1, noise_std: float = 0.003, voiced_thr: float = 10.0, coarse_f32: bool = False) -> np.ndarray:
    """f0 nearest-upsample + SineGen + tanh(Linear) (istftnet.py:548-709,797-799).

    f0_frames [B, 2F] -> har_source [B, L=2F*upsample] (float32, numpy).
    ``rand_ini`` [B, harmonics] uniform[0,1) (column 0 is zeroed as in :582);
    ``noise`` [B, L, harmonics] standard normal.
    All arithmetic is fp32 numpy, one rounding per reference op.
    """
    f32 = np.float32
    f0 = np.repeat(f0_frames.detach().to(torch.float32).numpy(), upsample, axis=1)[:, :, None]  # [B, L, 1]
    B, L, _ = f0.shape
    fn = (f0 * np.arange(1, harmonics + 1, dtype=f32)[None, None, :]).astype(f32)
    rad = np.mod((fn / f32(sr)).astype(f32), f32(1.0)).astype(f32)
    ini = np.array(rand_ini, dtype=f32, copy=True)
    ini[:, 0] = 0
    rad[:, 0, :] = (rad[:, 0, :] + ini).astype(f32)
    rad_t = rad.transpose(0, 2, 1)
    # KittenTTS keeps upsample_scale as an mx.array: ``1 / upsample_scale`` is then a float32 (0.0033333334 for 300) and the coarse grid always has
    # 2F + 1 points (kitten_tts/istftnet.py:572,595-599); Kokoro casts to int first and divides in python doubles (istftnet.py:567)
    small = interp_ref.output_size(L, scale_factor=float(np.float32(1.0) / np.float32(upsample)) if coarse_f32 else 1 / upsample)
    rad_ds = interp_ref.interpolate1d(rad_t, small, "linear")  # [B, H, small]
    phase = (np.cumsum(rad_ds, axis=2, dtype=f32) * f32(2.0)).astype(f32)
    phase = (phase * f32(np.pi)).astype(f32)  # (cumsum * 2) * mx.pi, left to right
    big = interp_ref.output_size(small, scale_factor=upsample)
    phase_up = interp_ref.interpolate1d((phase * f32(upsample)).astype(f32), big, "linear")
    sines = np.sin(phase_up).astype(f32).transpose(0, 2, 1)  # [B, big, H]
    sines = (sines * f32(sine_amp)).astype(f32)
    if sines.shape[1] > L:
        sines = sines[:, :L]
    elif sines.shape[1] < L:
        sines = np.pad(sines, ((0, 0), (0, L - sines.shape[1]), (0, 0)))
    uv = (f0 > voiced_thr).astype(f32)
    noise_amp = (uv * f32(noise_std) + (f32(1) - uv) * f32(sine_amp) / f32(3)).astype(f32)
    nz = (noise_amp * np.asarray(noise, dtype=f32)).astype(f32)
    sw = (sines * uv + nz).astype(f32)
    if p.sub("m_source.l_linear").quant:  # kitten_tts/istftnet.py:711-713
        sw = np.stack([fake_quant_dynamic_u8(torch.from_numpy(row)).numpy() for row in sw], axis=0)
    w = p("m_source.l_linear.weight").to(torch.float32).numpy()  # [1, H]
    b = p("m_source.l_linear.bias").to(torch.float32).numpy()
    merged = np.tanh((sw @ w.T.astype(f32) + b.astype(f32)).astype(f32)).astype(f32)
    return merged[:, :, 0]


def stft_mag_phase(x: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """MLXSTFT.transform (istftnet.py:473-506): [B, L] -> [B, 2*(n_fft/2+1), frames]."""
    win = dsp_ref.hanning(n_fft, periodic=True)
    out = []
    for row in x:
        spec = dsp_ref.stft(row, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=win,
                            center=True, pad_mode="reflect").T  # [bins, frames]
        mag = np.abs(spec).astype(np.float32)
        ph = np.arctan2(spec.imag, spec.real).astype(np.float32)
        out.append(np.concatenate([mag, ph], axis=0))
    return np.stack(out, axis=0)


def istft_head(x: Tensor, n_fft: int, hop: int) -> Tensor:
    """exp/sin split + MLXSTFT.inverse (istftnet.py:830-835,508-541): [B, n_fft+2, frames] -> [B, 1, samples]."""
    nb = n_fft // 2 + 1
    xx = x.detach().to(torch.float32).numpy()
    spec = np.exp(xx[:, :nb]).astype(np.float32)
    phase = np.sin(xx[:, nb:]).astype(np.float32)
    # mlx_unwrap is the identity here: |phase| <= 1 so every jump is < pi (istftnet.py:443-444)
    re = (spec * np.cos(phase).astype(np.float32)).astype(np.float32)
    im = (spec * np.sin(phase).astype(np.float32)).astype(np.float32)
    win = dsp_ref.hanning(n_fft, periodic=True)
    outs = [dsp_ref.istft(re[b] + 1j * im[b], hop_length=hop, win_length=n_fft, window=win,
                          center=True, normalized=True) for b in range(xx.shape[0])]
    return torch.from_numpy(np.stack(outs, axis=0))[:, None, :]


def generator(p: P, x: Tensor, s: Tensor, f0_curve: Tensor, cfg: dict, rand_ini, noise, trace=None, coarse_f32: bool = False) -> Tensor:
    """Generator.__call__ (istftnet.py:797-835).  ``trace`` (dict) collects stage outputs for tests."""
    rates, kernels = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    rk, rd = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    n_fft, hop = cfg["gen_istft_n_fft"], cfg["gen_istft_hop_size"]
    total_up = int(np.prod(rates)) * hop
    har_src = sine_source(p, f0_curve, rand_ini, noise, upsample=total_up, coarse_f32=coarse_f32)
    har = torch.from_numpy(stft_mag_phase(har_src, n_fft, hop)).to(x.dtype)  # [B, 22, frames]
    if trace is not None:
        trace.update(har_src=torch.from_numpy(har_src), har=har, xg=x)
    nk = len(rk)
    for i, (u, k) in enumerate(zip(rates, kernels)):
        x = leaky_relu(x, 0.1)
        nc = p.sub(f"noise_convs.{i}")
        wk = nc.raw("weight").shape[1]
        if i + 1 < len(rates):
            stride_f0 = int(np.prod(rates[i + 1:]))
            xs = conv1d_mlx(nc.fq(har), nc("weight"), nc("bias"), stride=stride_f0, padding=(stride_f0 + 1) // 2)  # fq: kitten_tts/istftnet.py:815-818
            nres_k = 7
        else:
            xs = conv1d_mlx(nc.fq(har), nc("weight"), nc("bias"))
            nres_k = 11
        assert wk == (stride_f0 * 2 if i + 1 < len(rates) else 1)
        if trace is not None:
            trace[f"nconv{i}"] = xs
        xs = adain_resblock1(p.sub(f"noise_res.{i}"), xs, s, nres_k, (1, 3, 5))
        if trace is not None:
            trace[f"nres{i}"] = xs
        x = conv_weighted(p.sub(f"ups.{i}"), x, transpose=True, stride=u, padding=(k - u) // 2)
        if i == len(rates) - 1:
            x = F.pad(x, (1, 0))  # "ReflectionPad1d" is a constant zero pad (istftnet.py:712-718)
        x = x + xs
        if trace is not None:
            trace[f"xu{i}"] = x
        acc = None
        for j in range(nk):
            r = adain_resblock1(p.sub(f"resblocks.{i * nk + j}"), x, s, rk[j], tuple(rd[j]))
            acc = r if acc is None else acc + r
        x = acc / nk
        if trace is not None:
            trace[f"stage{i}"] = x
    x = leaky_relu(x, 0.01)
    x = conv_weighted(p.sub("conv_post"), x, padding=3)
    if trace is not None:
        trace["post"] = x
    return istft_head(x, n_fft, hop)


def decoder(p: P, asr: Tensor, f0_curve: Tensor, n_curve: Tensor, s: Tensor, cfg: dict, rand_ini, noise, trace=None, coarse_f32: bool = False) -> Tensor:
    """Decoder.__call__ (istftnet.py:981-997) -> [B, 1, samples]."""
    f0 = conv_weighted(p.sub("F0_conv"), f0_curve[:, None, :], stride=2, padding=1)
    n = conv_weighted(p.sub("N_conv"), n_curve[:, None, :], stride=2, padding=1)
    x = torch.cat([asr, f0, n], dim=1)
    x = adain_resblk1d(p.sub("encode"), x, s, upsample=False)
    if trace is not None:
        trace["dec_in"] = torch.cat([asr, f0, n], dim=1)
        trace["enc"] = x
    asr_res = conv_weighted(p.sub("asr_res.0"), asr, padding=0)
    res = True
    for i in range(4):
        if res:
            x = torch.cat([x, asr_res, f0, n], dim=1)
        up = p.has(f"decode.{i}.pool.weight_v")
        x = adain_resblk1d(p.sub(f"decode.{i}"), x, s, upsample=up)
        if trace is not None:
            trace[f"dec{i}"] = x
        if up:
            res = False
    return generator(p.sub("generator"), x, s, f0_curve, cfg, rand_ini, noise, trace, coarse_f32=coarse_f32)


# ------------------------------------------------------------------ full model
class KokoroRef:
    """Oracle for ``Model.__call__`` (kokoro.py:111-177), batch 1 like the reference."""

    def __init__(self, weights: Dict[str, Tensor], config: dict, dtype=torch.float32,
                 param_dtype=torch.bfloat16):
        self.cfg = config
        self.p = P(weights, "", dtype, param_dtype)
        self.dtype = dtype

    def durations(self, input_ids: Tensor, ref_s: Tensor, speed: float = 1.0):
        """Returns (pred_dur int32 [T], d [1, T, 640], raw duration float [T])."""
        cfg, p = self.cfg, self.p
        ids = input_ids.view(1, -1)
        s = ref_s.to(self.dtype)[:, 128:]
        mask01 = torch.ones_like(ids)
        bert_out = albert(p.sub("bert"), ids, mask01, {**cfg["plbert"]})
        d_en = linear(p.sub("bert_encoder"), bert_out).transpose(1, 2)
        d = duration_encoder(p.sub("predictor.text_encoder"), d_en, s, cfg["n_layer"])
        x = bilstm(p.sub("predictor.lstm"), d)
        logits = linear(p.sub("predictor.duration_proj.linear_layer"), x)
        dur = torch.sigmoid(logits).sum(dim=-1) / speed
        dur = torch.nan_to_num(dur, nan=1.0, posinf=100.0, neginf=1.0)
        pred = torch.clamp(torch.round(dur), 1, 100).to(torch.int32)[0]
        return pred, d, dur[0]

    def forward(self, input_ids: Tensor, ref_s: Tensor, speed: float = 1.0,
                rand_ini: Optional[np.ndarray] = None, noise: Optional[np.ndarray] = None,
                pred_dur: Optional[Tensor] = None, noise_seed: int = 1234, return_intermediates=False,
                f0_override: Optional[Tensor] = None, n_override: Optional[Tensor] = None):
        """input_ids: LongTensor [T] INCLUDING the leading/trailing 0 tokens; ref_s [1, 256].  ``f0_override`` / ``n_override`` [1, 2F] replace the
        predicted pitch / energy curves (tests: the harmonic source integrates F0 into a phase, so vocoder comparisons inject the other side's curves)."""
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
            en = d.transpose(1, 2)[:, :, idx]  # one-hot matmul == gather (kokoro.py:161-164)
            pr = p.sub("predictor")
            x = bilstm(pr.sub("shared"), en.transpose(1, 2))  # [1, F, 512]
            f0 = x.transpose(1, 2)
            nn_ = x.transpose(1, 2)
            for i in range(3):
                f0 = adain_resblk1d(pr.sub(f"F0.{i}"), f0, s_pred, upsample=pr.has(f"F0.{i}.pool.weight_v"))
                nn_ = adain_resblk1d(pr.sub(f"N.{i}"), nn_, s_pred, upsample=pr.has(f"N.{i}.pool.weight_v"))
            f0 = conv1d_mlx(f0, pr("F0_proj.weight"), pr("F0_proj.bias"))[:, 0, :]
            nn_ = conv1d_mlx(nn_, pr("N_proj.weight"), pr("N_proj.bias"))[:, 0, :]
            if f0_override is not None:
                f0 = f0_override.to(self.dtype)
            if n_override is not None:
                nn_ = n_override.to(self.dtype)
            t_en = text_encoder(p.sub("text_encoder"), ids, cfg["n_layer"])
            asr = t_en[:, :, idx]
            if rand_ini is None or noise is None:
                rng = np.random.default_rng(noise_seed)
                up = int(np.prod(cfg["istftnet"]["upsample_rates"])) * cfg["istftnet"]["gen_istft_hop_size"]
                rand_ini = rng.uniform(size=(1, 9)).astype(np.float32)
                noise = rng.standard_normal((1, 2 * Fr * up, 9)).astype(np.float32)
            trace = {} if return_intermediates else None
            audio = decoder(p.sub("decoder"), asr, f0, nn_, ref_s[:, :128], cfg["istftnet"], rand_ini, noise, trace)[0]
            if return_intermediates:
                return audio, pred_dur, dict(d=d, en=en, f0=f0, n=nn_, asr=asr, raw_dur=raw, **trace)
            return audio, pred_dur
