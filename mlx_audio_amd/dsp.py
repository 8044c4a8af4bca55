"""``mlx_audio.dsp`` on MI355X: same names, keyword arguments and defaults as the reference (``mlx_audio/dsp.py:39-94,
385-752``), torch tensors on the ROCm device in and out, every transform a hand-written gfx950 kernel
(``csrc/fft.hip`` through the C ABI).  This module imports neither ``tts`` nor ``stt`` (the reference enforces the
same for its ``dsp``: ``mlx_audio/tests/test_dsp.py:10-27``).

One-time constants (windows, mel filterbanks, overlap-add envelopes) are built on the host with the reference's
arithmetic (Python floats / float32 numpy, cached) and uploaded; the per-call work -- framing, padding, FFT,
|X|^2, mel projection, log, overlap-add -- runs on the GPU.

Besides the reference's 1-D ``stft`` / ``istft`` (callers loop over the batch in Python there: istftnet.py:481,
qwen3_tts.py:94), both accept a leading batch axis and process it in one launch.  ``log_mel_spectrogram`` (Whisper,
stt/models/whisper/audio.py:41-82) and ``mel_spectrogram`` (Qwen3-TTS speaker encoder, qwen3_tts.py:64-120) are the
fused STFT -> power -> mel -> log front ends.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Optional, Union

import numpy as np
import torch

__all__ = ["hanning", "hamming", "blackman", "bartlett", "STR_TO_WINDOW_FN", "stft", "istft", "mel_filters", "ISTFTCache",
           "log_mel_spectrogram", "mel_spectrogram",
           # host-side level helpers of mlx_audio.dsp (dsp.py:96-382): out of the hot-path scope, resolved lazily from .host.loudness
           "integrated_loudness", "lfilter", "normalize_loudness", "normalize_peak"]

_HOST_LEVEL_NAMES = ("integrated_loudness", "lfilter", "normalize_loudness", "normalize_peak", "_biquad_coefficients", "_K_WEIGHT_HIGHPASS_FREQ",
                     "_K_WEIGHT_HIGHPASS_Q", "_K_WEIGHT_SHELF_FREQ", "_K_WEIGHT_SHELF_GAIN_DB", "_K_WEIGHT_SHELF_Q")


def __getattr__(name):   # ``from mlx_audio_amd.dsp import integrated_loudness`` keeps working like the reference; the numpy module loads on first use
    if name in _HOST_LEVEL_NAMES:
        from .host import loudness

        return getattr(loudness, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def _device():
    from . import ops

    ops.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


# ------------------------------------------------------------------------------------------------ windows
def _window(values) -> torch.Tensor:
    return torch.tensor(values, dtype=torch.float64).to(torch.float32)  # Python floats -> fp32, as mx.array(list)


@lru_cache(maxsize=None)
def hanning(size: int, periodic: bool = False) -> torch.Tensor:
    d = size if periodic else size - 1
    return _window([0.5 * (1 - math.cos(2 * math.pi * n / d)) for n in range(size)])


@lru_cache(maxsize=None)
def hamming(size: int, periodic: bool = False) -> torch.Tensor:
    d = size if periodic else size - 1
    return _window([0.54 - 0.46 * math.cos(2 * math.pi * n / d) for n in range(size)])


@lru_cache(maxsize=None)
def blackman(size: int, periodic: bool = False) -> torch.Tensor:
    d = size if periodic else size - 1
    return _window([0.42 - 0.5 * math.cos(2 * math.pi * n / d) + 0.08 * math.cos(4 * math.pi * n / d) for n in range(size)])


@lru_cache(maxsize=None)
def bartlett(size: int, periodic: bool = False) -> torch.Tensor:
    d = size if periodic else size - 1
    return _window([1 - 2 * abs(n - d / 2) / d for n in range(size)])


STR_TO_WINDOW_FN = {"hann": hanning, "hanning": hanning, "hamming": hamming, "blackman": blackman, "bartlett": bartlett}


def _resolve_window(window, length: int, periodic_like_istft: bool) -> torch.Tensor:
    if isinstance(window, str):
        fn = STR_TO_WINDOW_FN.get(window.lower())
        if fn is None:
            raise ValueError(f"Unknown window function: {window}")
        # stft: symmetric fn(win_length) (dsp.py:403); istft: fn(win_length + 1)[:-1], i.e. periodic (dsp.py:472)
        return fn(length + 1)[:-1] if periodic_like_istft else fn(length)
    return torch.as_tensor(window, dtype=torch.float32)


def _pad_to(w: torch.Tensor, n: int) -> torch.Tensor:
    return w if w.shape[0] >= n else torch.cat([w.cpu(), torch.zeros(n - w.shape[0])])


_PAD_MODES = {"reflect": 1, "constant": 2}


# ------------------------------------------------------------------------------------------------ stft / istft
def stft(x, n_fft: int = 800, hop_length: Optional[int] = None, win_length: Optional[int] = None,
         window: Union[str, torch.Tensor] = "hann", center: bool = True, pad_mode: str = "reflect") -> torch.Tensor:
    """``[L]`` -> complex64 ``[n_frames, n_fft//2+1]`` (or ``[B, L]`` -> ``[B, n_frames, n_fft//2+1]``)."""
    from . import ops

    dev = _device()
    hop_length = n_fft // 4 if hop_length is None else hop_length
    win_length = n_fft if win_length is None else win_length
    w = _pad_to(_resolve_window(window, win_length, False), n_fft).to(dev).contiguous()
    if center and pad_mode not in _PAD_MODES:
        raise ValueError(f"Invalid pad_mode {pad_mode}")
    x = torch.as_tensor(x, dtype=torch.float32).to(dev)
    batched = x.dim() == 2
    xb = (x if batched else x[None]).contiguous()
    L = xb.shape[1]
    padded = L + 2 * (n_fft // 2) if center else L
    n_frames = 1 + (padded - n_fft) // hop_length
    if n_frames <= 0 or (center and pad_mode == "reflect" and L <= n_fft // 2):
        raise ValueError(f"Input is too short (length={padded}) for n_fft={n_fft} with hop_length={hop_length} and center={center}.")
    out = ops.stft_frames(xb, n_fft, hop_length, w, _PAD_MODES[pad_mode] if center else 0, n_frames)
    return out if batched else out[0]


@lru_cache(maxsize=64)
def _ola_envelope(window_key: bytes, n: int, n_frames: int, hop: int, squared: bool) -> np.ndarray:
    w = np.frombuffer(window_key, dtype=np.float32)
    env = np.zeros((n_frames - 1) * hop + n, dtype=np.float32)
    wn = (w * w).astype(np.float32) if squared else w
    for f in range(n_frames):  # frame order, fp32 accumulation: what the scatter-add produces (dsp.py:499-500)
        env[f * hop: f * hop + n] += wn
    return env


def istft(x, hop_length: Optional[int] = None, win_length: Optional[int] = None, window: Union[str, torch.Tensor] = "hann",
          center: bool = True, length: Optional[int] = None, normalized: bool = False) -> torch.Tensor:
    """complex ``[n_fft//2+1, n_frames]`` (optionally with a leading batch axis) -> signal (dsp.py:436-513)."""
    from . import ops

    dev = _device()
    x = torch.as_tensor(x).to(dev)
    batched = x.dim() == 3
    xb = x if batched else x[None]
    n_frames = xb.shape[2]
    win_length = (n_frames - 1) * 2 if win_length is None else win_length  # sic: the reference reads axis 1 (dsp.py:463)
    hop_length = win_length // 4 if hop_length is None else hop_length
    n = (xb.shape[1] - 1) * 2
    if n != win_length:
        raise ValueError(f"istft: irfft length {n} must equal win_length {win_length} (the reference multiplies frames * window)")
    w = _pad_to(_resolve_window(window, win_length, True), win_length).to(torch.float32).cpu().contiguous()
    env = _ola_envelope(w.numpy().tobytes(), n, n_frames, hop_length, bool(normalized))
    total = env.shape[0]
    trim = win_length // 2 if (center and length is None) else 0
    out_len = total - 2 * trim if (center and length is None) else (min(length, total) if length is not None else total)
    spec = xb.to(torch.complex64).transpose(1, 2).contiguous()  # kernels take [B, n_frames, bins]
    y = ops.istft_frames(spec, n, hop_length, w.to(dev), torch.from_numpy(env).to(dev), 1, False, trim, out_len)
    return y if batched else y[0]


# ------------------------------------------------------------------------------------------------ mel filterbank
@lru_cache(maxsize=None)
def _mel_filters_host(sample_rate: int, n_fft: int, n_mels: int, f_min: float, f_max: Optional[float], norm: Optional[str],
                      mel_scale: Optional[str], precise: bool) -> np.ndarray:
    dt = np.float64 if precise else np.float32
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0

    def hz_to_mel(f: float) -> float:
        if mel_scale == "htk":
            return 2595.0 * math.log10(1.0 + f / 700.0)
        return min_log_hz / f_sp + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    def mel_to_hz(m: np.ndarray) -> np.ndarray:
        if mel_scale == "htk":
            return (700.0 * (10.0 ** (m / dt(2595.0)) - 1.0)).astype(dt)
        lin = (dt(f_sp) * m).astype(dt)
        log = (dt(min_log_hz) * np.exp(dt(logstep) * (m - dt(min_log_hz / f_sp)))).astype(dt)
        return np.where(m >= min_log_hz / f_sp, log, lin).astype(dt)

    f_max = f_max or sample_rate / 2
    all_freqs = np.linspace(0, sample_rate // 2, n_fft // 2 + 1).astype(dt)
    f_pts = mel_to_hz(np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2).astype(dt))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    fb = np.maximum(dt(0), np.minimum((-slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:])).astype(dt)
    if norm == "slaney":
        fb = (fb * (dt(2.0) / (f_pts[2: n_mels + 2] - f_pts[:n_mels]))[None, :]).astype(dt)
    return np.ascontiguousarray(fb.T).astype(np.float32)


def mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 0, f_max: Optional[float] = None,
                norm: Optional[str] = None, mel_scale: str = "htk", precise: bool = False) -> torch.Tensor:
    """Triangular mel filterbank ``[n_mels, n_fft//2+1]`` (dsp.py:519-609).  Any ``mel_scale`` other than "htk"
    (``None`` included) is Slaney; ``precise`` builds in float64.  Returned on the CPU (upload is the caller's, the
    fused front ends below cache their device copy)."""
    return torch.from_numpy(_mel_filters_host(sample_rate, n_fft, n_mels, f_min, f_max, norm, mel_scale, precise).copy())


# ------------------------------------------------------------------------------------------------ ISTFTCache
class ISTFTCache:
    """Batched inverse STFT with cached w^2 overlap-add envelopes (dsp.py:612-752).  The positions cache of the
    reference has no role here (the overlap-add is a gather inside the kernel), ``cache_info`` keeps its keys."""

    def __init__(self):
        self.norm_buffer_cache = {}
        self.position_cache = {}

    def get_norm_buffer(self, n_fft: int, hop_length: int, win_length: int, window: torch.Tensor, num_frames: int) -> torch.Tensor:
        wh = window.detach().to(torch.float32).cpu().contiguous()
        key = (n_fft, hop_length, win_length, hash(wh.numpy().tobytes()), num_frames)
        if key not in self.norm_buffer_cache:
            env = np.maximum(_ola_envelope(wh.numpy().tobytes(), wh.shape[0], num_frames, hop_length, True), np.float32(1e-10))
            self.norm_buffer_cache[key] = torch.from_numpy(env).to(_device())
        return self.norm_buffer_cache[key]

    def istft(self, real_part, imag_part, n_fft: int, hop_length: int, win_length: int, window, center: bool = True,
              audio_length: int = None, constrain_value_range: bool = False) -> torch.Tensor:
        """``real_part`` / ``imag_part``: ``[batch, freq, time]`` -> ``[batch, samples]`` (head trimmed by n_fft//2 if
        ``center``; the tail is kept, as in the reference)."""
        from . import ops

        dev = _device()
        w = _pad_to(torch.as_tensor(window, dtype=torch.float32).cpu(), n_fft)
        real_part = torch.as_tensor(real_part, dtype=torch.float32).to(dev)
        imag_part = torch.as_tensor(imag_part, dtype=torch.float32).to(dev)
        n_frames = real_part.shape[2]
        norm = self.get_norm_buffer(n_fft, hop_length, win_length, w, n_frames)
        total = (n_frames - 1) * hop_length + n_fft
        trim = n_fft // 2 if center else 0
        out_len = total - trim
        if audio_length is not None:
            out_len = min(out_len, audio_length)
        spec = torch.complex(real_part, imag_part).transpose(1, 2).contiguous()
        return ops.istft_frames(spec, n_fft, hop_length, w.to(dev), norm, 0, bool(constrain_value_range), trim, out_len)

    def clear_cache(self):
        self.norm_buffer_cache.clear()
        self.position_cache.clear()

    def cache_info(self):
        return {"norm_buffers": len(self.norm_buffer_cache), "position_indices": len(self.position_cache),
                "total_cached_items": len(self.norm_buffer_cache) + len(self.position_cache)}


# ------------------------------------------------------------------------------------------------ fused mel front ends
@lru_cache(maxsize=16)
def _device_consts(kind: str, dev_index: int):
    dev = torch.device("cuda", dev_index)
    if kind == "whisper80" or kind == "whisper128":
        n_mels = 80 if kind == "whisper80" else 128
        return hanning(400).to(dev), mel_filters(16000, 400, n_mels, norm="slaney", mel_scale=None).to(dev).contiguous()
    raise KeyError(kind)


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0) -> torch.Tensor:
    """Whisper's log-mel front end (stt/models/whisper/audio.py:41-82): ``[L]`` (or ``[B, L]``) 16 kHz samples ->
    ``[n_frames, n_mels]``.  Symmetric Hann(400), hop 160, reflect-centred, last frame dropped, Slaney filters,
    ``log10(max(., 1e-10))``, clamp to (global max - 8), ``(. + 4) / 4`` -- one fused kernel + a finishing pass."""
    from . import ops

    dev = _device()
    x = torch.as_tensor(audio, dtype=torch.float32).to(dev)
    batched = x.dim() == 2
    x = x if batched else x[None]
    if padding > 0:
        x = torch.nn.functional.pad(x, (0, padding))
    x = x.contiguous()
    win, fb = _device_consts("whisper80" if n_mels == 80 else "whisper128", dev.index or 0)
    n_frames = (1 + (x.shape[1] + 400 - 400) // 160) - 1
    out = ops.logmel(x, 400, 160, win, 1, n_frames, fb, 0)
    return out if batched else out[0]


def mel_spectrogram(audio, n_fft: int = 1024, num_mels: int = 128, sample_rate: int = 24000, hop_size: int = 256,
                    win_size: int = 1024, fmin: int = 0, fmax: Optional[int] = 12000) -> torch.Tensor:
    """Qwen3-TTS speaker-encoder mel (qwen3_tts.py:64-120): manual reflect pad (n_fft - hop)/2, symmetric Hann,
    ``sqrt(|X|^2 + 1e-9)``, Slaney mel, ``log(clip(., 1e-5))``: ``[B, L]`` (or ``[L]``) -> ``[B, frames, num_mels]``."""
    from . import ops

    dev = _device()
    x = torch.as_tensor(audio, dtype=torch.float32).to(dev)
    if x.dim() == 1:
        x = x[None]
    pad = (n_fft - hop_size) // 2
    x = torch.cat([x[:, 1: pad + 1].flip(1), x, x[:, -(pad + 1): -1].flip(1)], 1).contiguous()
    win = _pad_to(hanning(win_size), n_fft).to(dev)
    fb = mel_filters(sample_rate, n_fft, num_mels, fmin, fmax, norm="slaney", mel_scale="slaney").to(dev).contiguous()
    n_frames = 1 + (x.shape[1] - n_fft) // hop_size
    return ops.logmel(x, n_fft, hop_size, win, 0, n_frames, fb, 1)


# ------------------------------------------------------------------------------------------------ Kaldi fbank (dsp.py:806-997)
def mel_scale_kaldi(freq):
    return 1127.0 * np.log(1.0 + np.asarray(freq, dtype=np.float32) / np.float32(700.0))


def inverse_mel_scale_kaldi(mel_freq):
    return 700.0 * (np.exp(np.asarray(mel_freq, dtype=np.float32) / np.float32(1127.0)) - 1.0)


def _next_power_of_2(x: int) -> int:
    return 1 if x == 0 else 2 ** (x - 1).bit_length()


def get_mel_banks_kaldi(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float, high_freq: float):
    """Kaldi mel filterbank (dsp.py:846-895), float32 on the host: (bins [num_bins, n_fft/2], center_freqs [num_bins])."""
    assert num_bins > 3, "Must have at least 3 mel bins"
    assert window_length_padded % 2 == 0
    f32 = np.float32
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    assert (0.0 <= low_freq < nyquist) and (0.0 < high_freq <= nyquist)
    fft_bin_width = f32(sample_freq / window_length_padded)
    mel_low, mel_high = float(mel_scale_kaldi(low_freq)), float(mel_scale_kaldi(high_freq))
    delta = f32((mel_high - mel_low) / (num_bins + 1))
    idx = np.arange(num_bins, dtype=f32).reshape(-1, 1)
    left, center, right = f32(mel_low) + idx * delta, f32(mel_low) + (idx + f32(1.0)) * delta, f32(mel_low) + (idx + f32(2.0)) * delta
    center_freqs = inverse_mel_scale_kaldi(center)
    mel = mel_scale_kaldi(fft_bin_width * np.arange(num_fft_bins, dtype=f32)).reshape(1, -1)
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    bins = np.maximum(f32(0.0), np.minimum(up, down)).astype(f32)
    return torch.from_numpy(bins), torch.from_numpy(center_freqs.squeeze().astype(f32))


def compute_fbank_kaldi(waveform, sample_rate: int = 48000, win_len: int = 1920, win_inc: int = 384, num_mels: int = 60, win_type: str = "hamming",
                        preemphasis: float = 0.97, dither: float = 1.0, snip_edges: bool = True, low_freq: float = 20.0, high_freq: float = 0.0,
                        noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Kaldi-compatible log mel-filterbank features ``[time, num_mels]`` (``mlx_audio.dsp.compute_fbank_kaldi``, dsp.py:898-997): same
    arguments and defaults; ``noise`` (extra) supplies the standard-normal dither draws ``[n_frames, window]`` explicitly, otherwise they
    come from ``torch.randn`` on the device (the reference draws from ``mx.random.normal``: not reproducible across the two either way).
    Two launches: Kaldi framing (DC removal, pre-emphasis, window, zero pad) and the fused FFT -> |X|^2 -> mel -> log kernel."""
    from . import ops

    dev = _device()
    x = torch.as_tensor(waveform, dtype=torch.float32)
    if x.dim() == 2:
        x = x[0]
    x = x.to(dev).contiguous()
    frame_length_ms, frame_shift_ms = win_len / sample_rate * 1000, win_inc / sample_rate * 1000
    shift = int(sample_rate * frame_shift_ms * 0.001)
    win = int(sample_rate * frame_length_ms * 0.001)
    P = _next_power_of_2(win)
    L = x.numel()
    if snip_edges:
        if L < win:
            return torch.zeros((0, num_mels), dtype=torch.float32, device=dev)
        m, pad = 1 + (L - win) // shift, 0
    else:
        m = (L + (shift // 2)) // shift
        pad = win // 2 - shift // 2
        if pad <= 0:
            raise NotImplementedError("compute_fbank_kaldi(snip_edges=False) with window <= shift")
    if m == 0:
        return torch.zeros((0, num_mels), dtype=torch.float32, device=dev)
    n = torch.arange(win, dtype=torch.float32)
    if win_type == "hamming":
        w = 0.54 - 0.46 * torch.cos(2 * math.pi * n / (win - 1))
    elif win_type == "hanning":
        w = 0.5 - 0.5 * torch.cos(2 * math.pi * n / (win - 1))
    elif win_type == "povey":
        w = torch.pow(0.5 - 0.5 * torch.cos(2 * math.pi * n / (win - 1)), 0.85)
    else:
        w = torch.ones(win)
    if dither != 0.0 and noise is None:
        noise = torch.randn((m, win), dtype=torch.float32, device=dev)
    if dither == 0.0:
        noise = None
    frames = ops.kaldi_frames(x, win, shift, pad, P, m, w.to(dev), float(preemphasis),
                              None if noise is None else torch.as_tensor(noise, dtype=torch.float32).to(dev).contiguous(), float(dither))
    bins, _ = get_mel_banks_kaldi(num_mels, P, float(sample_rate), low_freq, high_freq)
    fb = torch.nn.functional.pad(bins, (0, 1)).contiguous().to(dev)           # [num_mels, P/2 + 1] (dsp.py:990)
    ones = torch.ones(P, dtype=torch.float32, device=dev)
    return ops.logmel(frames.view(1, m * P), P, P, ones, 0, m, fb, 2)[0]


def compute_deltas_kaldi(specgram, win_length: int = 5, mode: str = "edge") -> torch.Tensor:
    """Delta coefficients of a spectrogram ``(..., freq, time)`` (``mlx_audio/dsp.py:760-804``, Kaldi / torchaudio ``compute_deltas``):
    ``d_t = sum_{n=1..N} n (c_{t+n} - c_{t-n}) / (2 sum n^2)`` with ``N = (win_length - 1) // 2`` and edge (or zero) padding in time.  A handful of
    shifted adds on the tensor's own device (host glue around the fbank kernel, not a hot-path kernel).  The reference multiplies a ``win_length``-wide
    window by ``2 N + 1`` weights, which only broadcasts for odd ``win_length``; even values are rejected here instead of failing inside."""
    if win_length < 3:
        raise ValueError(f"win_length should be >= 3, got {win_length}")
    if win_length % 2 == 0:
        raise ValueError(f"win_length must be odd (the reference's window / weight shapes only agree for odd values), got {win_length}")
    if mode not in ("edge", "constant"):
        raise ValueError(f"mode must be 'edge' or 'constant', got {mode!r}")
    x = specgram if isinstance(specgram, torch.Tensor) else torch.as_tensor(np.asarray(specgram))
    if not x.is_floating_point():
        x = x.to(torch.float32)
    n = (win_length - 1) // 2
    denom = float(n * (n + 1) * (2 * n + 1)) / 3.0
    flat = x.reshape(-1, x.shape[-1])
    T = flat.shape[1]
    if mode == "edge":
        padded = torch.cat([flat[:, :1].expand(-1, n), flat, flat[:, -1:].expand(-1, n)], dim=1)
    else:
        padded = torch.nn.functional.pad(flat, (n, n))
    out = torch.zeros_like(flat)
    for k in range(1, n + 1):
        out = out + float(k) * (padded[:, n + k:n + k + T] - padded[:, n - k:n - k + T])
    return (out / denom).reshape(x.shape)


__all__ += ["compute_deltas_kaldi", "mel_scale_kaldi", "inverse_mel_scale_kaldi", "get_mel_banks_kaldi", "compute_fbank_kaldi"]  # the rest of mlx_audio.dsp.__all__
