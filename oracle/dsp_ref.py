"""numpy restatement of the reference's DSP front/back ends (TEST ORACLE, not product).

Follows ``/root/reference/mlx_audio/dsp.py`` (windows :39-94, stft :385-433,
istft :436-513, mel_filters :519-609, ISTFTCache :612-752) and the two mel
front ends that sit directly on top of it
(``stt/models/whisper/audio.py:41-82``, ``tts/models/qwen3_tts/qwen3_tts.py:64-120``).

The reference delegates the FFT itself to ``mx.fft.rfft/irfft`` (mlx 0.31.2, not
vendored); here it is ``numpy.fft`` evaluated in float64 and rounded once to
float32/complex64, which is the correctly-rounded value any fp32 FFT
approximates.  Everything else mirrors the reference's float32 arithmetic.

Parity status: **pinned twice**.  (1) The reference's own golden vectors and known-answer tests for this path (SURVEY section 8c:
``tts/tests/test_qwen3_tts.py:175-353`` STFT + mel vectors, ``tts/tests/test_istftnet_fidelity.py:18-46`` MLXSTFT round trip, ``mlx_audio/tests/test_dsp.py:62-95`` ISTFTCache bound), transcribed
into tests/golden/reference_vectors.json and checked by tests/test_oracle_golden.py.  (2) The reference's ``dsp.py`` and ``mel_spectrogram``
executed as they are (imported from /root/reference over the numpy stand-in for MLX, tests/golden/make_reference_fixtures.py ``run_dsp``) on seeded
noise + tones: ``ref_dsp.npz``, against which tests/test_reference_fixtures_cpu.py finds this restatement bit-identical on the forward transforms
and filter banks and within 2e-7 on the inverses.
"""
from __future__ import annotations

import math
from typing import Optional, Union

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- windows
def hanning(size: int, periodic: bool = False) -> np.ndarray:
    """dsp.py:39-50: 0.5*(1-cos(2*pi*n/denom)), denom=N (periodic) or N-1."""
    denom = size if periodic else size - 1
    return np.asarray(
        [0.5 * (1 - math.cos(2 * math.pi * n / denom)) for n in range(size)], dtype=F32
    )


def hamming(size: int, periodic: bool = False) -> np.ndarray:
    """dsp.py:53-65."""
    denom = size if periodic else size - 1
    return np.asarray(
        [0.54 - 0.46 * math.cos(2 * math.pi * n / denom) for n in range(size)], dtype=F32
    )


def blackman(size: int, periodic: bool = False) -> np.ndarray:
    """dsp.py:68-80."""
    denom = size if periodic else size - 1
    return np.asarray(
        [
            0.42
            - 0.5 * math.cos(2 * math.pi * n / denom)
            + 0.08 * math.cos(4 * math.pi * n / denom)
            for n in range(size)
        ],
        dtype=F32,
    )


def bartlett(size: int, periodic: bool = False) -> np.ndarray:
    """dsp.py:83-87."""
    denom = size if periodic else size - 1
    return np.asarray([1 - 2 * abs(n - denom / 2) / denom for n in range(size)], dtype=F32)


STR_TO_WINDOW_FN = {
    "hann": hanning,
    "hanning": hanning,
    "hamming": hamming,
    "blackman": blackman,
    "bartlett": bartlett,
}


def _resolve_window(window, length: int, for_istft: bool) -> np.ndarray:
    if isinstance(window, str):
        fn = STR_TO_WINDOW_FN.get(window.lower())
        if fn is None:
            raise ValueError(f"Unknown window function: {window}")
        # dsp.py:403 (stft: symmetric fn(win_length)) vs dsp.py:472
        # (istft: fn(win_length+1)[:-1] == periodic)
        return fn(length + 1)[:-1] if for_istft else fn(length)
    return np.asarray(window, dtype=F32)


# --------------------------------------------------------------------------- stft
def stft(
    x,
    n_fft: int = 800,
    hop_length: Optional[int] = None,
    win_length: Optional[int] = None,
    window: Union[str, np.ndarray] = "hann",
    center: bool = True,
    pad_mode: str = "reflect",
) -> np.ndarray:
    """dsp.py:385-433.  1-D float32 ``x`` -> complex64 ``[n_frames, n_fft//2+1]``."""
    x = np.asarray(x, dtype=F32)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    w = _resolve_window(window, win_length, for_istft=False)
    if w.shape[0] < n_fft:  # right zero-pad the window (dsp.py:408-410)
        w = np.concatenate([w, np.zeros(n_fft - w.shape[0], dtype=F32)])
    if center:
        p = n_fft // 2
        if pad_mode == "constant":
            x = np.concatenate([np.zeros(p, F32), x, np.zeros(p, F32)])
        elif pad_mode == "reflect":
            x = np.concatenate([x[1 : p + 1][::-1], x, x[-(p + 1) : -1][::-1]])
        else:
            raise ValueError(f"Invalid pad_mode {pad_mode}")
    n_frames = 1 + (x.shape[0] - n_fft) // hop_length
    if n_frames <= 0:
        raise ValueError(
            f"Input is too short (length={x.shape[0]}) for n_fft={n_fft} with "
            f"hop_length={hop_length} and center={center}."
        )
    idx = np.arange(n_frames)[:, None] * hop_length + np.arange(n_fft)[None, :]
    frames = (x[idx] * w[None, :]).astype(F32)  # fp32 product, as frames*w in fp32
    return np.fft.rfft(frames.astype(np.float64), axis=-1).astype(np.complex64)


# --------------------------------------------------------------------------- istft
def istft(
    x,
    hop_length: Optional[int] = None,
    win_length: Optional[int] = None,
    window: Union[str, np.ndarray] = "hann",
    center: bool = True,
    length: Optional[int] = None,
    normalized: bool = False,
) -> np.ndarray:
    """dsp.py:436-513.  complex ``[n_fft//2+1, n_frames]`` -> float32 signal."""
    x = np.asarray(x)
    if win_length is None:
        win_length = (x.shape[1] - 1) * 2  # sic: reference uses axis 1 (dsp.py:463)
    if hop_length is None:
        hop_length = win_length // 4
    w = _resolve_window(window, win_length, for_istft=True)
    if w.shape[0] < win_length:
        w = np.concatenate([w, np.zeros(win_length - w.shape[0], dtype=F32)])
    n_frames = x.shape[1]
    total = (n_frames - 1) * hop_length + win_length
    frames = np.fft.irfft(x.astype(np.complex128), axis=0).T.astype(F32)  # [frames, n]
    recon = np.zeros(total, dtype=F32)
    wsum = np.zeros(total, dtype=F32)
    wnorm = (w * w).astype(F32) if normalized else w
    contrib = (frames * w[None, :]).astype(F32)
    # scatter-add in frame order (dsp.py:499-500); fp32 accumulation
    for f in range(n_frames):
        s = f * hop_length
        recon[s : s + win_length] += contrib[f]
        wsum[s : s + win_length] += wnorm
    ok = wsum > 1e-10
    recon = np.where(ok, recon / np.where(ok, wsum, 1), recon).astype(F32)
    if center and length is None:
        recon = recon[win_length // 2 : -win_length // 2]
    if length is not None:
        recon = recon[:length]
    return recon


# --------------------------------------------------------------------------- mel filterbank
def mel_filters(
    sample_rate: int,
    n_fft: int,
    n_mels: int,
    f_min: float = 0,
    f_max: Optional[float] = None,
    norm: Optional[str] = None,
    mel_scale: Optional[str] = "htk",
    precise: bool = False,
) -> np.ndarray:
    """dsp.py:519-609.  Returns float32 ``[n_mels, n_fft//2+1]``.

    ``mel_scale`` anything other than "htk" (including ``None``) selects Slaney
    (dsp.py:541).  ``precise`` builds in float64 then casts (dsp.py:605-608).
    """
    dt = np.float64 if precise else F32

    def hz_to_mel(freq: float) -> float:
        if mel_scale == "htk":
            return 2595.0 * math.log10(1.0 + freq / 700.0)
        f_sp = 200.0 / 3
        mels = freq / f_sp
        if freq >= 1000.0:
            mels = 1000.0 / f_sp + math.log(freq / 1000.0) / (math.log(6.4) / 27.0)
        return mels

    def mel_to_hz(m: np.ndarray) -> np.ndarray:
        if mel_scale == "htk":
            return (700.0 * (10.0 ** (m / dt(2595.0)) - 1.0)).astype(dt)
        f_sp = 200.0 / 3
        min_log_mel = 1000.0 / f_sp
        logstep = math.log(6.4) / 27.0
        lin = (dt(f_sp) * m).astype(dt)
        log = (dt(1000.0) * np.exp(dt(logstep) * (m - dt(min_log_mel)))).astype(dt)
        return np.where(m >= min_log_mel, log, lin).astype(dt)

    f_max = f_max or sample_rate / 2
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs).astype(dt)
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2).astype(dt)
    f_pts = mel_to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]  # [n_freqs, n_mels+2]
    down = (-slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(dt(0), np.minimum(down, up)).astype(dt)
    if norm == "slaney":
        enorm = dt(2.0) / (f_pts[2 : n_mels + 2] - f_pts[:n_mels])
        fb = (fb * enorm[None, :]).astype(dt)
    return np.ascontiguousarray(fb.T).astype(F32)


# --------------------------------------------------------------------------- ISTFTCache
class ISTFTCache:
    """dsp.py:612-752: batched irfft + w**2 overlap-add with cached normaliser."""

    def __init__(self):
        self.norm_buffer_cache = {}
        self.position_cache = {}

    def get_positions(self, num_frames, frame_length, hop_length):
        key = (num_frames, frame_length, hop_length)
        if key not in self.position_cache:
            pos = np.arange(num_frames)[:, None] * hop_length + np.arange(frame_length)[None, :]
            self.position_cache[key] = pos.reshape(-1)
        return self.position_cache[key]

    def get_norm_buffer(self, n_fft, hop_length, win_length, window, num_frames):
        key = (n_fft, hop_length, win_length, hash(tuple(np.asarray(window).tolist())), num_frames)
        if key not in self.norm_buffer_cache:
            flen = window.shape[0]
            ola = (num_frames - 1) * hop_length + flen
            buf = np.zeros(ola, dtype=F32)
            w2 = (window.astype(F32) ** 2).astype(F32)
            for f in range(num_frames):
                buf[f * hop_length : f * hop_length + flen] += w2
            self.norm_buffer_cache[key] = np.maximum(buf, F32(1e-10))
        return self.norm_buffer_cache[key]

    def istft(
        self,
        real_part,
        imag_part,
        n_fft,
        hop_length,
        win_length,
        window,
        center=True,
        audio_length=None,
        constrain_value_range=False,
    ):
        real_part = np.asarray(real_part, dtype=F32)
        imag_part = np.asarray(imag_part, dtype=F32)
        window = np.asarray(window, dtype=F32)
        if window.shape[0] < n_fft:
            window = np.concatenate([window, np.zeros(n_fft - window.shape[0], F32)])
        spec = real_part.astype(np.float64) + 1j * imag_part.astype(np.float64)
        frames = np.fft.irfft(spec.transpose(0, 2, 1), n=n_fft, axis=-1).astype(F32)
        if constrain_value_range:
            frames = np.clip(frames, -window, window)
        frames = (frames * window).astype(F32)
        bsz, nfr, flen = frames.shape
        ola = (nfr - 1) * hop_length + flen
        norm = self.get_norm_buffer(n_fft, hop_length, win_length, window, nfr)
        out = np.zeros((bsz, ola), dtype=F32)
        for f in range(nfr):
            out[:, f * hop_length : f * hop_length + flen] += frames[:, f]
        out = (out / norm[None, :]).astype(F32)
        if center:
            out = out[:, n_fft // 2 :]
        if audio_length is not None:
            out = out[:, :audio_length]
        return out

    def clear_cache(self):
        self.norm_buffer_cache.clear()
        self.position_cache.clear()


# --------------------------------------------------------------------------- mel front ends
def whisper_log_mel(audio, n_mels: int = 80, padding: int = 0) -> np.ndarray:
    """stt/models/whisper/audio.py:41-82 -> float32 ``[n_frames, n_mels]``."""
    audio = np.asarray(audio, dtype=F32)
    if padding > 0:
        audio = np.concatenate([audio, np.zeros(padding, F32)])
    spec = stft(audio, window=hanning(400), n_fft=400, hop_length=160)
    mags = (np.abs(spec[:-1, :]).astype(F32) ** 2).astype(F32)
    fb = mel_filters(16000, 400, n_mels, norm="slaney", mel_scale=None)
    mel = (mags @ fb.T).astype(F32)
    log_spec = np.log10(np.maximum(mel, F32(1e-10))).astype(F32)
    log_spec = np.maximum(log_spec, log_spec.max() - F32(8.0))
    return ((log_spec + F32(4.0)) / F32(4.0)).astype(F32)


def whisper_log_mel_f64(audio, n_mels: int = 80, padding: int = 0) -> np.ndarray:
    """The same chain evaluated entirely in float64 (window, framing, DFT, power, filterbank, log10, clamp): the value both the reference's
    float32 pipeline and the device kernel approximate.  Used as the yardstick on the speech fixture (tests/golden/*.wav)."""
    audio = np.asarray(audio, dtype=np.float64)
    if padding > 0:
        audio = np.concatenate([audio, np.zeros(padding)])
    n_fft, hop = 400, 160
    w = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n_fft) / (n_fft - 1)))  # audio.py:72: hanning(N_FFT), the symmetric form (dsp.py:40-50)
    x = np.pad(audio, n_fft // 2, mode="reflect")
    n = 1 + (len(x) - n_fft) // hop
    idx = np.arange(n)[:, None] * hop + np.arange(n_fft)[None, :]
    spec = np.fft.rfft(x[idx] * w[None, :], axis=-1)
    mags = np.abs(spec[:-1]) ** 2
    fb = mel_filters(16000, 400, n_mels, norm="slaney", mel_scale=None).astype(np.float64)
    log_spec = np.log10(np.maximum(mags @ fb.T, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0



def qwen3_mel_spectrogram(
    audio,
    n_fft: int = 1024,
    num_mels: int = 128,
    sample_rate: int = 24000,
    hop_size: int = 256,
    win_size: int = 1024,
    fmin: float = 0.0,
    fmax: float = 12000.0,
) -> np.ndarray:
    """tts/models/qwen3_tts/qwen3_tts.py:64-120 -> float32 ``[B, frames, num_mels]``."""
    audio = np.asarray(audio, dtype=F32)
    if audio.ndim == 1:
        audio = audio[None, :]
    fb = mel_filters(sample_rate, n_fft, num_mels, fmin, fmax, norm="slaney", mel_scale="slaney")
    pad = (n_fft - hop_size) // 2
    outs = []
    for s in audio:
        s = np.concatenate([s[1 : pad + 1][::-1], s, s[-(pad + 1) : -1][::-1]])
        spec = stft(s, n_fft=n_fft, hop_length=hop_size, win_length=win_size, window="hann", center=False)
        mag = np.sqrt((np.abs(spec).astype(F32) ** 2 + F32(1e-9)).astype(F32)).astype(F32)
        mel = (mag @ fb.T).astype(F32)
        outs.append(np.log(np.clip(mel, F32(1e-5), None)).astype(F32))
    return np.stack(outs, axis=0)


# ------------------------------------------------------------------------------------------------ Kaldi fbank (dsp.py:806-997)
def mel_scale_kaldi(freq):
    """dsp.py:806-808."""
    return 1127.0 * np.log(1.0 + np.asarray(freq, dtype=np.float32) / np.float32(700.0))


def inverse_mel_scale_kaldi(mel_freq):
    """dsp.py:811-813."""
    return 700.0 * (np.exp(np.asarray(mel_freq, dtype=np.float32) / np.float32(1127.0)) - 1.0)


def get_strided_kaldi(waveform: np.ndarray, window_size: int, window_shift: int, snip_edges: bool) -> np.ndarray:
    """dsp.py:821-843."""
    num_samples = waveform.shape[0]
    if snip_edges:
        if num_samples < window_size:
            return np.zeros((0, 0), np.float32)
        m = 1 + (num_samples - window_size) // window_shift
    else:
        m = (num_samples + (window_shift // 2)) // window_shift
        pad = window_size // 2 - window_shift // 2
        if pad > 0:
            pad_left = waveform[1: pad + 1][::-1]
            pad_right = waveform[-1: -pad - 1: -1] if pad > 1 else waveform[-1:0:-1]
            waveform = np.concatenate([pad_left, waveform, pad_right])
        else:
            pad_right = waveform[::-1]
            waveform = np.concatenate([waveform[-pad:], pad_right])
    if (m - 1) * window_shift + window_size > waveform.shape[0]:
        # the reference's mx.as_strided view would run past the padded buffer here (snip_edges=False and num_samples % shift >= shift / 2:
        # undefined contents); the restatement refuses instead of reading out of bounds, and so does the HIP kernel
        raise ValueError("get_strided_kaldi: frames run past the reflected edges (out-of-bounds read in the reference)")
    waveform = np.ascontiguousarray(waveform)
    return np.lib.stride_tricks.as_strided(waveform, shape=(m, window_size), strides=(window_shift * waveform.itemsize, waveform.itemsize)).copy()


def get_mel_banks_kaldi(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float, high_freq: float):
    """dsp.py:846-895 (float32 like the reference's mx arrays)."""
    f32 = np.float32
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = f32(sample_freq / window_length_padded)
    mel_low, mel_high = float(mel_scale_kaldi(low_freq)), float(mel_scale_kaldi(high_freq))
    delta = f32((mel_high - mel_low) / (num_bins + 1))
    idx = np.arange(num_bins, dtype=f32).reshape(-1, 1)
    left, center, right = f32(mel_low) + idx * delta, f32(mel_low) + (idx + f32(1.0)) * delta, f32(mel_low) + (idx + f32(2.0)) * delta
    mel = mel_scale_kaldi(fft_bin_width * np.arange(num_fft_bins, dtype=f32)).reshape(1, -1)
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    return np.maximum(f32(0.0), np.minimum(up, down)).astype(f32), inverse_mel_scale_kaldi(center).squeeze()


def compute_fbank_kaldi(waveform, sample_rate: int = 48000, win_len: int = 1920, win_inc: int = 384, num_mels: int = 60, win_type: str = "hamming",
                        preemphasis: float = 0.97, dither: float = 1.0, snip_edges: bool = True, low_freq: float = 20.0, high_freq: float = 0.0,
                        noise=None) -> np.ndarray:
    """dsp.py:898-997 statement by statement; ``noise`` [n_frames, window] replaces ``mx.random.normal`` (explicit, reproducible)."""
    x = np.asarray(waveform, dtype=np.float32)
    if x.ndim == 2:
        x = x[0]
    frame_length_ms, frame_shift_ms = win_len / sample_rate * 1000, win_inc / sample_rate * 1000
    shift = int(sample_rate * frame_shift_ms * 0.001)
    win = int(sample_rate * frame_length_ms * 0.001)
    P = 1 if win == 0 else 2 ** (win - 1).bit_length()
    fr = get_strided_kaldi(x, win, shift, snip_edges).astype(np.float32)
    if fr.shape[0] == 0:
        return np.zeros((0, num_mels), np.float32)
    if dither != 0.0:
        fr = fr + (np.asarray(noise, np.float32) if noise is not None else np.random.standard_normal(fr.shape).astype(np.float32)) * np.float32(dither)
    fr = fr - fr.mean(axis=1, keepdims=True)
    if preemphasis != 0.0:
        fr = np.concatenate([fr[:, 0:1], fr[:, 1:] - np.float32(preemphasis) * fr[:, :-1]], axis=1)
    n = np.arange(win, dtype=np.float32)
    if win_type == "hamming":
        w = 0.54 - 0.46 * np.cos(2 * np.pi * n / (win - 1))
    elif win_type == "hanning":
        w = 0.5 - 0.5 * np.cos(2 * np.pi * n / (win - 1))
    elif win_type == "povey":
        w = np.power(0.5 - 0.5 * np.cos(2 * np.pi * n / (win - 1)), 0.85)
    else:
        w = np.ones(win)
    fr = (fr * w.astype(np.float32)).astype(np.float32)
    if P != win:
        fr = np.pad(fr, [(0, 0), (0, P - win)])
    spec = np.abs(np.fft.rfft(fr.astype(np.float64), n=P, axis=1)) ** 2.0
    bins, _ = get_mel_banks_kaldi(num_mels, P, float(sample_rate), low_freq, high_freq)
    bins = np.pad(bins, [(0, 0), (0, 1)])
    return np.log(np.maximum(spec @ bins.T.astype(np.float64), 1e-8)).astype(np.float32)


# ------------------------------------------------------------------------------------------------ log-mel front ends of four more dsp callers
# ("the 40-odd dsp.stft / mel_filters callers get the fused mel kernel").  Each restates the reference function it cites and is pinned
# by tests/test_frontends_cpu.py to what the reference's own source computes over the MLX stand-in (tests/golden/ref_frontends.npz).
def _centre_pad_window(window: np.ndarray, n_fft: int) -> np.ndarray:
    """torch.stft / NeMo convention: a window shorter than n_fft sits in the MIDDLE of the frame (parakeet/audio.py:58-69, sortformer.py:79-84)."""
    if window.shape[0] >= n_fft:
        return window.astype(F32)
    left = (n_fft - window.shape[0]) // 2
    return np.concatenate([np.zeros(left, F32), window.astype(F32), np.zeros(n_fft - window.shape[0] - left, F32)])


def _per_feature_norm(x: np.ndarray, axis: int, eps: float = 1e-5) -> np.ndarray:
    """(x - mean) / (std + eps) along ``axis`` with Bessel's correction (parakeet/audio.py:80-85, sortformer.py:105-112)."""
    mean = x.mean(axis=axis, keepdims=True, dtype=F32)
    n = max(x.shape[axis] - 1, 1)
    var = ((x - mean) ** 2).sum(axis=axis, keepdims=True, dtype=F32) / F32(n)
    return ((x - mean) / (np.sqrt(var) + F32(eps))).astype(F32)


def nemo_log_mel(x, sample_rate=16000, n_fft=512, hop_length=160, win_length=400, n_mels=80, window="hann", preemph=0.97, log_guard=2.0 ** -24) -> np.ndarray:
    """The NeMo FilterbankFeatures chain both Parakeet and Sortformer restate: pre-emphasis (first sample kept), centred zero padding, centre-padded
    window, |X|^2, Slaney mel, ln(mel + guard) -> float32 ``[n_frames, n_mels]``."""
    x = np.asarray(x, dtype=F32)
    if preemph > 0:
        x = np.concatenate([x[:1], x[1:] - F32(preemph) * x[:-1]]).astype(F32)
    fn = {"hann": hanning, "hanning": hanning, "hamming": hamming, "blackman": blackman, "bartlett": bartlett}.get(window, hanning)
    w = _centre_pad_window(fn(win_length), n_fft)
    spec = stft(x, n_fft=n_fft, hop_length=hop_length, win_length=n_fft, window=w, pad_mode="constant")
    power = (np.abs(spec).astype(F32) ** 2).astype(F32)
    fb = mel_filters(sample_rate, n_fft, n_mels, norm="slaney", mel_scale="slaney")
    return np.log((power @ fb.T).astype(F32) + F32(log_guard)).astype(F32)


def parakeet_log_mel(x, sample_rate=16000, normalize="per_feature", window_size=0.025, window_stride=0.01, window="hann", features=80, n_fft=512,
                     pad_to=0, pad_value=0.0, preemph=0.97, log_zero_guard_value=2.0 ** -24) -> np.ndarray:
    """stt/models/parakeet/audio.py:39-94 -> ``[1, n_frames, features]``."""
    x = np.asarray(x, dtype=F32)
    if pad_to > 0 and x.shape[-1] < pad_to:
        x = np.concatenate([x, np.full(pad_to - x.shape[-1], pad_value, F32)])
    y = nemo_log_mel(x, sample_rate, n_fft, int(window_stride * sample_rate), int(window_size * sample_rate), features, window, preemph, log_zero_guard_value)
    if normalize == "per_feature":
        y = _per_feature_norm(y, axis=0)
    else:
        y = ((y - y.mean(dtype=F32)) / (y.std(dtype=F32) + F32(1e-5))).astype(F32)
    return y[None]


def sortformer_mel_features(waveform, sample_rate=16000, n_fft=512, hop_length=160, win_length=400, n_mels=80, preemphasis_coeff=0.97,
                            normalize="per_feature", pad_to=16) -> np.ndarray:
    """vad/models/sortformer/sortformer.py:43-123 -> ``[batch, n_mels, n_frames (padded to a multiple of pad_to with zeros)]``."""
    w = np.asarray(waveform, dtype=F32)
    if w.ndim == 1:
        w = w[None]
    feats = np.stack([nemo_log_mel(r, sample_rate, n_fft, hop_length, win_length, n_mels, "hann", preemphasis_coeff).T for r in w])
    if normalize == "per_feature":
        feats = _per_feature_norm(feats, axis=2)
    if pad_to > 0 and feats.shape[2] % pad_to:
        feats = np.concatenate([feats, np.zeros(feats.shape[:2] + (pad_to - feats.shape[2] % pad_to,), F32)], axis=2)
    return feats


def s3_log_mel(audio, sample_rate=16000, n_mels=128, n_fft=400, hop_length=160, padding=0) -> np.ndarray:
    """codec/models/s3/utils.py:8-42: Whisper's chain with a PERIODIC Hann window and no frame dropped -> ``[n_mels, n_frames]``."""
    audio = np.asarray(audio, dtype=F32)
    if padding > 0:
        audio = np.concatenate([audio, np.zeros(padding, F32)])
    spec = stft(audio, window=hanning(n_fft + 1)[:-1], n_fft=n_fft, hop_length=hop_length, win_length=n_fft)
    mags = (np.abs(spec).astype(F32) ** 2).astype(F32)
    fb = mel_filters(sample_rate, n_fft, n_mels, norm="slaney", mel_scale="slaney")
    log_spec = np.log10(np.maximum((fb @ mags.T).astype(F32), F32(1e-10))).astype(F32)
    log_spec = np.maximum(log_spec, log_spec.max() - F32(8.0))
    return ((log_spec + F32(4.0)) / F32(4.0)).astype(F32)


def voxtral_log_mel(audio, n_mels=128, window_size=400, hop_length=160, sample_rate=16000, global_log_mel_max=1.5) -> np.ndarray:
    """stt/models/voxtral_realtime/audio.py:21-96: periodic Hann, reflect-centred frames, last frame dropped, Slaney filters up to 8 kHz, FIXED maximum
    for the clamp -> ``[n_mels, n_frames - 1]``."""
    audio = np.asarray(audio, dtype=F32)
    n = np.arange(window_size, dtype=F32)
    window = (0.5 * (1.0 - np.cos(2.0 * np.pi * n / window_size))).astype(F32)
    spec = stft(audio, window=window, n_fft=window_size, hop_length=hop_length, win_length=window_size)
    mags = (np.abs(spec[:-1]).astype(F32) ** 2).astype(F32)
    fb = mel_filters(sample_rate, window_size, n_mels, 0, 8000, norm="slaney", mel_scale="slaney")
    log_spec = np.log10(np.maximum((fb @ mags.T).astype(F32), F32(1e-10))).astype(F32)
    log_spec = np.maximum(log_spec, F32(global_log_mel_max - 8.0))
    return ((log_spec + F32(4.0)) / F32(4.0)).astype(F32)
