"""Sample-rate conversion at the edge of the hot path: ``resample_audio_array`` (``mlx_audio/resample.py:29-47``) and ``utils.resample_audio``
(``mlx_audio/utils.py:541-578``) with the reference's filter, on the host.

The reference's resampler is host code too (scipy): a polyphase FIR whose anti-aliasing filter is the resampy ``kaiser_best`` design -- a Kaiser-windowed
sinc with 64 zero crossings per side, roll-off 0.9475937167399596 and beta 14.769656459379492, designed at the upsampled rate with its cut-off at
``rolloff / max(up, down)`` of that Nyquist (``resample.py:15-26``) -- applied by ``scipy.signal.resample_poly`` with edge padding (``:40-47``).  The
design constants ARE the algorithm; they are restated here and the properties the reference's tests pin (energy above the new Nyquist removed,
pass band at unit gain, length, identity at equal rates: ``mlx_audio/tests/test_dsp.py:299-349``) are asserted in ``tests/test_audio_io_cpu.py``.
``resample_audio_chunks`` (``resample.py:50-161``) converts a stream of time-first chunks block by block with the SAME samples as one whole-buffer call
(asserted bit-exactly, like ``test_dsp.py:351-380``).

Tensors that already live on the GPU are converted there: ``resample_audio`` hands a CUDA tensor to ``mi355_resample_poly`` (the same
taps as a float64 table split by phase, float64 sums, edge padding: ``polyphase_table`` / ``resample_on_device``), so audio between two device stages never
crosses PCIe; numpy input keeps the reference's host path.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

KAISER_BEST_ZEROS = 64
KAISER_BEST_ROLLOFF = 0.9475937167399596
KAISER_BEST_BETA = 14.769656459379492


@lru_cache(maxsize=32)
def polyphase_design(orig_sample_rate: int, sample_rate: int):
    """(up, down, FIR taps) of the conversion ``orig_sample_rate`` -> ``sample_rate``."""
    from scipy import signal

    g = math.gcd(int(orig_sample_rate), int(sample_rate))
    up, down = int(sample_rate) // g, int(orig_sample_rate) // g
    widest = max(up, down)
    taps = signal.firwin(2 * KAISER_BEST_ZEROS * widest + 1, KAISER_BEST_ROLLOFF / widest, window=("kaiser", KAISER_BEST_BETA))
    return up, down, taps


def polyphase_table(orig_sample_rate: int, sample_rate: int, n_in: int):
    """What ``scipy.signal.resample_poly(x, up, down, window=taps, padtype="edge")`` does to ``n_in`` samples, as data for ``mi355_resample_poly``:
    (up, down, table [K, up] float64 -- tap-major --, first, n_out).  resample_poly scales the taps by ``up``, prepends ``down - half % down`` zeros so that output
    ``first = (half + zeros) // down`` is the one centred on input 0, appends zeros until the filter output is long enough, runs upfirdn and keeps
    outputs ``first .. first + n_out``; output m of upfirdn only meets the taps of phase ``m * down % up``: ``table[k][p] = padded[p + k * up]``."""
    up, down, taps = polyphase_design(int(orig_sample_rate), int(sample_rate))
    half = (len(taps) - 1) // 2
    n_out = -(-n_in * up // down)
    lead = down - half % down
    first = (half + lead) // down
    tail = 0
    while ((n_in - 1) * up + len(taps) + lead + tail - 1) // down + 1 < n_out + first:
        tail += 1
    padded = np.concatenate((np.zeros(lead), np.asarray(taps, dtype=np.float64) * up, np.zeros(tail)))
    K = -(-len(padded) // up)
    table = np.zeros(K * up, dtype=np.float64)
    table[:len(padded)] = padded
    return up, down, table.reshape(K, up), first, n_out


_DEVICE_TABLES: dict = {}


def resample_on_device(audio, orig_sample_rate: int, sample_rate: int, axis: int = -1):
    """CUDA tensor in -> CUDA float32 tensor out through the HIP polyphase kernel (no host copy); same samples as ``resample_audio_array``."""
    import torch

    from . import ops

    if not audio.is_cuda:
        raise ValueError("resample_on_device expects a tensor on the GPU; use resample_audio_array for host arrays")
    ops.require_gpu()
    if orig_sample_rate == sample_rate:
        return audio
    x = audio.to(torch.float32).movedim(axis, -1)
    lead_shape, n_in = x.shape[:-1], x.shape[-1]
    if n_in == 0:
        raise ValueError("cannot resample an empty signal")
    up, down, table, first, n_out = polyphase_table(orig_sample_rate, sample_rate, n_in)
    key = (up, down, table.shape[0], first, str(audio.device))
    if key not in _DEVICE_TABLES:
        _DEVICE_TABLES[key] = torch.from_numpy(table).to(audio.device)
    rows = x.reshape(-1, n_in).contiguous()
    out = torch.empty(rows.shape[0], n_out, device=audio.device, dtype=torch.float32)
    for r0 in range(0, rows.shape[0], 65535):
        ops.resample_poly(rows[r0:r0 + 65535], _DEVICE_TABLES[key], up, down, first, n_out, out[r0:r0 + 65535])
    return out.reshape(*lead_shape, n_out).movedim(-1, axis)


def resample_audio_array(audio: np.ndarray, orig_sample_rate: int, sample_rate: int, axis: int = -1) -> np.ndarray:
    """In-memory array through the polyphase FIR; float32 out, the input itself when the rates are equal."""
    if orig_sample_rate == sample_rate:
        return audio
    from scipy import signal

    up, down, taps = polyphase_design(int(orig_sample_rate), int(sample_rate))
    return signal.resample_poly(np.asarray(audio), up, down, axis=axis, window=taps, padtype="edge").astype(np.float32, copy=False)


def resample_audio(audio, orig_sample_rate: int, sample_rate: int, axis: int = -1):
    """``mlx_audio.utils.resample_audio``: numpy in -> numpy out, torch tensor in -> torch tensor out (same device), identity at equal rates."""
    if orig_sample_rate == sample_rate:
        return audio
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(audio, torch.Tensor) and audio.is_cuda:
        return resample_on_device(audio, orig_sample_rate, sample_rate, axis=axis)
    if torch is not None and isinstance(audio, torch.Tensor):
        out = resample_audio_array(audio.detach().cpu().numpy(), orig_sample_rate, sample_rate, axis=axis)
        return torch.from_numpy(np.ascontiguousarray(out)).to(audio.device)
    return resample_audio_array(np.asarray(audio), orig_sample_rate, sample_rate, axis=axis)


class _FrameWindow:
    """Rolling view over a stream of time-first chunks: ``take(a, b)`` returns frames [a, b) (clipped to what the stream holds), dropping everything
    before the smallest start it will be asked for again."""

    def __init__(self, chunks, first: np.ndarray):
        self._it = iter(chunks)
        self._buf = first
        self._origin = 0
        self._eof = False
        self.trailing = first.shape[1:]

    def _grow_to(self, end: int) -> None:
        while not self._eof and self._origin + self._buf.shape[0] < end:
            try:
                part = np.asarray(next(self._it), dtype=np.float32)
            except StopIteration:
                self._eof = True
                return
            if part.shape[0] == 0:
                continue
            if part.shape[1:] != self.trailing:
                raise ValueError("all audio chunks must have matching shapes")
            self._buf = np.concatenate((self._buf, part), axis=0)

    def available(self, end: int) -> int:
        self._grow_to(end)
        return self._origin + self._buf.shape[0]

    def take(self, a: int, b: int) -> np.ndarray:
        return self._buf[a - self._origin:b - self._origin]

    def forget_before(self, a: int) -> None:
        if a > self._origin:
            self._buf = self._buf[a - self._origin:].copy()
            self._origin = a


def resample_audio_chunks(chunks, orig_sample_rate: int, sample_rate: int, num_input_frames: int, chunk_duration_seconds: float = 1.0) -> np.ndarray:
    """Converts time-first audio chunks without materialising the whole input; the result equals ``resample_audio_array(whole, ..., axis=0)`` sample for
    sample.  The stream is cut into core blocks whose boundaries are multiples of ``down`` input frames (so a block's first output has the same polyphase
    phase as in the whole-buffer call: ``start * up / down`` is an integer), each block is converted together with a halo that covers the filter's support
    on both sides, and only the block's own outputs are kept; the edge padding of ``resample_poly`` is therefore only ever seen at the true ends."""
    if chunk_duration_seconds <= 0:
        raise ValueError("chunk_duration_seconds must be positive")
    it = iter(chunks)
    first = None
    for c in it:
        c = np.asarray(c, dtype=np.float32)
        if c.shape[0] > 0:
            first = c
            break
    if first is None:
        return np.empty((0,), dtype=np.float32)
    total = max(0, int(num_input_frames))
    if total == 0:
        return np.empty((0, *first.shape[1:]), dtype=np.float32)
    if orig_sample_rate == sample_rate:
        return np.concatenate([first] + [np.asarray(c, dtype=np.float32) for c in it], axis=0)[:total]
    from scipy import signal

    up, down, taps = polyphase_design(int(orig_sample_rate), int(sample_rate))
    block = max(down, int(chunk_duration_seconds * orig_sample_rate) // down * down)
    reach = math.ceil(((len(taps) - 1) // 2 + down) / up)          # input frames one output can see on either side (+ the centring shift)
    halo = math.ceil(reach / down) * down
    n_out = math.ceil(total * up / down)
    out = np.empty((n_out, *first.shape[1:]), dtype=np.float32)
    win = _FrameWindow(it, first)
    start, written = 0, 0
    while start < total:
        stop = min(total, start + block)
        have = win.available(min(total, stop + halo))
        if have < stop:                 # the stream ended before num_input_frames
            stop = have
            if stop <= start:
                break
        lo, hi = max(0, start - halo), min(min(total, stop + halo), have)
        seg = signal.resample_poly(win.take(lo, hi), up, down, axis=0, window=taps, padtype="edge").astype(np.float32, copy=False)
        o0, o1 = start * up // down, min(n_out, math.ceil(stop * up / down))
        shift = o0 - lo * up // down
        out[o0:o1] = seg[shift:shift + (o1 - o0)]
        written = o1
        start = stop
        win.forget_before(max(0, start - halo))
    return out[:written]
