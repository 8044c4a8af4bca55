"""Audio file I/O at the edges of the hot path: ``read`` / ``write`` / ``sf_read`` / ``sf_write`` with the signatures of ``mlx_audio/audio_io.py``
(``:232-248, 486-491, 605-639``), for the container this build handles itself: RIFF / WAVE (PCM 8 / 16 / 24 / 32 bit, IEEE float 32 / 64,
``WAVE_FORMAT_EXTENSIBLE``) and headerless PCM16 on the write side.

Conventions kept from the reference: decoded audio passes through signed 16-bit samples (the reference decodes with miniaudio to s16,
``audio_io.py:339-341``), ``dtype='float32' | 'float64'`` divides by 32768 (``:349-352``), mono comes back 1-D unless ``always_2d`` (``:343-368``),
``write`` clips floats to [-1, 1] and scales by 32767 with truncation (``:526-531``), integer input other than int16 is cast, the format is taken
from the extension (WAV for a ``BytesIO``), unknown formats raise ``ValueError`` (``:601-602``).  The compressed containers (mp3 / flac / ogg / opus /
webm / m4a) go through **ffmpeg when one is on PATH**, with the reference's own command lines (ffprobe for rate / channels, ``-f s16le -acodec pcm_s16le``
to decode, ``audio_io.py:59-187``; raw s16le in, ``-b:a`` / libopus / FLAC-in-Ogg out to encode, ``:405-483``); the reference's second decoder
(miniaudio, for mp3 / flac / vorbis without ffmpeg) is not part of this build, so without ffmpeg those raise ``RuntimeError`` naming the gap.  A
``sample_rate`` different from the file's goes through ``mlx_audio_amd.resample`` (the reference's ``kaiser_best`` polyphase FIR; the reference uses
miniaudio's converter for upsampling and this FIR for downsampling, ``audio_io.py:318-330``: here one filter serves both directions).
"""
from __future__ import annotations

import io
import json
import shutil
import struct
import subprocess
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

_COMPRESSED = ("flac", "mp3", "ogg", "opus", "vorbis", "webm", "m4a", "aac")
FileLike = Union[str, Path, io.BytesIO]


def _sniff_compressed(buf: bytes) -> Optional[str]:
    """Container of a byte stream by its magic bytes (audio_io.py:39-56), ``None`` for anything else."""
    if buf[:4] == b"OggS":
        return "Ogg"
    if buf[4:8] == b"ftyp":
        return "MP4 / M4A"
    if buf[:4] == b"fLaC":
        return "FLAC"
    if buf[:3] == b"ID3" or buf[:2] in (b"\xff\xfb", b"\xff\xfa"):
        return "MP3"
    if buf[:4] == b"\x1a\x45\xdf\xa3":
        return "WebM"
    return None


def _decode_ffmpeg(src: Union[str, Path, bytes], sample_rate: Optional[int], nchannels: Optional[int], kind: str) -> Tuple[np.ndarray, int, int]:
    """Compressed container -> (int16 samples [frames, channels], rate, channels) through ffprobe + ffmpeg, the reference's way (audio_io.py:59-187):
    the stream's own rate / channel count unless the caller asks for others (ffmpeg then converts)."""
    ffmpeg, ffprobe = shutil.which("ffmpeg"), shutil.which("ffprobe")
    if ffmpeg is None or ffprobe is None:
        raise RuntimeError(f"{kind} decoding needs ffmpeg / ffprobe on PATH (the reference's miniaudio decoder is not part of this build); convert to WAV first")
    from_bytes = isinstance(src, (bytes, bytearray))
    target = "pipe:0" if from_bytes else str(src)
    probe = subprocess.run([ffprobe, "-v", "quiet", "-print_format", "json", "-show_streams", "-select_streams", "a:0"] + (["-i", target] if from_bytes else [target]),
                           input=bytes(src) if from_bytes else None, capture_output=True)
    if probe.returncode != 0:
        raise RuntimeError(f"ffprobe failed: {probe.stderr.decode(errors='replace')}")
    streams = json.loads(probe.stdout.decode() or "{}").get("streams")
    if not streams:
        raise RuntimeError("No audio streams found in file")
    rate = int(sample_rate or streams[0].get("sample_rate", 44100))
    nch = int(nchannels or streams[0].get("channels", 2))
    dec = subprocess.run([ffmpeg, "-i", target, "-f", "s16le", "-acodec", "pcm_s16le", "-ar", str(rate), "-ac", str(nch), "pipe:1"],
                         input=bytes(src) if from_bytes else None, capture_output=True)
    if dec.returncode != 0:
        raise RuntimeError(f"ffmpeg decoding failed: {dec.stderr.decode(errors='replace')}")
    pcm = np.frombuffer(dec.stdout, dtype="<i2")
    return pcm[: pcm.size - pcm.size % nch].reshape(-1, nch), rate, nch


def _encode_ffmpeg(pcm: np.ndarray, samplerate: int, nch: int, fmt: str, bitrate: str = "128k") -> bytes:
    """int16 PCM -> compressed bytes through ffmpeg with the reference's options (audio_io.py:405-483): raw s16le in; mp3 at ``bitrate``; opus / webm on
    libopus; ogg / vorbis as FLAC inside an Ogg container."""
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise RuntimeError(f"{fmt.upper()} encoding needs ffmpeg on PATH; use WAV")
    cmd = [ffmpeg, "-y", "-f", "s16le", "-ar", str(int(samplerate)), "-ac", str(int(nch)), "-i", "pipe:0"]
    if fmt == "mp3":
        cmd += ["-b:a", bitrate]
    elif fmt in ("opus", "webm"):
        cmd += ["-c:a", "libopus", "-b:a", bitrate]
    elif fmt in ("ogg", "vorbis"):
        cmd += ["-c:a", "flac"]
    if fmt in ("m4a", "aac"):   # the MP4-family muxer wants seekable output; on a pipe it needs fragmented output with the moov atom up front
        cmd += ["-movflags", "frag_keyframe+empty_moov"]
    cmd += ["-f", "ogg" if fmt == "vorbis" else ("ipod" if fmt in ("m4a", "aac") else fmt), "pipe:1"]
    r = subprocess.run(cmd, input=np.ascontiguousarray(pcm).astype("<i2").tobytes(), capture_output=True)
    if r.returncode != 0:
        raise RuntimeError(f"ffmpeg encoding failed: {r.stderr.decode(errors='replace')}")
    return r.stdout


def _riff_chunks(buf: bytes):
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError("Unsupported format: not a RIFF / WAVE stream")
    pos = 12
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        yield cid, buf[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)


def _decode_wav(buf: bytes) -> Tuple[np.ndarray, int, int]:
    """-> (int16 samples [frames, channels], sample rate, channels)."""
    fmt = data = None
    for cid, body in _riff_chunks(buf):
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError("malformed WAV: missing fmt / data chunk")
    tag, nch, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:   # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes are the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    if nch <= 0 or rate <= 0:
        raise ValueError("malformed WAV: bad channel count / sample rate")
    if tag == 1:
        if bits == 16:
            s = np.frombuffer(data[: len(data) // 2 * 2], dtype="<i2").astype(np.int16)
        elif bits == 8:
            s = ((np.frombuffer(data, dtype=np.uint8).astype(np.int16) - 128) << 8).astype(np.int16)
        elif bits == 24:
            b = np.frombuffer(data[: len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3)
            s = ((b[:, 1].astype(np.int32) | (b[:, 2].astype(np.int32) << 8)).astype(np.uint16)).view(np.int16)   # top 16 of 24 bits
        elif bits == 32:
            s = (np.frombuffer(data[: len(data) // 4 * 4], dtype="<i4") >> 16).astype(np.int16)
        else:
            raise ValueError(f"unsupported PCM width: {bits} bits")
    elif tag == 3:
        f = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
        s = np.clip(np.rint(f * 32768.0), -32768, 32767).astype(np.int16)
    else:
        raise RuntimeError(f"WAV codec tag {tag:#x} (compressed) needs ffmpeg / miniaudio, which this build does not contain")
    frames = s.size // nch
    return s[: frames * nch].reshape(frames, nch), rate, nch


def read(file: FileLike, always_2d: bool = False, dtype: str = "float64", sample_rate: Optional[int] = None, nchannels: Optional[int] = None):
    """Reads a WAV file / buffer -> (samples, sample_rate); see the module docstring for the conventions."""
    if sample_rate is not None and sample_rate <= 0:
        raise ValueError(f"sample_rate must be positive, got {sample_rate}")
    if nchannels is not None and nchannels <= 0:
        raise ValueError(f"nchannels must be positive, got {nchannels}")
    decoded = None
    if isinstance(file, (str, Path)):
        ext = Path(file).suffix.lstrip(".").lower()
        if ext in _COMPRESSED:
            decoded = _decode_ffmpeg(file, sample_rate, nchannels, ext.upper())
        else:
            with open(file, "rb") as f:
                buf = f.read()
    elif isinstance(file, io.BytesIO):
        file.seek(0)
        buf = file.read()
        file.seek(0)
    else:
        raise TypeError(f"Unsupported file type: {type(file)}")
    if decoded is None:
        kind = _sniff_compressed(buf)
        decoded = _decode_ffmpeg(buf, sample_rate, nchannels, kind) if kind else _decode_wav(buf)
    pcm, rate, nch = decoded
    x = pcm.astype(np.float64) / 32768.0
    if nchannels is not None and nchannels != nch:
        if nchannels == 1:
            x = x.mean(axis=1, keepdims=True)
        elif nch == 1:
            x = np.repeat(x, nchannels, axis=1)
        else:
            raise ValueError(f"cannot convert {nch} channels to {nchannels}")
        nch = nchannels
    if sample_rate is not None and sample_rate != rate:
        from .resample import resample_audio_array

        x = resample_audio_array(x, rate, int(sample_rate), axis=0).astype(np.float64)   # the reference's kaiser_best polyphase FIR (resample.py)
        rate = int(sample_rate)
    if dtype in ("float32", "float64"):
        out = x.astype(dtype)
    else:
        out = np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16).astype(dtype)
    if nch == 1 and not always_2d:
        out = out[:, 0]
    return out, rate


def _to_int16(data) -> np.ndarray:
    if not isinstance(data, np.ndarray):
        if hasattr(data, "detach"):   # torch tensor (the engines return device tensors)
            data = data.detach().cpu().numpy()
        else:
            data = np.asarray(data)
    if data.dtype == np.float16:
        data = data.astype(np.float32)
    if data.dtype in (np.float32, np.float64):   # in the array's own precision, like the reference (a float32 product truncates differently from a float64 one)
        return (np.clip(data, -1.0, 1.0) * 32767).astype(np.int16)
    return data if data.dtype == np.int16 else data.astype(np.int16)


def write(file: FileLike, data, samplerate: int, format: Optional[str] = None) -> None:
    """Writes 16-bit PCM WAV (or headerless ``pcm`` / ``raw``).  ``data``: ``(samples,)`` or ``(samples, channels)``; numpy array or torch tensor."""
    if format is None:
        format = Path(file).suffix.lstrip(".").lower() if isinstance(file, (str, Path)) else "wav"
    format = format.lower()
    pcm = _to_int16(data)
    nch = 1 if pcm.ndim == 1 else int(pcm.shape[1])
    payload = np.ascontiguousarray(pcm).astype("<i2").tobytes()
    if format in ("pcm", "raw"):
        blob = payload
    elif format == "wav":
        hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, 1, nch, int(samplerate), int(samplerate) * nch * 2, nch * 2, 16,
                          b"data", len(payload))
        blob = hdr + payload
    elif format in _COMPRESSED:
        blob = _encode_ffmpeg(pcm, samplerate, nch, format)
    else:
        raise ValueError(f"Unsupported output format: {format}")
    if isinstance(file, io.BytesIO):
        file.write(blob)
        file.seek(0)
    else:
        with open(file, "wb") as f:
            f.write(blob)


def sf_read(file: FileLike, always_2d: bool = False):
    """soundfile-style alias (audio_io.py:605-620)."""
    return read(file, always_2d=always_2d, dtype="float64")


def sf_write(file: FileLike, data, samplerate: int, format: Optional[str] = None) -> None:
    """soundfile-style alias (audio_io.py:623-639)."""
    write(file, data, samplerate, format=format)
