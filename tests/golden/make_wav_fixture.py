"""Generates ``tests/golden/genesis_1_1_af_heart_16k.wav`` from a speech clip the reference ships
(``examples/bible-audiobook/audios/bible-akjv/af_heart/00000001-Genesis-1:1.wav``, 24 kHz int16 mono, 6.6 s): polyphase-resampled to
Whisper's 16 kHz and stored as int16 PCM (105600 samples, 206 KiB).  Run in the build container (``/root/reference`` is not on the GPU box);
the parity tests read only the committed file.
"""
import os

import numpy as np
import scipy.io.wavfile as wavfile
from scipy.signal import resample_poly

SRC = "/root/reference/examples/bible-audiobook/audios/bible-akjv/af_heart/00000001-Genesis-1:1.wav"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genesis_1_1_af_heart_16k.wav")

if __name__ == "__main__":
    sr, x = wavfile.read(SRC)
    assert sr == 24000 and x.dtype == np.int16 and x.ndim == 1
    y = resample_poly(x.astype(np.float64), 2, 3)
    y = np.clip(np.round(y), -32768, 32767).astype(np.int16)
    wavfile.write(DST, 16000, y)
    print(DST, y.shape, float(np.abs(y).max()))
