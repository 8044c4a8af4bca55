"""Writes tests/golden/reference_vectors.json.

The reference (mlx-audio) cannot be imported here (``import mlx`` fails, no wheel,
no network), so these are NOT regenerated from the reference: they are the
known-answer vectors *transcribed* from the reference's own tests, each with the
file:line it was read from.  Re-running this script only re-serialises the
literals below; it never touches /root/reference.
"""
import json
import os

VECTORS = {
    "qwen3_mel_spectrogram": {
        "source": "mlx_audio/tts/tests/test_qwen3_tts.py:175-353",
        "input": "np.random.seed(42); np.random.randn(12000).astype(np.float32)",
        "shape": [1, 46, 128],
        "rtol": 2e-3,
        "atol": 2e-3,
        "bins": [0, 1, 2, 63, 126, 127],
        "frame0": [-0.21803714, 0.06630915, -0.31858957, -0.02480409, -0.4512914, -0.5911693],
        "frame23": [0.08127937, 0.4368576, 0.43200976, -0.7714137, -0.24601418, 0.04274124],
        "frame_last": [-0.16861804, 0.0474052, -0.3970174, -0.01738772, -0.28846806, -0.10941511],
        "mean": -0.37329558,
        "std": 0.37445435,
        "sine_1khz": {
            "input": "sin(2*pi*1000*arange(12000)/24000) float32",
            "bins": [0, 1, 2, 10, 20, 63, 126, 127],
            "frame0": [-1.2959518, -1.2937515, -1.2902284, -1.2074544, -0.9268621, -2.3822036, -5.331841, -5.33782],
        },
    },
    "conv_transpose_weight_norm": {
        "source": "mlx_audio/tts/tests/test_istftnet_fidelity.py:18-31",
        "weight_v": [1.0, 2.0, 3.0],
        "weight_g_squared": 14.0,
        "x": [1.0, 2.0, 3.0, 4.0],
        "stride": 2,
        "padding": 0,
        "drop_first": 1,
        "expected": [2.0, 5.0, 4.0, 9.0, 6.0, 13.0, 8.0, 12.0],
        "rtol": 1e-4,
    },
    "mlxstft_roundtrip": {
        "source": "mlx_audio/tts/tests/test_istftnet_fidelity.py:34-46",
        "n_fft": 20, "hop": 5, "length": 2000, "freq_hz": 220, "sr": 24000, "amp": 0.5,
        "atol": 1e-3, "edge": 20,
    },
    "interpolate": {
        "source": "mlx_audio/tts/tests/test_interpolate.py:40-97",
        "nearest_in": [1.0, 2.0, 3.0, 4.0],
        "nearest_up8": [1.0, 1.0, 2.0, 2.0, 3.0, 3.0, 4.0, 4.0],
        "nearest_down2": [1.0, 3.0],
        "linear_in": [1.0, 3.0, 5.0, 7.0],
        "linear_ac_true_7": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0],
        "linear_ac_false_7": [1.0, 1.7142857, 2.8571429, 4.0, 5.1428576, 6.2857141, 7.0],
        "rtol": 1e-5,
    },
    "sinegen_shapes": {
        "source": "mlx_audio/tts/tests/test_sinegen_length_alignment.py:8-17",
        "upsample_scale": 300, "harmonic_num": 8, "f0": 120.0, "length": 2,
        "sine_shape": [1, 2, 9],
    },
    "istft_cache_bound": {
        "source": "mlx_audio/tests/test_dsp.py:62-95",
        "rng": "np.random.default_rng(0): real then imag, normal(size=(2,9,8)).astype(float32)*8",
        "n_fft": 16, "hop": 4, "min_diff": 0.1, "bound": 1.000001,
    },
    "whisper_greedy_update": {
        "source": "mlx_audio/stt/tests/test_whisper_decoding.py:101-119",
        "eot": 99,
        "tokens": [[1, 2], [1, 2], [1, 2]],
        "logits": [[0.0, 1.0, 2.0], [3.0, 1.0, 0.0], [0.0, 5.0, 1.0]],
        "expected_tokens": [[1, 2, 2], [1, 2, 0], [1, 2, 1]],
        "completed": False,
    },
    "whisper_timestamp_rules_shape": {
        "source": "mlx_audio/stt/tests/test_whisper_decoding.py:122-136",
        "timestamp_begin": 4, "no_timestamps": None, "sample_begin": 2, "max_initial_timestamp_index": None,
        "logits_shape": [3, 8], "tokens": [[1, 2], [1, 2], [1, 2]],
    },
    "whisper_dims": {
        "source": "mlx_audio/stt/tests/test_models.py:43-54 (whisper-small), stt/models/whisper/audio.py:12-24",
        "n_mels": 80, "n_audio_ctx": 1500, "n_audio_state": 768, "n_audio_head": 12, "n_audio_layer": 12, "n_vocab": 51865,
        "n_text_ctx": 448, "n_text_state": 768, "n_text_head": 12, "n_text_layer": 12,
        "n_samples": 480000, "n_frames": 3000, "hop": 160, "n_fft": 400,
    },
    "kokoro_shapes": {
        "source": "SURVEY.md 8 header; examples/bible-audiobook/audios/*: every length is k*600 samples",
        "samples_per_frame": 600,
    },
}

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(out, "w") as f:
        json.dump(VECTORS, f, indent=1)
    print("wrote", out)
