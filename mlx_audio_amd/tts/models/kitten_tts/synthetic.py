"""Seeded synthetic KittenTTS checkpoints (no network => no real weights): the reference's post-``sanitize`` parameter names and the shapes
its ``Model(ModelConfig)`` constructs (``tts/models/kitten_tts/kitten_tts.py:94-174,251-289``)."""
from __future__ import annotations

from typing import Dict, List

import torch

from ..kokoro.synthetic import make_kokoro_weights

# a "nano"-like shape: everything KittenTTS reads from its config differs from Kokoro's constants
KITTEN_CONFIG = {
    "model_type": "kitten_tts",
    "hidden_dim": 128,
    "max_conv_dim": 256,
    "max_dur": 50,
    "n_layer": 3,
    "n_mels": 80,
    "n_token": 178,
    "style_dim": 128,
    "text_encoder_kernel_size": 5,
    "asr_res_dim": 64,
    "decoder_out_dim": 256,
    "plbert": {
        "num_hidden_layers": 4,
        "num_attention_heads": 4,
        "hidden_size": 256,
        "intermediate_size": 512,
        "max_position_embeddings": 512,
        "embedding_size": 128,
        "inner_group_num": 1,
        "num_hidden_groups": 1,
        "hidden_dropout_prob": 0.0,
        "attention_probs_dropout_prob": 0.0,
        "type_vocab_size": 2,
        "layer_norm_eps": 1e-12,
    },
    "istftnet": {
        "resblock_kernel_sizes": [3, 7, 11],
        "upsample_rates": [10, 6],
        "upsample_initial_channel": 256,
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "upsample_kernel_sizes": [20, 12],
        "gen_istft_n_fft": 20,
        "gen_istft_hop_size": 5,
    },
    "sample_rate": 24000,
    "voices_path": "voices.npz",
}


def tiny_config() -> dict:
    """Small enough for the CPU oracle to finish in seconds; odd widths on purpose (asr_res_dim 24, decoder widths 96 / 64)."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in KITTEN_CONFIG.items()}
    cfg.update(hidden_dim=64, max_conv_dim=96, asr_res_dim=24, decoder_out_dim=64, n_layer=2)
    cfg["plbert"] = dict(cfg["plbert"], num_hidden_layers=2, num_attention_heads=2, hidden_size=64, intermediate_size=128,
                         max_position_embeddings=64, embedding_size=32)
    cfg["istftnet"] = dict(cfg["istftnet"], upsample_initial_channel=64)
    return cfg


def make_kitten_weights(config: dict = None, seed: int = 0) -> Dict[str, torch.Tensor]:
    cfg = config or KITTEN_CONFIG
    dims = (cfg["max_conv_dim"], cfg.get("decoder_out_dim") or cfg["max_conv_dim"], cfg["asr_res_dim"])
    return make_kokoro_weights(cfg, seed, decoder_dims=dims, alpha_sep="_")


def converter_quant_modules(weights: Dict[str, torch.Tensor]) -> List[str]:
    """The module list the reference's converter writes for a fully int8 ONNX export (kitten_tts/convert.py:401-436): every module that owns a
    quantised weight -- all weight-normed convs, all linear layers (AdaIN ``fc`` included), the F0 / N projections and noise convs -- plus the
    six LSTMs."""
    mods = set()
    for k, v in weights.items():
        if "embeddings." in k or k.endswith("embedding.weight"):
            continue
        if k.endswith(".weight_v") or (k.endswith(".weight") and v.dim() in (2, 3)) or k.endswith(".Wx_forward"):
            mods.add(k.rsplit(".", 1)[0])
    return sorted(mods)
