"""KittenTTS forward pass on MI355X: Kokoro's engine (``..kokoro.engine``) with KittenTTS's differences.

The reference's KittenTTS (``tts/models/kitten_tts/``) is its Kokoro port re-parameterised and converted from an int8 ONNX export:

  * decoder widths come from the config -- ``max_conv_dim`` / ``decoder_out_dim`` / ``asr_res_dim`` (kitten_tts.py:121-156) instead of Kokoro's
    fixed 1024 / 512 / 64 (kokoro/istftnet.py:948-975); the LSTMs are ``hidden_dim // 2`` wide (the 32 / 64 / 128 / 256 instantiations of
    ``mi355_lstm_bidir``);
  * ALBERT's FFN uses the ONNX tanh-GELU (kitten_tts.py:244-263), the Snake parameters are per-index attributes ``alpha1_0`` ...
    (kitten_tts/istftnet.py:379-384), predicted durations are clipped from below only (kitten_tts.py:398); the harmonic source's coarse phase
    grid has 2F + 1 points instead of Kokoro's 2F, because ``SineGen.upsample_scale`` stays an mx.array there and ``1 / upsample_scale`` reaches
    ``interpolate`` as the float32 0.0033333334 (kitten_tts/istftnet.py:572,595-599) -- found by running the reference's own files
    (tests/golden/make_reference_fixtures.py);
  * the modules listed in ``activation_quant_modules`` see ``fake_quant_dynamic_u8`` of their input (kitten_tts/quant.py): per-tensor min / max ->
    uint8 grid -> back to float.  That needs the tensor's extrema before its first use, so a flagged conv cannot fuse its AdaIN / Snake
    prologue: ``KokoroEngine._convq`` materialises and quantises the input with ``mi355_fake_quant_u8`` and runs the conv without prologue;
    flagged LSTMs quantise their input and, per time step, the hidden vector inside the recurrence kernel (``quant_h``); a flagged AdaIN
    takes its (gamma, beta) from the projection of the quantised style vector; a flagged ``l_linear`` quantises the 9 harmonic terms inside
    the source kernel (``mi355_sine_source.quant_ws``).

Not reproduced: ``mlx_unwrap`` -- KittenTTS dropped it (kitten_tts/istftnet.py:524-527) and for Kokoro it is the identity because the phase
fed to the iSTFT is ``sin(x)`` with magnitude <= 1.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from ....ops import ACT_GELU_TANH
from ..kokoro.engine import KokoroEngine


class KittenEngine(KokoroEngine):
    ffn_act = ACT_GELU_TANH
    alpha_name = "alpha{w}_{i}"
    max_frames = -1  # mx.clip(mx.round(duration), a_min=1, a_max=None)
    coarse_f32 = True  # SineGen.upsample_scale stays an mx.array: 1 / upsample_scale is a float32 -> the coarse phase grid has 2F + 1 points

    def _decoder_dims(self, config: dict):
        cd = int(config["max_conv_dim"])
        gd = int(config.get("decoder_out_dim") or cd)
        if gd != int(config["istftnet"]["upsample_initial_channel"]):
            raise ValueError(f"KittenTTS: decoder_out_dim {gd} must equal istftnet.upsample_initial_channel "
                             f"{config['istftnet']['upsample_initial_channel']} (the generator's input width, kitten_tts.py:139-156)")
        return cd, gd, int(config["asr_res_dim"])

    def __init__(self, weights: Dict[str, torch.Tensor], config: dict, device="cuda", param_dtype=torch.float32, precision: int = None,
                 quant_modules: Sequence[str] = None):
        if precision is None:
            # float32 checkpoints (what the ONNX converter writes): fp16 weight images + fp16 hi / lo activations; 16-bit checkpoints are exact in mode 2
            precision = 4 if param_dtype == torch.float32 else 2
        pb = config["plbert"]
        if int(pb.get("num_hidden_groups", 1)) != 1 or int(pb.get("inner_group_num", 1)) != 1:
            raise NotImplementedError("KittenTTS engine: ALBERT with more than one layer group / inner layer is not supported")
        if int(config["style_dim"]) != 128:
            # the reference splits the voice row at the literal column 128 (kitten_tts.py:392,411)
            raise ValueError(f"KittenTTS: style_dim must be 128 (the voice row is split at column 128), got {config['style_dim']}")
        if quant_modules is None:
            quant_modules = config.get("activation_quant_modules") or ()
        super().__init__(weights, config, device=device, param_dtype=param_dtype, precision=precision, quant_modules=tuple(quant_modules))
