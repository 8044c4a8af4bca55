"""Qwen3-TTS speech tokenizer behind the reference's interface (``tts/models/qwen3_tts/speech_tokenizer.py:1060-1219``): ``decode`` /
``batch_decode`` / ``streaming_decode`` over the MI355X decoder engine (``codec.Qwen3CodecDecoder``), plus the checkpoint key handling of its
``sanitize`` (``:1220-1449``) for the decoder half.

The ENCODER half (Mimi-style SEANet + transformer + RVQ encode, used only for in-context voice cloning) is not part of this build:
``has_encoder`` is False, ``encode`` raises, and ``sanitize`` drops the ``encoder.*`` keys.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .config import Qwen3TTSTokenizerConfig


def check_array_shape_qwen3(arr) -> bool:
    """True when a 3-D conv weight is already (out, K, in) rather than PyTorch's (out, in, K): the reference's heuristic
    (``qwen3_tts.py:123-157``): a unit axis decides by the size of the other one (> 64 = channels), otherwise the smaller middle axis is K."""
    if len(arr.shape) != 3:
        return False
    _, d2, d3 = arr.shape
    if d2 == 1:
        return d3 > 64
    if d3 == 1:
        return not d2 > 64
    return d2 < d3


class Qwen3TTSSpeechTokenizer:
    def __init__(self, config: Qwen3TTSTokenizerConfig, device="cuda", precision: int = 2):
        self.config = config
        self.encoder_valid_num_quantizers = config.encoder_valid_num_quantizers
        self.input_sample_rate = config.input_sample_rate
        self.output_sample_rate = config.output_sample_rate
        self.decode_upsample_rate = config.decode_upsample_rate
        self.encode_downsample_rate = config.encode_downsample_rate
        self.device = device
        self.precision = precision
        self.decoder = None       # Qwen3CodecDecoder, built by load_weights
        self.encoder_model = None

    @property
    def has_encoder(self) -> bool:
        return False

    def encode(self, audio):
        raise ValueError("Encoder not available for this speech tokenizer (the MI355X build ships the decoder half only)")

    @staticmethod
    def sanitize(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """PyTorch checkpoint -> decoder parameter names / layouts (speech_tokenizer.py:1380-1447): transposed-conv weights (in, out, K) ->
        (out, K, in), conv / 1x1 projection weights (out, in, K) -> (out, K, in) unless the shape heuristic says they already are, decoder
        codebooks materialised as ``embedding_sum / clip(cluster_usage, 1e-5)`` under ``...codebook.embed.weight``.  ``encoder.*`` is dropped."""
        out: Dict[str, torch.Tensor] = {}
        books: Dict[str, Dict[str, torch.Tensor]] = {}
        for k, v in weights.items():
            if k.startswith("encoder."):
                continue
            if "_codebook.cluster_usage" in k or "_codebook.embedding_sum" in k:
                base = k.rsplit("._codebook.", 1)[0]
                books.setdefault(base, {})["cluster_usage" if "cluster_usage" in k else "embedding_sum"] = v
                continue
            is_transpose_conv = ("upsample" in k and ".0.conv.weight" in k) or ("decoder.decoder" in k and "block.1.conv.weight" in k)
            if is_transpose_conv and v.dim() == 3:
                v = v if check_array_shape_qwen3(v) else v.permute(1, 2, 0).contiguous()
            elif ("conv.weight" in k or "_proj.weight" in k) and v.dim() == 3:
                v = v if check_array_shape_qwen3(v) else v.permute(0, 2, 1).contiguous()
            out[k] = v
        for base, d in books.items():
            if "cluster_usage" in d and "embedding_sum" in d:
                out[f"{base}.codebook.embed.weight"] = d["embedding_sum"].float() / d["cluster_usage"].float()[:, None].clamp_min(1e-5)
        return out

    def load_weights(self, weights, strict: bool = False):
        """``weights``: sanitized dict / pair list with the reference's module paths (``decoder.`` prefix); builds the device engine."""
        from .codec import Qwen3CodecDecoder

        w = dict(weights)
        dec = {k[len("decoder."):]: v for k, v in w.items() if k.startswith("decoder.")}
        try:
            self.decoder = Qwen3CodecDecoder(dec, self.config.decoder_config, device=self.device, precision=self.precision)
        except KeyError as e:
            raise ValueError(f"Qwen3-TTS speech tokenizer checkpoint is missing parameter {e}") from e
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ decode
    def decode(self, audio_codes: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio_codes int [B, time, num_quantizers] -> (audio [B, samples], valid sample counts [B]) (speech_tokenizer.py:1100-1118: frames whose
        first code is > 0 count as valid)."""
        codes = audio_codes.permute(0, 2, 1)
        wav = self.decoder.chunked_decode(codes).squeeze(1)
        lengths = (audio_codes[..., 0] > 0).sum(dim=1) * self.decode_upsample_rate
        return wav, lengths

    def batch_decode(self, codes_list: List[torch.Tensor]) -> Tuple[List[torch.Tensor], List[int]]:
        """Variable-length sequences in one batched pass (:1120-1176): zero-padded to the longest, trimmed to ``len_i * upsample`` samples."""
        if not codes_list:
            return [], []
        normed = [c[None] if c.dim() == 2 else c for c in codes_list]
        lens = [int(c.shape[1]) for c in normed]
        m = max(lens)
        batch = torch.cat([torch.nn.functional.pad(c, (0, 0, 0, m - c.shape[1])) for c in normed], dim=0)
        wav = self.decoder.chunked_decode(batch.permute(0, 2, 1)).squeeze(1)
        out_lens = [n * self.decode_upsample_rate for n in lens]
        return [wav[b, :n] if 0 < n < wav.shape[1] else wav[b] for b, n in enumerate(out_lens)], out_lens

    def streaming_decode(self, audio_codes: torch.Tensor, chunk_tokens: int = 100):
        """Yields [B, samples] chunks of ``chunk_tokens`` frames decoded with 25 frames of left context (:1178-1217)."""
        codes = audio_codes.permute(0, 2, 1)
        total, left = codes.shape[-1], 25
        start = 0
        while start < total:
            end = min(start + chunk_tokens, total)
            ctx = left if start - left > 0 else start
            wav = self.decoder(codes[..., start - ctx:end])
            yield wav[..., ctx * self.decode_upsample_rate:].squeeze(1)
            start = end
