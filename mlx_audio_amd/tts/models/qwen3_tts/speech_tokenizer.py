"""Qwen3-TTS speech tokenizer behind the reference's interface (``tts/models/qwen3_tts/speech_tokenizer.py:1060-1219``): ``decode`` /
``batch_decode`` / ``streaming_decode`` over the MI355X decoder engine (``codec.Qwen3CodecDecoder``), plus the checkpoint key handling of its
``sanitize`` (``:1220-1449``) for the decoder half.

The ENCODER half (``Qwen3TTSSpeechTokenizerEncoder``, ``:957-1058``: the Mimi modules -- SeanetEncoder, ProjectedTransformer, ConvDownsample1d,
SplitResidualVectorQuantizer -- under the tokenizer's own configuration: non-traditional RoPE, an explicit causal mask instead of the context
window, the first 16 of 32 codebooks) runs on ``codec.models.mimi.MimiEncoder`` when the checkpoint carries it (round 3): ``sanitize`` maps the
HuggingFace encoder keys (``:1236-1381, 1418-1441``), ``has_encoder`` / ``encode`` follow the reference (``:1083-1098``).
"""
from __future__ import annotations

import re
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
        return self.encoder_model is not None

    def encode(self, audio):
        """audio [B, 1, samples] -> codes int64 [B, 16, ceil(samples / 1920)] (``:1083-1095``, ``:1037-1058``)."""
        if self.encoder_model is None:
            raise ValueError("Encoder not available for this speech tokenizer")
        codes = self.encoder_model(torch.as_tensor(audio))
        return codes[:, : self.encoder_valid_num_quantizers, :]

    @staticmethod
    def encoder_mimi_config(ec: Dict):
        """The reference's ``Qwen3TTSTokenizerEncoderConfig`` (config.py:139-174; raw dict here) as the Mimi engine's configuration (``:963-1035``)."""
        from ....codec.models.mimi.mimi import MimiConfig

        g = lambda k, d: ec.get(k, d) if ec else d
        ratios = list(g("upsampling_ratios", [8, 6, 5, 4]))
        frame_rate, sr = float(g("frame_rate", 12.5)), int(g("sampling_rate", 24000))
        enc_rate = sr
        for r in ratios:
            enc_rate /= r
        if int(g("num_residual_layers", 1)) != 1 or bool(g("use_conv_shortcut", False)) or int(g("audio_channels", 1)) != 1 or not bool(g("use_causal_conv", True)):
            raise NotImplementedError("Qwen3-TTS tokenizer encoder: one residual layer per stage, true skip, mono, causal is what the engine builds")
        if int(g("num_key_value_heads", 8)) != int(g("num_attention_heads", 8)):
            raise NotImplementedError("Qwen3-TTS tokenizer encoder: kv_repeat must be 1 (the reference asserts the same, transformer.py:85)")
        return MimiConfig(dimension=int(g("hidden_size", 512)), nfilters=int(g("num_filters", 64)), ratios=ratios, ksize=int(g("kernel_size", 7)),
                          residual_ksize=int(g("residual_kernel_size", 3)), last_ksize=int(g("last_kernel_size", 3)), compress=int(g("compress", 2)),
                          num_heads=int(g("num_attention_heads", 8)), num_layers=int(g("num_hidden_layers", 8)), dim_feedforward=int(g("intermediate_size", 2048)),
                          context=int(g("sliding_window", 250)), max_period=float(int(g("rope_theta", 10000.0))), max_seq_len=int(g("max_position_embeddings", 8000)),
                          quantizer_nq=int(g("num_quantizers", 32)), quantizer_bins=int(g("codebook_size", 2048)), quantizer_dim=int(g("codebook_dim", 256)),
                          upsample_stride=int(enc_rate / frame_rate), sample_rate=sr, frame_rate=frame_rate, rope_interleaved=False, attn_window=0)

    @staticmethod
    def sanitize(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """PyTorch checkpoint -> decoder parameter names / layouts (speech_tokenizer.py:1380-1447): transposed-conv weights (in, out, K) ->
        (out, K, in), conv / 1x1 projection weights (out, in, K) -> (out, K, in) unless the shape heuristic says they already are, decoder
        codebooks materialised as ``embedding_sum / clip(cluster_usage, 1e-5)`` under ``...codebook.embed.weight``.  ``encoder.*`` is dropped."""
        out: Dict[str, torch.Tensor] = {}
        books: Dict[str, Dict[str, torch.Tensor]] = {}
        # encoder half (:1229-1381): SeanetEncoder layer N of the HF checkpoint -> module path; q / k / v -> one in_proj; codebooks regrouped below
        conv_map = {0: "encoder_model.encoder.init_conv1d", 3: "encoder_model.encoder.layers.0.downsample", 6: "encoder_model.encoder.layers.1.downsample",
                    9: "encoder_model.encoder.layers.2.downsample", 12: "encoder_model.encoder.layers.3.downsample", 14: "encoder_model.encoder.final_conv1d"}
        res_map, blk_map = {1: 0, 4: 1, 7: 2, 10: 3}, {1: 0, 3: 1}
        tr_map = {"self_attn.o_proj.weight": "self_attn.out_proj.weight", "mlp.fc1.weight": "gating.linear1.weight", "mlp.fc2.weight": "gating.linear2.weight",
                  "input_layernorm.weight": "norm1.weight", "input_layernorm.bias": "norm1.bias", "post_attention_layernorm.weight": "norm2.weight",
                  "post_attention_layernorm.bias": "norm2.bias", "self_attn_layer_scale.scale": "layer_scale_1.scale", "mlp_layer_scale.scale": "layer_scale_2.scale"}
        qkv: Dict[int, Dict[str, torch.Tensor]] = {}
        enc_books: Dict[str, Dict[str, torch.Tensor]] = {}
        for k, v in weights.items():
            if k.startswith("encoder."):
                if k.startswith("encoder.encoder.layers."):
                    parts = k.split(".")
                    n = int(parts[3])
                    if "block" in k:
                        if n not in res_map or int(parts[5]) not in blk_map:
                            continue
                        base, suffix = f"encoder_model.encoder.layers.{res_map[n]}.residuals.0.block.{blk_map[int(parts[5])]}", ".".join(parts[6:])
                    else:
                        if n not in conv_map:
                            continue
                        base, suffix = conv_map[n], ".".join(parts[4:])
                    out[f"{base}.conv.{suffix}"] = v.transpose(-1, -2).contiguous() if ("weight" in suffix and v.dim() == 3) else v
                elif k.startswith("encoder.encoder_transformer.layers."):
                    parts = k.split(".")
                    li, rest = int(parts[3]), ".".join(parts[4:])
                    hit = next((nm for nm in ("q", "k", "v") if f"self_attn.{nm}_proj.weight" in rest), None)
                    if hit:
                        qkv.setdefault(li, {})[hit] = v
                    else:
                        for src, dst in tr_map.items():
                            if src in rest:
                                out[f"encoder_model.encoder_transformer.transformer.layers.{li}.{dst}"] = v
                                break
                elif k.startswith("encoder.downsample."):
                    suffix = k.replace("encoder.downsample.", "")
                    out[f"encoder_model.downsample.conv.conv.{suffix}"] = v.transpose(-1, -2).contiguous() if ("weight" in suffix and v.dim() == 3) else v
                elif k.startswith("encoder.quantizer."):
                    rest = k.replace("encoder.quantizer.", "")
                    if ".codebook.cluster_usage" in rest or ".codebook.embed_sum" in rest:
                        enc_books.setdefault(rest.rsplit(".codebook.", 1)[0], {})["cluster_usage" if "cluster_usage" in rest else "embedding_sum"] = v
                    elif ".codebook.initialized" in rest:
                        pass
                    elif "input_proj.weight" in rest or "output_proj.weight" in rest:
                        half = "rvq_first" if "semantic_residual_vector_quantizer" in rest else "rvq_rest"
                        proj = "input_proj" if "input_proj" in rest else "output_proj"
                        out[f"encoder_model.quantizer.{half}.{proj}.weight"] = v.transpose(-1, -2).contiguous() if v.dim() == 3 else v
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
        for li, d in qkv.items():
            if "q" in d and "k" in d and "v" in d:
                out[f"encoder_model.encoder_transformer.transformer.layers.{li}.self_attn.in_proj.weight"] = torch.cat([d["q"], d["k"], d["v"]], dim=0)
        for base, d in enc_books.items():
            if "cluster_usage" in d and "embedding_sum" in d:
                m = re.search(r"layers\.(\d+)", base)
                half = "rvq_first" if "semantic_residual_vector_quantizer" in base else ("rvq_rest" if "acoustic_residual_vector_quantizer" in base else None)
                if m and half:
                    pfx = f"encoder_model.quantizer.{half}.vq.layers.{int(m.group(1))}.codebook"
                    out[f"{pfx}.embedding_sum"], out[f"{pfx}.cluster_usage"] = d["embedding_sum"], d["cluster_usage"]
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
        enc = {k[len("encoder_model."):]: v for k, v in w.items() if k.startswith("encoder_model.")}
        if enc and self.config.encoder_config is not None:   # the encoder half (in-context voice cloning): the Mimi engine under the tokenizer's configuration
            from ....codec.models.mimi.mimi import MimiEncoder

            try:
                self.encoder_model = MimiEncoder(enc, self.encoder_mimi_config(self.config.encoder_config), device=self.device, precision=self.precision)
            except KeyError as e:
                raise ValueError(f"Qwen3-TTS speech tokenizer checkpoint is missing encoder parameter {e}") from e
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
