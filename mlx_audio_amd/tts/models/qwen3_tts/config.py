"""Configuration records of Qwen3-TTS (``mlx_audio/tts/models/qwen3_tts/config.py:36-136``): same field names and defaults."""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Any, Dict, List, Optional


def filter_dict_for_dataclass(cls, data: Dict[str, Any]) -> Dict[str, Any]:
    valid = {f.name for f in fields(cls)}
    return {k: v for k, v in data.items() if k in valid}


@dataclass
class Qwen3TTSTalkerCodePredictorConfig:
    vocab_size: int = 2048
    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 5
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    hidden_act: str = "silu"
    max_position_embeddings: int = 65536
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    attention_bias: bool = False
    num_code_groups: int = 16


@dataclass
class Qwen3TTSTalkerConfig:
    code_predictor_config: Optional[Qwen3TTSTalkerCodePredictorConfig] = None
    vocab_size: int = 3072
    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    hidden_act: str = "silu"
    max_position_embeddings: int = 32768
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    rope_scaling: Optional[Dict] = field(default_factory=lambda: {"interleaved": True, "mrope_section": [24, 20, 20], "rope_type": "default"})
    attention_bias: bool = False
    num_code_groups: int = 16
    text_hidden_size: int = 2048
    text_vocab_size: int = 151936
    codec_eos_token_id: int = 2150
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149

    def __post_init__(self):
        if self.code_predictor_config is None:
            self.code_predictor_config = Qwen3TTSTalkerCodePredictorConfig()
        elif isinstance(self.code_predictor_config, dict):
            self.code_predictor_config = Qwen3TTSTalkerCodePredictorConfig(
                **filter_dict_for_dataclass(Qwen3TTSTalkerCodePredictorConfig, self.code_predictor_config))


def talker_1p7b() -> Qwen3TTSTalkerConfig:
    """BASELINE config[3] (Qwen3-TTS-1.7B): hidden 2048 / intermediate 6144 / 28 layers / 16-8 heads (the reference
    defaults above are the 0.6B sizes)."""
    return Qwen3TTSTalkerConfig(hidden_size=2048, intermediate_size=6144)


@dataclass
class Qwen3TTSTokenizerDecoderConfig:
    attention_bias: bool = False
    latent_dim: int = 1024
    codebook_dim: int = 512
    codebook_size: int = 2048
    decoder_dim: int = 1536
    hidden_act: str = "silu"
    hidden_size: int = 512
    intermediate_size: int = 1024
    layer_scale_initial_scale: float = 0.01
    max_position_embeddings: int = 8000
    head_dim: int = 64
    num_attention_heads: int = 16
    num_hidden_layers: int = 8
    num_key_value_heads: int = 16
    num_quantizers: int = 16
    num_semantic_quantizers: int = 1
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    semantic_codebook_size: int = 4096
    sliding_window: int = 72  # stored but not applied by the reference (speech_tokenizer.py:242, 400-404): full causal mask
    upsample_rates: List[int] = field(default_factory=lambda: [8, 5, 4, 3])
    upsampling_ratios: List[int] = field(default_factory=lambda: [2, 2])
    vector_quantization_hidden_dimension: int = 512
