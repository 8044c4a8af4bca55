"""Seeded synthetic Kokoro-82M checkpoints and inputs (no network => no real weights).

Shapes are exactly those implied by the reference ``ModelConfig``
(``tts/models/kokoro/kokoro.py:38-54``; constants pinned by
``tts/tests/test_models.py:143-173``) and parameter names are the reference's
post-``sanitize`` MLX names, so a real ``Kokoro-82M-bf16`` checkpoint and a
synthetic one are interchangeable everywhere (loader, oracle, benchmark).

Values are variance-preserving random draws rounded to bf16 (the checkpoint dtype
of config[1]) and returned as float32 tensors holding bf16-representable values.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

KOKORO_CONFIG = {
    "istftnet": {
        "upsample_kernel_sizes": [20, 12],
        "upsample_rates": [10, 6],
        "gen_istft_hop_size": 5,
        "gen_istft_n_fft": 20,
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "resblock_kernel_sizes": [3, 7, 11],
        "upsample_initial_channel": 512,
    },
    "dim_in": 64,
    "dropout": 0.2,
    "hidden_dim": 512,
    "max_conv_dim": 512,
    "max_dur": 50,
    "multispeaker": True,
    "n_layer": 3,
    "n_mels": 80,
    "n_token": 178,
    "style_dim": 128,
    "text_encoder_kernel_size": 5,
    "plbert": {
        "hidden_size": 768,
        "num_attention_heads": 12,
        "intermediate_size": 2048,
        "max_position_embeddings": 512,
        "num_hidden_layers": 12,
        "dropout": 0.1,
    },
    "vocab": {chr(0x100 + i): i for i in range(1, 178)},
    "sample_rate": 24000,
    "model_type": "kokoro",
}


def tiny_config() -> dict:
    """A structurally identical but much smaller model for fast CPU/GPU parity tests."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in KOKORO_CONFIG.items()}
    cfg["plbert"] = dict(hidden_size=64, num_attention_heads=4, intermediate_size=128,
                         max_position_embeddings=64, num_hidden_layers=2, dropout=0.1)
    return cfg


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _Gen:
    def __init__(self, seed: int, alpha_sep: str = "."):
        self.g = torch.Generator().manual_seed(seed)
        self.w: Dict[str, torch.Tensor] = {}
        self.alpha_sep = alpha_sep  # "." = Kokoro's list (alpha1.0); "_" = KittenTTS's attributes (alpha1_0)

    def normal(self, name, shape, std=1.0, mean=0.0):
        self.w[name] = _bf16(torch.randn(shape, generator=self.g) * std + mean)

    def uniform(self, name, shape, bound):
        self.w[name] = _bf16((torch.rand(shape, generator=self.g) * 2 - 1) * bound)

    def const(self, name, shape, value):
        self.w[name] = _bf16(torch.full(shape, float(value)))

    # ---- composite modules
    def linear(self, pre, out_f, in_f, bias=True, gain=1.0, bias_mean=0.0):
        self.normal(f"{pre}.weight", (out_f, in_f), gain / math.sqrt(in_f))
        if bias:
            self.normal(f"{pre}.bias", (out_f,), 0.02, bias_mean)

    def layernorm(self, pre, n):
        self.normal(f"{pre}.weight", (n,), 0.05, 1.0)
        self.normal(f"{pre}.bias", (n,), 0.05)

    def conv_weighted(self, pre, out_c, k, in_c, bias=True, gain=1.0, bias_n=None, bias_mean=0.0):
        """weight_v (out_c, k, in_c), weight_g = gain*||v|| (so the folded weight ~= gain*v)."""
        v = torch.randn((out_c, k, in_c), generator=self.g) / math.sqrt(k * in_c)
        v = _bf16(v)
        g = v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt() * gain
        self.w[f"{pre}.weight_v"] = v
        self.w[f"{pre}.weight_g"] = _bf16(g)
        if bias:
            self.normal(f"{pre}.bias", (bias_n or out_c,), 0.02, bias_mean)

    def lstm(self, pre, in_f, hid):
        b = 1.0 / math.sqrt(hid)
        for d in ("forward", "backward"):
            self.uniform(f"{pre}.Wx_{d}", (4 * hid, in_f), b)
            self.uniform(f"{pre}.Wh_{d}", (4 * hid, hid), b)
            self.uniform(f"{pre}.bias_ih_{d}", (4 * hid,), b)
            self.uniform(f"{pre}.bias_hh_{d}", (4 * hid,), b)

    def adain(self, pre, style, ch):
        self.linear(f"{pre}.fc", 2 * ch, style, gain=0.3)

    def adain_resblk1d(self, pre, din, dout, style, upsample=False):
        self.conv_weighted(f"{pre}.conv1", dout, 3, din)
        self.conv_weighted(f"{pre}.conv2", dout, 3, dout)
        self.adain(f"{pre}.norm1", style, din)
        self.adain(f"{pre}.norm2", style, dout)
        if din != dout:
            self.conv_weighted(f"{pre}.conv1x1", dout, 1, din, bias=False)
        if upsample:
            # depthwise ConvTranspose1d(k3, s2): weight_v (C, 3, 1); gain ~ sqrt(3/2) keeps variance
            self.conv_weighted(f"{pre}.pool", din, 3, 1, gain=1.2)

    def adain_resblock1(self, pre, ch, k, style):
        for i in range(3):
            self.conv_weighted(f"{pre}.convs1.{i}", ch, k, ch)
            self.conv_weighted(f"{pre}.convs2.{i}", ch, k, ch)
            self.adain(f"{pre}.adain1.{i}", style, ch)
            self.adain(f"{pre}.adain2.{i}", style, ch)
            self.normal(f"{pre}.alpha1{self.alpha_sep}{i}", (1, ch, 1), 0.15, 1.0)
            self.normal(f"{pre}.alpha2{self.alpha_sep}{i}", (1, ch, 1), 0.15, 1.0)


def make_kokoro_weights(config: dict = None, seed: int = 0, decoder_dims=(1024, 512, 64), alpha_sep: str = ".") -> Dict[str, torch.Tensor]:
    """``decoder_dims`` = (decoder block width, generator input width, asr_res width): Kokoro's constants (istftnet.py:948-975);
    KittenTTS reads them from its config (mlx_audio_amd/tts/models/kitten_tts/synthetic.py)."""
    cfg = config or KOKORO_CONFIG
    g = _Gen(seed, alpha_sep)
    cd, gd, ad = decoder_dims
    pb, hid, sty = cfg["plbert"], cfg["hidden_dim"], cfg["style_dim"]
    H, emb = pb["hidden_size"], pb.get("embedding_size", 128)
    # ---- PL-BERT (CustomAlbert)
    g.normal("bert.embeddings.word_embeddings.weight", (cfg["n_token"], emb), 0.5)
    g.normal("bert.embeddings.position_embeddings.weight", (pb["max_position_embeddings"], emb), 0.5)
    g.normal("bert.embeddings.token_type_embeddings.weight", (2, emb), 0.5)
    g.layernorm("bert.embeddings.LayerNorm", emb)
    g.linear("bert.encoder.embedding_hidden_mapping_in", H, emb)
    lay = "bert.encoder.albert_layer_groups.0.albert_layers.0"
    for n in ("query", "key", "value", "dense"):
        g.linear(f"{lay}.attention.{n}", H, H)
    g.layernorm(f"{lay}.attention.LayerNorm", H)
    g.layernorm(f"{lay}.full_layer_layer_norm", H)
    g.linear(f"{lay}.ffn", pb["intermediate_size"], H)
    g.linear(f"{lay}.ffn_output", H, pb["intermediate_size"])
    g.linear("bert.pooler", H, H)
    g.linear("bert_encoder", hid, H)
    # ---- prosody predictor
    for i in range(cfg["n_layer"]):
        g.lstm(f"predictor.text_encoder.lstms.{2 * i}", hid + sty, hid // 2)
        g.linear(f"predictor.text_encoder.lstms.{2 * i + 1}.fc", 2 * hid, sty, gain=0.3)
    g.lstm("predictor.lstm", hid + sty, hid // 2)
    # bias -2.6: sigmoid ~0.07 x 50 bins ~ 3.5 frames / phoneme
    g.linear("predictor.duration_proj.linear_layer", cfg["max_dur"], hid, gain=8.0, bias_mean=-2.6)
    g.lstm("predictor.shared", hid + sty, hid // 2)
    for br in ("F0", "N"):
        g.adain_resblk1d(f"predictor.{br}.0", hid, hid, sty)
        g.adain_resblk1d(f"predictor.{br}.1", hid, hid // 2, sty, upsample=True)
        g.adain_resblk1d(f"predictor.{br}.2", hid // 2, hid // 2, sty)
    # F0 in Hz: mean 120, spread ~80 => voiced/unvoiced mix around the 10 Hz threshold
    g.normal("predictor.F0_proj.weight", (1, 1, hid // 2), 240.0 / math.sqrt(hid // 2))
    g.const("predictor.F0_proj.bias", (1,), 120.0)
    g.normal("predictor.N_proj.weight", (1, 1, hid // 2), 1.0 / math.sqrt(hid // 2))
    g.const("predictor.N_proj.bias", (1,), 0.0)
    # ---- text encoder
    g.normal("text_encoder.embedding.weight", (cfg["n_token"], hid), 1.0)
    k = cfg["text_encoder_kernel_size"]
    for i in range(cfg["n_layer"]):
        g.conv_weighted(f"text_encoder.cnn.{i}.0", hid, k, hid)
        g.layernorm(f"text_encoder.cnn.{i}.1", hid)
    g.lstm("text_encoder.lstm", hid, hid // 2)
    # ---- decoder
    ist = cfg["istftnet"]
    g.adain_resblk1d("decoder.encode", hid + 2, cd, sty)
    for i in range(3):
        g.adain_resblk1d(f"decoder.decode.{i}", cd + 2 + ad, cd, sty)
    g.adain_resblk1d("decoder.decode.3", cd + 2 + ad, gd, sty, upsample=True)
    g.conv_weighted("decoder.F0_conv", 1, 3, 1, gain=0.02)   # keeps the F0 channel O(1)
    g.conv_weighted("decoder.N_conv", 1, 3, 1)
    g.conv_weighted("decoder.asr_res.0", ad, 1, hid)
    gen = "decoder.generator"
    g.normal(f"{gen}.m_source.l_linear.weight", (1, 9), 1.0)
    g.normal(f"{gen}.m_source.l_linear.bias", (1,), 0.02)
    c0 = ist["upsample_initial_channel"]
    nfft = ist["gen_istft_n_fft"]
    rates, kers = ist["upsample_rates"], ist["upsample_kernel_sizes"]
    nk = len(ist["resblock_kernel_sizes"])
    ch = c0
    for i, (u, kk) in enumerate(zip(rates, kers)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        # ConvWeighted(cout, cin, k, encode=True): weight_v (cin, k, cout), bias (cout); gain sqrt(u)
        g.conv_weighted(f"{gen}.ups.{i}", cin, kk, cout, bias_n=cout, gain=math.sqrt(u * cin / cout))
        ch = cout
        for j, rk in enumerate(ist["resblock_kernel_sizes"]):
            g.adain_resblock1(f"{gen}.resblocks.{i * nk + j}", ch, rk, sty)
        if i + 1 < len(rates):
            sf = int(math.prod(rates[i + 1:]))
            g.normal(f"{gen}.noise_convs.{i}.weight", (ch, sf * 2, nfft + 2), 1.0 / math.sqrt(sf * 2 * (nfft + 2)))
            g.normal(f"{gen}.noise_convs.{i}.bias", (ch,), 0.02)
            g.adain_resblock1(f"{gen}.noise_res.{i}", ch, 7, sty)
        else:
            g.normal(f"{gen}.noise_convs.{i}.weight", (ch, 1, nfft + 2), 1.0 / math.sqrt(nfft + 2))
            g.normal(f"{gen}.noise_convs.{i}.bias", (ch,), 0.02)
            g.adain_resblock1(f"{gen}.noise_res.{i}", ch, 11, sty)
    # conv_post: log-magnitudes around -2 (|spec| ~ 0.1), phases O(1)
    g.conv_weighted(f"{gen}.conv_post", nfft + 2, 7, ch, gain=0.5)
    b = g.w[f"{gen}.conv_post.bias"]
    b[: nfft // 2 + 1] -= 2.0
    g.w[f"{gen}.conv_post.bias"] = _bf16(b)
    return g.w


def as_float32_checkpoint(weights: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """The same checkpoint with every value moved off the bf16 grid (x (1 + u 2^-9), u uniform in [-1, 1)): what a genuinely float32 checkpoint looks
    like to the engines' 16-bit weight images (tests of precision 4)."""
    g = torch.Generator().manual_seed(7000 + seed)
    return {k: (v * (1.0 + (torch.rand(v.shape, generator=g) * 2 - 1) * 2.0 ** -9)).to(torch.float32) if v.is_floating_point() else v
            for k, v in sorted(weights.items())}


def make_voice_pack(seed: int = 1, rows: int = 510) -> torch.Tensor:
    """Voice pack [rows, 1, 256] ~ N(0, 0.1) (fp32), indexed by ``len(phonemes) - 1``."""
    gen = torch.Generator().manual_seed(seed)
    return torch.randn((rows, 1, 256), generator=gen) * 0.1


def make_phoneme_ids(n: int, seed: int = 2, n_token: int = 178) -> torch.Tensor:
    """n phoneme ids uniform in 1..n_token-1, wrapped with the 0 BOS/EOS tokens -> LongTensor [n+2]."""
    gen = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, n_token, (n,), generator=gen)
    return torch.cat([torch.zeros(1, dtype=torch.long), ids, torch.zeros(1, dtype=torch.long)])


def forced_durations(T: int, total_frames: int, seed: int = 3) -> torch.Tensor:
    """Deterministic int32 durations (each >= 1) summing to ``total_frames`` (T=80, F=264)."""
    gen = torch.Generator().manual_seed(seed)
    base = torch.ones(T, dtype=torch.int64)
    extra = total_frames - T
    assert extra >= 0
    if extra:
        picks = torch.randint(0, T, (extra,), generator=gen)
        base += torch.bincount(picks, minlength=T)
    return base.to(torch.int32)
