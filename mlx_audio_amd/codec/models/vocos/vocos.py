"""Vocos (mel / EnCodec-feature vocoder) on MI355X: host schedule over the HIP kernels.

Mirrors ``mlx_audio/codec/models/vocos/vocos.py`` (class and method names, config layout, load-time transposes):
  * ``MelSpectrogramFeatures`` (:25-51)   -> ``mel.log_mel_spectrogram``: one fused STFT / |X| / mel / log launch;
  * ``VocosBackbone`` (:214-273)          -> embed conv k7 as implicit GEMM, LayerNorm kernel (AdaLayerNorm: affine-free statistics with the
                                             per-item scale / shift vectors as weight / bias), ``ConvNeXtBlock`` (:137-190) = depthwise k7
                                             kernel + LayerNorm + GEMM with exact-GELU epilogue + GEMM with gamma and residual in the epilogue;
  * ``ISTFTHead`` (:116-134)              -> GEMM, ``polar_spec`` (exp / clip / cos / sin -> complex), the generic iSTFT kernels with the
                                             reference's plain-window overlap-add normalisation (``dsp.istft``, ``normalized=False``).
``EncodecFeatures`` needs the EnCodec encoder, which is outside this build: constructing it raises; ``decode`` with
externally computed features and ``bandwidth_id`` works.

Weights: the published checkpoints are float32.  The MFMA path holds them as fp16 images (11-bit mantissa) and splits the fp32 activations
into fp16 hi + lo (``precision = 4``); the measured deviation from the float32 oracle is stated in DESIGN.md section 4 and asserted in
``tests/test_vocos_gpu.py``.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional

import torch

from .... import dsp, ops
from ....ops import ACT_GELU
from .mel import log_mel_spectrogram


def make_vocos_weights(config: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random float32 parameters of the exact shapes ``Vocos.from_hparams(config)`` allocates (names after the load-time transposes)."""
    g = torch.Generator().manual_seed(seed)
    b = config["backbone"]["init_args"]
    h = config["head"]["init_args"]
    cin, dim, inter, nl = b["input_channels"], b["dim"], b["intermediate_dim"], b["num_layers"]
    k_in, k_dw = b.get("input_kernel_size", 7), b.get("dw_kernel_size", 7)
    ada = b.get("adanorm_num_embeddings")
    ls = b.get("layer_scale_init_value") or 1 / nl

    def rn(*shape, std):
        return torch.randn(*shape, generator=g) * std

    w: Dict[str, torch.Tensor] = {"backbone.embed.weight": rn(dim, k_in, cin, std=(k_in * cin) ** -0.5), "backbone.embed.bias": rn(dim, std=0.02)}

    def norm(name):
        if ada:
            w[name + ".scale.weight"] = 1.0 + rn(dim, ada, std=0.05)
            w[name + ".scale.bias"] = rn(dim, std=0.02)
            w[name + ".shift.weight"] = rn(dim, ada, std=0.05)
            w[name + ".shift.bias"] = rn(dim, std=0.02)
        else:
            w[name + ".weight"] = 1.0 + rn(dim, std=0.1)
            w[name + ".bias"] = rn(dim, std=0.05)

    norm("backbone.norm")
    for i in range(nl):
        p = f"backbone.convnext.{i}."
        w[p + "dwconv.weight"] = rn(dim, k_dw, 1, std=k_dw ** -0.5)
        w[p + "dwconv.bias"] = rn(dim, std=0.02)
        norm(p + "norm")
        w[p + "pwconv1.weight"] = rn(inter, dim, std=dim ** -0.5)
        w[p + "pwconv1.bias"] = rn(inter, std=0.02)
        w[p + "pwconv2.weight"] = rn(dim, inter, std=inter ** -0.5)
        w[p + "pwconv2.bias"] = rn(dim, std=0.02)
        w[p + "gamma"] = ls * (1.0 + rn(dim, std=0.1))
    w["backbone.final_layer_norm.weight"] = 1.0 + rn(dim, std=0.1)
    if b.get("bias", True):
        w["backbone.final_layer_norm.bias"] = rn(dim, std=0.05)
    w["head.out.weight"] = rn(h["n_fft"] + 2, dim, std=0.5 * dim ** -0.5)
    w["head.out.bias"] = rn(h["n_fft"] + 2, std=0.1)
    return w


class FeatureExtractor:
    """Base class for feature extractors (vocos.py:18-22)."""

    def __call__(self, audio, **kwargs):
        raise NotImplementedError("Subclasses must implement the forward method.")


class MelSpectrogramFeatures(FeatureExtractor):
    def __init__(self, sample_rate=24_000, n_fft=1024, hop_length=256, n_mels=100, padding="center"):
        if padding not in ["center", "same"]:
            raise ValueError("Padding must be 'center' or 'same'.")
        self.padding = padding
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels

    def __call__(self, audio, **kwargs):
        return log_mel_spectrogram(audio, sample_rate=self.sample_rate, n_mels=self.n_mels, n_fft=self.n_fft, hop_length=self.hop_length, padding=0)


class EncodecFeatures(FeatureExtractor):
    """vocos.py:54-116: EnCodec codes as features -- ``get_encodec_codes`` (preprocess -> ``Encodec.encode`` at ``bandwidths[bandwidth_id]``) and
    ``get_features_from_codes`` (sum over the quantizers of their codebook rows: one ``embed_sum`` launch over the stacked codebooks).

    The reference downloads the EnCodec checkpoint from the hub by name; there is no hub here, so the model comes in as ``encodec=`` (an
    ``mlx_audio_amd...encodec.Encodec``; ``preprocessor`` defaults to its ``preprocess_audio`` partial) or ``encodec_model`` names a LOCAL directory
    (``config.json`` + ``model.safetensors``, ``Encodec.from_pretrained``)."""

    HUB = {"encodec_24khz": "mlx-community/encodec-24khz-float32", "encodec_48khz": "mlx-community/encodec-48khz-float32"}

    def __init__(self, encodec_model: str = "encodec_24khz", bandwidths: List[float] = [1.5, 3.0, 6.0, 12.0], train_codebooks: bool = False,
                 encodec=None, preprocessor=None, device="cuda:0"):
        import functools
        import os

        from ..encodec.encodec import Encodec, preprocess_audio

        if encodec is None:
            if os.path.isdir(str(encodec_model)):
                encodec, preprocessor = Encodec.from_pretrained(encodec_model, device=device)
            elif encodec_model in self.HUB:
                raise FileNotFoundError(f"EncodecFeatures: '{encodec_model}' is {self.HUB[encodec_model]} on the hub, which is not reachable here; pass "
                                        "encodec=<Encodec> or a local checkpoint directory as encodec_model")
            else:
                raise ValueError(f"Unsupported encodec_model: {encodec_model}. Supported options are 'encodec_24khz' and 'encodec_48khz'.")
        self.encodec = encodec
        self.preprocessor = preprocessor or functools.partial(preprocess_audio, sampling_rate=encodec.sampling_rate, chunk_length=encodec.chunk_length,
                                                              chunk_stride=encodec.chunk_stride)
        self.num_q = self.encodec.quantizer.get_num_quantizers_for_bandwidth(bandwidth=max(bandwidths))
        self.bandwidths = list(bandwidths)

    @property
    def codebook_weights(self) -> torch.Tensor:
        """The first ``num_q`` codebooks stacked on axis 0 (vocos.py:81-83)."""
        return self.encodec.quantizer.table[: self.num_q * self.encodec.quantizer.codebook_size]

    def get_encodec_codes(self, audio, bandwidth_id) -> torch.Tensor:
        """audio [L] / [L, C] -> codes [nq, 1, T] (vocos.py:86-97)."""
        features, mask = self.preprocessor(audio)
        if isinstance(bandwidth_id, torch.Tensor):
            bandwidth_id = int(bandwidth_id.flatten().tolist()[0])
        elif isinstance(bandwidth_id, list):
            bandwidth_id = bandwidth_id[0]
        codes, _ = self.encodec.encode(features, mask, bandwidth=self.bandwidths[bandwidth_id])
        return codes.reshape(codes.shape[-2], 1, codes.shape[-1])

    def get_features_from_codes(self, codes) -> torch.Tensor:
        """codes [nq, B, T] -> features [B, T, codebook_dim] (vocos.py:99-108)."""
        codes = torch.as_tensor(codes)
        if codes.shape[0] > self.num_q:
            raise IndexError(f"get_features_from_codes: {codes.shape[0]} quantizers given, the extractor holds {self.num_q}")
        return self.encodec.quantizer.decode(codes.permute(1, 0, 2))

    def __call__(self, audio, **kwargs):
        bandwidth_id = kwargs.get("bandwidth_id")
        if bandwidth_id is None:
            raise ValueError("The 'bandwidth_id' argument is required")
        return self.get_features_from_codes(self.get_encodec_codes(audio, bandwidth_id=bandwidth_id))


class _Norm:
    def __init__(self, w: Dict[str, torch.Tensor], name: str, dev):
        self.ada = (name + ".scale.weight") in w
        self.dev = dev
        if self.ada:  # Linear(num_embeddings -> dim) on the conditioning vector: a few hundred MACs, evaluated on the host in float32
            self.sw, self.sb = w[name + ".scale.weight"].float().cpu(), w[name + ".scale.bias"].float().cpu()
            self.hw, self.hb = w[name + ".shift.weight"].float().cpu(), w[name + ".shift.bias"].float().cpu()
        else:
            self.weight = w[name + ".weight"].to(dev)
            self.bias = w[name + ".bias"].to(dev) if (name + ".bias") in w else None

    def __call__(self, x: torch.Tensor, y: torch.Tensor, cond: Optional[torch.Tensor]):
        if not self.ada:
            return ops.layernorm(x, y, weight=self.weight, bias=self.bias, eps=1e-6)
        assert cond is not None, "AdaLayerNorm needs bandwidth_id"
        scale = (cond @ self.sw.t() + self.sb).to(self.dev)   # [Bc, dim]; Bc == 1 broadcasts over the batch like the reference
        shift = (cond @ self.hw.t() + self.hb).to(self.dev)
        for b in range(x.shape[0]):   # the affine-free statistics + per-item affine pair: one LayerNorm launch per item
            i = b if scale.shape[0] > 1 else 0
            ops.layernorm(x[b:b + 1], y[b:b + 1], weight=scale[i].contiguous(), bias=shift[i].contiguous(), eps=1e-6)
        return y


class VocosBackbone:
    def __init__(self, input_channels: int, dim: int, intermediate_dim: int, num_layers: int, layer_scale_init_value: Optional[float] = None,
                 adanorm_num_embeddings: Optional[int] = None, bias: bool = True, input_kernel_size: int = 7, dw_kernel_size: int = 7):
        self.input_channels, self.dim, self.intermediate_dim, self.num_layers = input_channels, dim, intermediate_dim, num_layers
        self.layer_scale_init_value = layer_scale_init_value or 1 / num_layers
        self.adanorm = adanorm_num_embeddings is not None
        self.adanorm_num_embeddings = adanorm_num_embeddings
        self.bias, self.input_kernel_size, self.dw_kernel_size = bias, input_kernel_size, dw_kernel_size
        self._loaded = False

    def load(self, w: Dict[str, torch.Tensor], dev):
        self.dev = dev
        self.embed = ops.pack_conv(w["backbone.embed.weight"], w["backbone.embed.bias"], dev, f16=True)
        self.norm = _Norm(w, "backbone.norm", dev)
        self.blocks = []
        for i in range(self.num_layers):
            p = f"backbone.convnext.{i}."
            self.blocks.append(dict(dw_w=w[p + "dwconv.weight"][:, :, 0].to(torch.float32).contiguous().to(dev), dw_b=w[p + "dwconv.bias"].to(dev),
                                    norm=_Norm(w, p + "norm", dev),
                                    pw1=ops.pack_conv(w[p + "pwconv1.weight"], w[p + "pwconv1.bias"], dev, f16=True),
                                    pw2=ops.pack_conv(w[p + "pwconv2.weight"], w[p + "pwconv2.bias"], dev, f16=True),
                                    gamma=w[p + "gamma"].to(dev) if (p + "gamma") in w else None))
        self.final_w = w["backbone.final_layer_norm.weight"].to(dev)
        self.final_b = w["backbone.final_layer_norm.bias"].to(dev) if "backbone.final_layer_norm.bias" in w else None
        self._loaded = True

    def __call__(self, x: torch.Tensor, return_layers: bool = False, **kwargs):
        assert self._loaded, "VocosBackbone has no weights"
        bandwidth_id = kwargs.get("bandwidth_id", None)
        cond = None if bandwidth_id is None else torch.as_tensor(bandwidth_id).to(device="cpu", dtype=torch.float32).reshape(-1, self.adanorm_num_embeddings or 1)
        x = torch.as_tensor(x, dtype=torch.float32).to(self.dev)
        if x.shape[-1] != self.input_channels:  # vocos.py:253-255
            x = x.transpose(1, 2)
        x = x.contiguous()
        B, T, _ = x.shape

        def f(c):
            return torch.empty((B, T, c), dtype=torch.float32, device=self.dev)

        h = f(self.dim)
        ops.conv_gemm(x, self.embed, h, pad=self.embed.k // 2, precision=4)
        if self.adanorm:
            assert cond is not None  # vocos.py:259-261
        self.norm(h, h, cond)
        layers = [h.clone()] if return_layers else None
        d, m = f(self.dim), f(self.intermediate_dim)
        for blk in self.blocks:
            ops.dwconv(h, blk["dw_w"], blk["dw_b"], d, pad=blk["dw_w"].shape[1] // 2)
            blk["norm"](d, d, cond)
            ops.conv_gemm(d, blk["pw1"], m, post_act=ACT_GELU, precision=4)
            ops.conv_gemm(m, blk["pw2"], h, colscale=blk["gamma"], res=h, precision=4)
            if return_layers:
                layers.append(h.clone())
        out = f(self.dim)
        ops.layernorm(h, out, weight=self.final_w, bias=self.final_b, eps=1e-6)
        return (out, layers) if return_layers else out


class ISTFTHead:
    def __init__(self, dim: int, n_fft: int, hop_length: int, padding: str = "center"):
        self.dim, self.n_fft, self.hop_length = dim, n_fft, hop_length  # `padding` is accepted and unused, like the reference (vocos.py:117-121)
        self._loaded = False

    def load(self, w: Dict[str, torch.Tensor], dev):
        self.dev = dev
        self.out = ops.pack_conv(w["head.out.weight"], w["head.out.bias"], dev, f16=True)
        self.window = dsp.hanning(self.n_fft).to(torch.float32)
        self._loaded = True

    def __call__(self, x: torch.Tensor, return_spec: bool = False):
        """x [B, T, dim] -> audio [(T - 1) * hop] for B == 1 (the reference squeezes the batch axis), [B, (T - 1) * hop] otherwise."""
        assert self._loaded, "ISTFTHead has no weights"
        B, T, _ = x.shape
        nb = self.n_fft // 2 + 1
        y = torch.empty((B, T, 2 * nb), dtype=torch.float32, device=self.dev)
        ops.conv_gemm(x, self.out, y, precision=4)
        spec = ops.polar_spec(y, nb, 1e2)
        w = self.window.cpu().contiguous()
        env = dsp._ola_envelope(w.numpy().tobytes(), self.n_fft, T, self.hop_length, False)
        trim = self.n_fft // 2
        audio = ops.istft_frames(spec, self.n_fft, self.hop_length, w.to(self.dev), torch.from_numpy(env).to(self.dev), 1, False, trim,
                                 env.shape[0] - 2 * trim)
        audio = audio[0] if B == 1 else audio
        return (audio, spec) if return_spec else audio


class Vocos:
    def __init__(self, feature_extractor: FeatureExtractor, backbone: VocosBackbone, head: ISTFTHead):
        self.feature_extractor, self.backbone, self.head = feature_extractor, backbone, head

    @classmethod
    def from_hparams(cls, config: dict, weights: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", seed: int = 0, encodec=None) -> "Vocos":
        """Model from the hyper-parameters of a Vocos ``config.yaml`` (vocos.py:288-303).  ``weights`` (extra): a parameter dict in the
        reference's names; omitted, the parameters are random like a freshly constructed reference model.  ``encodec`` (extra): the EnCodec model of
        an ``EncodecFeatures`` extractor (the reference fetches it from the hub by name)."""
        ops.require_gpu()
        fe = config["feature_extractor"]
        if "MelSpectrogramFeatures" in fe["class_path"]:
            feature_extractor = MelSpectrogramFeatures(**fe["init_args"])
        elif "EncodecFeatures" in fe["class_path"]:
            try:
                feature_extractor = EncodecFeatures(**fe["init_args"], encodec=encodec, device=device)
            except FileNotFoundError:   # hub name without a local model: decode() with external features still works; __call__ raises
                feature_extractor = None
        else:
            raise ValueError(f"unknown feature extractor {fe['class_path']}")
        backbone = VocosBackbone(**config["backbone"]["init_args"])
        head = ISTFTHead(**config["head"]["init_args"])
        model = cls(feature_extractor, backbone, head)
        model.config = config
        model.load_weights(weights if weights is not None else make_vocos_weights(config, seed), device)
        return model

    def load_weights(self, weights: Dict[str, torch.Tensor], device="cuda:0"):
        dev = torch.device(device)
        w = {k: torch.as_tensor(v).detach().to(torch.float32).cpu() for k, v in weights.items()
             if k.startswith(("backbone.", "head.")) and "window" not in k}
        self.backbone.load(w, dev)
        self.head.load(w, dev)
        return self

    @classmethod
    def from_pretrained(cls, path_or_repo: str, device="cuda:0") -> "Vocos":
        """Local directory with ``config.yaml`` + ``model.safetensors`` (vocos.py:305-353; no hub download here: there is no network)."""
        import yaml
        from safetensors.torch import load_file

        path = Path(path_or_repo)
        if not path.exists():
            raise FileNotFoundError(f"{path_or_repo}: Vocos.from_pretrained needs a local directory (no hub access in this build)")
        weights = load_file(str(path / "model.safetensors"))
        with open(path / "config.yaml", "r") as f:
            config = yaml.safe_load(f)
        new = {}
        for k, v in weights.items():  # torch checkpoint layout -> the reference's (vocos.py:337-347)
            basename, pname = k.rsplit(".", 1)
            if ("backbone.embed" in basename or "dwconv" in basename) and pname == "weight":
                new[k] = v.movedim(1, 2)
            else:
                new[k] = v
        return cls.from_hparams(config, weights=new, device=device)

    def __call__(self, audio_input, **kwargs: Any):
        if self.feature_extractor is None:
            raise FileNotFoundError("this Vocos was configured with EncodecFeatures by hub name and no EnCodec model was supplied "
                                    "(from_hparams(..., encodec=<Encodec>)); decode(features, bandwidth_id=...) works without one")
        features = self.feature_extractor(audio_input, **kwargs)
        return self.decode(features, **kwargs)

    def get_encodec_codes(self, audio_input, bandwidth_id: int):
        if not isinstance(self.feature_extractor, EncodecFeatures):
            raise ValueError("This model does not support getting encodec codes.")
        return self.feature_extractor.get_encodec_codes(audio_input, bandwidth_id)

    def decode(self, features_input, **kwargs: Any):
        x = self.backbone(features_input, **kwargs)
        return self.head(x)

    def decode_from_codes(self, codes, **kwargs: Any):
        """vocos.py:372-375: codes [nq, B, T] -> audio."""
        if not isinstance(self.feature_extractor, EncodecFeatures):
            raise ValueError("decode_from_codes needs an EncodecFeatures extractor (its codebooks)")
        return self.decode(self.feature_extractor.get_features_from_codes(codes), **kwargs)
