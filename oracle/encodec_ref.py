"""PyTorch-CPU restatement of the EnCodec decode and encode paths (TEST ORACLE, not product).

Follows /root/reference/mlx_audio/codec/models/encodec/encodec.py statement by statement:
  * ``:89-167``   LSTM: x @ Wx^T + bias for every step, then per step hidden @ Wh^T (zeros at step 0) + the Metal ``lstm`` kernel: gate chunks
                  i | f | g | o of the 4H pre-activations, sigmoid(x) = 1 / (1 + exp(-|x|)) mirrored for x < 0, precise tanh,
                  c = f c + i g, h = o tanh(c).  (The kernel's index arithmetic is only consistent for batch 1 -- ``elem = b * 4H + y`` with y running
                  to B * H; restated as the per-sequence LSTM it implements at B = 1, which is also what it means.)
  * ``:170-254``  EncodecConv1d: causal left padding ``(k - 1) * dilation + 1 - stride`` ... precisely ``padding_total = kernel_size - stride`` with the
                  UN-dilated kernel size (the reference's own arithmetic: the effective kernel only enters the extra right padding), reflect or
                  zero padding, then nn.Conv1d
  * ``:257-293``  EncodecConvTranspose1d: full transposed conv, then trim ``padding_total = kernel_size - stride`` samples (causal: all on the right
                  for trim_right_ratio = 1)
  * ``:296-306``  EncodecLSTM: stacked LSTMs + skip connection
  * ``:309-344``  EncodecResnetBlock: ELU, conv k3 (dim -> dim / compress), ELU, conv k1, + shortcut conv k1
  * ``:391-444``  EncodecDecoder: conv k7 -> LSTM -> per ratio (ELU, convT K = 2 r, resnet blocks) -> ELU -> conv k7
  * ``:447-547``  Euclidean codebooks / RVQ decode: sum of the codebook rows
  * ``:679-777``  Encodec._decode_frame / decode / _linear_overlap_add
  * ``:340-389``  EncodecEncoder (round 5): conv k7 -> per ratio (reversed) resnet blocks, ELU, conv K = 2 r stride r -> LSTM -> ELU -> conv k7
  * ``:452-469, 516-533``  Euclidean codebook search ``argmax -(|x|^2 - 2 x e^T + |e|^2)`` and the residual loop of ``quantizer.encode``
  * ``:556-650``  Encodec._encode_frame (optional loudness normalisation) / encode (chunk loop)
    -- pinned by ``ref_encodec_encode.npz`` (the reference's own ``Encodec.encode`` run; every code equal)
Parameter names are the reference's module paths (``decoder.layers.N...``), layouts MLX's (conv ``[out, K, in]``).  ``norm_type = "weight_norm"``
checkpoints carry folded weights (the reference's modules hold plain ``nn.Conv1d``); ``time_group_norm`` (the 48 kHz model) is not restated.
Parity status: **pinned to the reference's own modules**: tests/golden/make_reference_fixtures.py runs the reference's ``Encodec.decode`` (over the
numpy stand-in for MLX, with the Metal kernel's source restated as a Python callable in tests/golden/mlx_shim.py) on a seeded checkpoint;
tests/test_reference_fixtures_cpu.py holds this oracle to the result.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def lstm_sigmoid(x: Tensor) -> Tensor:
    y = 1.0 / (1.0 + torch.exp(-x.abs()))
    return torch.where(x < 0, 1.0 - y, y)


class EncodecDecoderRef:
    def __init__(self, weights: Dict[str, Tensor], config: dict, dtype=torch.float32):
        self.w = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in weights.items()}
        self.c = dict(audio_channels=1, num_filters=32, kernel_size=7, num_residual_layers=1, dilation_growth_rate=2, codebook_size=1024, codebook_dim=128,
                      hidden_size=128, num_lstm_layers=2, residual_kernel_size=3, use_causal_conv=True, normalize=False, pad_mode="reflect",
                      norm_type="weight_norm", last_kernel_size=7, trim_right_ratio=1.0, compress=2, upsampling_ratios=[8, 5, 4, 2], sampling_rate=24000,
                      chunk_length_s=None, overlap=None, use_conv_shortcut=True)
        self.c.update(config)
        assert self.c["norm_type"] == "weight_norm"
        self.dtype = dtype

    # ------------------------------------------------------------------ modules
    def conv(self, x: Tensor, name: str, kernel_size: int, dilation: int = 1, stride: int = 1) -> Tensor:
        """x [B, L, C] (encodec.py:213-254)."""
        c = self.c
        k_eff = (kernel_size - 1) * dilation + 1
        padding_total = kernel_size - stride
        length = x.shape[1]
        n_frames = int(math.ceil((length - k_eff + padding_total) / stride + 1)) - 1
        extra = n_frames * stride + k_eff - padding_total - length
        if c["use_causal_conv"]:
            pl, pr = padding_total, extra
        else:
            pr = padding_total // 2
            pl = padding_total - pr
            pr += extra
        if c["pad_mode"] == "reflect":
            prefix = x[:, 1:pl + 1].flip(1)
            suffix = x[:, max(length - (pr + 1), 0):-1].flip(1)
            x = torch.cat([prefix, x, suffix], dim=1)
        else:
            x = F.pad(x, (0, 0, pl, pr))
        w = self.w[name + ".conv.weight"]  # [out, K, in]
        return F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), self.w.get(name + ".conv.bias"), stride=stride, dilation=dilation).transpose(1, 2)

    def convT(self, x: Tensor, name: str, kernel_size: int, stride: int) -> Tensor:
        c = self.c
        w = self.w[name + ".conv.weight"]  # [out, K, in]
        y = F.conv_transpose1d(x.transpose(1, 2), w.permute(2, 0, 1), self.w.get(name + ".conv.bias"), stride=stride).transpose(1, 2)
        padding_total = kernel_size - stride
        pr = math.ceil(padding_total * c["trim_right_ratio"]) if c["use_causal_conv"] else padding_total // 2
        pl = padding_total - pr
        return y[:, pl:y.shape[1] - pr]

    def lstm(self, x: Tensor, name: str) -> Tensor:
        wx, wh, b = self.w[name + ".Wx"], self.w[name + ".Wh"], self.w.get(name + ".bias")
        H = wh.shape[1]
        xp = x @ wx.t() + (b if b is not None else 0.0)
        B, T, _ = xp.shape
        hidden, cell = None, torch.zeros(B, H, dtype=xp.dtype)
        outs = []
        for t in range(T):
            hp = torch.zeros(B, 4 * H, dtype=xp.dtype) if hidden is None else hidden @ wh.t()
            g = hp + xp[:, t]
            i, f, gg, o = lstm_sigmoid(g[:, :H]), lstm_sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), lstm_sigmoid(g[:, 3 * H:])
            cell = f * cell + i * gg
            hidden = o * torch.tanh(cell)
            outs.append(hidden)
        return torch.stack(outs, dim=1)

    def resnet(self, x: Tensor, name: str, dim: int, dilations: List[int]) -> Tensor:
        c = self.c
        h = x
        for i, (k, d) in enumerate(zip((c["residual_kernel_size"], 1), dilations)):
            h = self.conv(F.elu(h), f"{name}.block.{2 * i + 1}", k, dilation=d)
        sc = self.conv(x, f"{name}.shortcut", 1) if c["use_conv_shortcut"] else x
        return sc + h

    # ------------------------------------------------------------------ decoder (encodec.py:391-444)
    def decoder(self, z: Tensor, return_stages: bool = False):
        c = self.c
        st = {}
        scaling = int(2 ** len(c["upsampling_ratios"]))
        idx = 0
        h = self.conv(z, f"decoder.layers.{idx}", c["kernel_size"])
        st["conv_in"] = h
        idx += 1
        y = h
        for l in range(c["num_lstm_layers"]):
            y = self.lstm(y, f"decoder.layers.{idx}.lstm.{l}")
        h = y + h
        st["lstm"] = h
        idx += 1
        for bi, ratio in enumerate(c["upsampling_ratios"]):
            cur = scaling * c["num_filters"]
            idx += 1  # nn.ELU()
            h = self.convT(F.elu(h), f"decoder.layers.{idx}", ratio * 2, ratio)
            idx += 1
            for j in range(c["num_residual_layers"]):
                h = self.resnet(h, f"decoder.layers.{idx}", cur // 2, [c["dilation_growth_rate"] ** j, 1])
                idx += 1
            st[f"block{bi}"] = h
            scaling //= 2
        idx += 1  # nn.ELU()
        out = self.conv(F.elu(h), f"decoder.layers.{idx}", c["last_kernel_size"])
        return (out, st) if return_stages else out

    # ------------------------------------------------------------------ quantizer + frames (encodec.py:447-547, 679-777)
    def quantizer_decode(self, codes: Tensor) -> Tensor:
        """codes int [B, nq, T] -> [B, T, codebook_dim]: running sum in codebook order."""
        out = None
        for i in range(codes.shape[1]):
            q = self.w[f"quantizer.layers.{i}.codebook.embed"][codes[:, i].long()]
            out = q if out is None else q + out
        return out

    def decode_frame(self, codes: Tensor, scale: Optional[Tensor] = None) -> Tensor:
        out = self.decoder(self.quantizer_decode(codes))
        return out * scale if scale is not None else out

    @property
    def chunk_length(self):
        return None if self.c["chunk_length_s"] is None else int(self.c["chunk_length_s"] * self.c["sampling_rate"])

    @property
    def chunk_stride(self):
        if self.c["chunk_length_s"] is None or self.c["overlap"] is None:
            return None
        return max(1, int((1.0 - self.c["overlap"]) * self.chunk_length))

    @staticmethod
    def linear_overlap_add(frames: List[Tensor], stride: int) -> Tensor:
        N, fl, C = frames[0].shape
        total = stride * (len(frames) - 1) + frames[-1].shape[1]
        tv = torch.linspace(0, 1, fl + 2, dtype=frames[0].dtype)[1:-1]
        weight = (0.5 - (tv - 0.5).abs())[:, None]
        sw = torch.zeros(total, 1, dtype=frames[0].dtype)
        out = torch.zeros(N, total, C, dtype=frames[0].dtype)
        off = 0
        for fr in frames:
            n = fr.shape[1]
            out[:, off:off + n] += weight[:n] * fr
            sw[off:off + n] += weight[:n]
            off += stride
        return out / sw

    def decode(self, audio_codes: Tensor, audio_scales, padding_mask: Optional[Tensor] = None) -> Tensor:
        """audio_codes int [n_chunks, B, nq, T] (the reference indexes ``audio_codes[:, 0]`` without chunking: [B, 1, nq, T])."""
        if self.chunk_length is None:
            if audio_codes.shape[1] != 1:
                raise ValueError(f"Expected one frame, got {len(audio_codes)}")
            audio = self.decode_frame(audio_codes[:, 0], audio_scales[0])
        else:
            audio = self.linear_overlap_add([self.decode_frame(f, s) for f, s in zip(audio_codes, audio_scales)], self.chunk_stride or 1)
        if padding_mask is not None and padding_mask.shape[1] < audio.shape[1]:
            audio = audio[:, :padding_mask.shape[1]]
        return audio

    # ------------------------------------------------------------------ encoder (encodec.py:340-389)
    def encoder(self, x: Tensor, return_stages: bool = False):
        """x [B, L, channels] -> embeddings [B, T, hidden_size]."""
        c = self.c
        st = {}
        idx = 0
        h = self.conv(x.to(self.dtype), f"encoder.layers.{idx}", c["kernel_size"])
        st["conv_in"] = h
        idx += 1
        scaling = 1
        for bi, ratio in enumerate(reversed(c["upsampling_ratios"])):
            cur = scaling * c["num_filters"]
            for j in range(c["num_residual_layers"]):
                h = self.resnet(h, f"encoder.layers.{idx}", cur, [c["dilation_growth_rate"] ** j, 1])
                idx += 1
            idx += 1  # nn.ELU()
            h = self.conv(F.elu(h), f"encoder.layers.{idx}", ratio * 2, stride=ratio)
            idx += 1
            st[f"block{bi}"] = h
            scaling *= 2
        y = h
        for l in range(c["num_lstm_layers"]):
            y = self.lstm(y, f"encoder.layers.{idx}.lstm.{l}")
        h = y + h
        st["lstm"] = h
        idx += 2  # the LSTM, nn.ELU()
        z = self.conv(F.elu(h), f"encoder.layers.{idx}", c["last_kernel_size"])
        st["embeddings"] = z
        return (z, st) if return_stages else z

    def num_quantizers_for_bandwidth(self, bandwidth: Optional[float]) -> int:
        c = self.c
        frame_rate = math.ceil(c["sampling_rate"] / int(torch.tensor(c["upsampling_ratios"]).prod()))
        tb = c.get("target_bandwidths") or [1.5, 3.0, 6.0, 12.0, 24.0]
        n = int(1000 * tb[-1] // (frame_rate * 10))
        if bandwidth is not None and bandwidth > 0.0:
            n = int(max(1, math.floor(bandwidth * 1000 / (math.log2(c["codebook_size"]) * frame_rate))))
        return n

    def quantizer_encode(self, embeddings: Tensor, bandwidth: Optional[float] = None, return_margins: bool = False):
        """embeddings [B, T, D] -> codes [B, nq, T] (encodec.py:516-533); margins = top-2 gap of ``dist / 2`` (the score scale of mi355_rvq_encode)."""
        residual = embeddings.to(self.dtype)
        codes, margins = [], []
        for i in range(self.num_quantizers_for_bandwidth(bandwidth)):
            if f"quantizer.layers.{i}.codebook.embed" not in self.w:   # ``self.layers[:num_quantizers]``: a slice past the end stops at the last layer
                break
            embed = self.w[f"quantizer.layers.{i}.codebook.embed"]
            flat = residual.reshape(-1, residual.shape[-1])
            dist = -((flat ** 2).sum(1, keepdim=True) - 2 * flat @ embed.t() + (embed.t() ** 2).sum(0, keepdim=True))
            top = torch.topk(dist, 2, dim=1).values
            ind = dist.argmax(-1).reshape(residual.shape[:-1])
            margins.append(((top[:, 0] - top[:, 1]) / 2).reshape(residual.shape[:-1]))
            residual = residual - embed[ind]
            codes.append(ind)
        out = torch.stack(codes, 1)
        return (out, torch.stack(margins, 1)) if return_margins else out

    def encode_frame(self, x: Tensor, bandwidth: float, mask: Tensor):
        c = self.c
        scale = None
        if c["normalize"]:
            x = x * mask[..., None].to(x.dtype)
            mono = x.sum(dim=2, keepdim=True) / x.shape[2]
            scale = torch.sqrt((mono ** 2).mean(dim=1, keepdim=True)) + 1e-8
            x = x / scale
        return self.quantizer_encode(self.encoder(x), bandwidth), scale

    def encode(self, input_values: Tensor, padding_mask: Optional[Tensor] = None, bandwidth: Optional[float] = None):
        """input_values [B, samples, channels] -> (codes [n_chunks, B, nq, T], scales) (encodec.py:585-650)."""
        c = self.c
        tb = c.get("target_bandwidths") or [1.5, 3.0, 6.0, 12.0, 24.0]
        bandwidth = tb[0] if bandwidth is None else bandwidth
        n = input_values.shape[1]
        chunk_length, stride = (n, n) if self.chunk_length is None else (self.chunk_length, self.chunk_stride)
        if padding_mask is None:
            padding_mask = torch.ones(input_values.shape[:2], dtype=torch.bool)
        step = chunk_length - stride
        if (n % stride) != step:
            raise ValueError("The input length is not properly padded for batched chunked encoding. Make sure to pad the input correctly.")
        frames, scales = [], []
        for off in range(0, n - step, stride):
            f, s = self.encode_frame(input_values[:, off:off + chunk_length], bandwidth, padding_mask[:, off:off + chunk_length].bool())
            frames.append(f)
            scales.append(s)
        return torch.stack(frames), scales


EncodecRef = EncodecDecoderRef   # both halves live on the one class
