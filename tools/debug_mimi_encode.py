#!/usr/bin/env python
"""Stage-by-stage comparison of the Mimi encoder engine with the oracle at the real sizes (first layer), per conv and per tile variant."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_audio_amd import ops  # noqa: E402
from mlx_audio_amd.codec.models.mimi import mimi as M  # noqa: E402
from oracle.mimi_ref import MimiConfig as RC  # noqa: E402
from oracle.mimi_ref import MimiEncoderRef  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    cfg = M.mimi_202407(32)
    w = {**M.make_mimi_decoder_weights(cfg, seed=0), **M.make_mimi_encoder_weights(cfg, seed=0)}
    ref = MimiEncoderRef(w, RC(**{k: getattr(cfg, k) for k in RC.__dataclass_fields__}))
    eng = M.MimiEncoder(w, cfg, device=DEV)
    pcm = M.make_pcm(2, 28_800 + 333, seed=3)
    B, S = 2, pcm.shape[-1]
    x_ref = ref._sconv(pcm.transpose(1, 2), "encoder.init_conv1d.conv.conv")
    x0 = pcm.to(DEV).reshape(B, -1).contiguous()
    x = torch.zeros(B, (S + 3) // 4 * 4, cfg.nfilters, device=DEV)
    ops.conv_gemm(x0[:, :, None], eng.init_conv, x, lout=S, flat=dict(ldx=1, x_off=-(eng.init_k - 1), channels=1))
    torch.cuda.synchronize()
    print("init conv", rel(x[:, :S], x_ref))
    lyr = eng.layers[0]
    p = "encoder.layers.0"
    h_ref = ref._sconv(x_ref, p + ".residuals.0.block.0.conv.conv", elu=True)
    y_ref = ref._sconv(h_ref, p + ".residuals.0.block.1.conv.conv", elu=True) + x_ref
    d_ref = ref._sconv(y_ref, p + ".downsample.conv.conv", stride=4, elu=True)
    xr = torch.zeros_like(x)
    xr[:, :S] = x_ref.to(DEV)
    for tile in (0, 64064, 6128064):
        h = torch.zeros(B, x.shape[1], lyr["c0"].cout, device=DEV)
        ops.conv_gemm(xr, lyr["c0"], h, pad=lyr["c0"].k - 1, lout=S, pre_act=ops.ACT_ELU, tile=tile)
        torch.cuda.synchronize()
        print("c0 tile", tile, rel(h[:, :S], h_ref))
    hr = torch.zeros(B, x.shape[1], lyr["c0"].cout, device=DEV)
    hr[:, :S] = h_ref.to(DEV)
    for tile in (0, 64064, 6128064):
        for inplace in (False, True):
            y = xr.clone()
            out = y if inplace else torch.zeros_like(y)
            ops.conv_gemm(hr, lyr["c1"], out, lout=S, pre_act=ops.ACT_ELU, res=y, tile=tile)
            torch.cuda.synchronize()
            print("c1 tile", tile, "inplace", inplace, rel(out[:, :S], y_ref))
    yr = torch.zeros_like(x)
    yr[:, :S] = y_ref.to(DEV)
    Ln = x.shape[1] // 4
    for tile in (0, 64128, 6128128):
        d = torch.zeros(B, Ln, lyr["down"].cout, device=DEV)
        ops.conv_gemm(yr.view(B, Ln, 4 * cfg.nfilters), lyr["down"], d, pad=1, lout=Ln, pre_act=ops.ACT_ELU, tile=tile)
        torch.cuda.synchronize()
        print("down tile", tile, rel(d, d_ref), tuple(d.shape), tuple(d_ref.shape))


if __name__ == "__main__":
    main()
