"""PyTorch-CPU restatement of CSM-1B's ``generate_frame`` (TEST ORACLE, not product).

Follows ``tts/models/sesame/sesame.py`` of the reference:
  * :204-299  create_llama_model_args (llama-1B backbone: 16 L / 2048 / 32-8 heads / d_h 64 / ff 8192; llama-100M depth decoder:
              4 L / 1024 / 8-2 heads / d_h 128 / ff 8192; rms eps 1e-5; rope theta 5e5 with Llama-3 scaling factor 32)
  * :301-343  SesameModel (Identity token embeddings, sesame Attention with Llama3ScaledRoPE, text / audio embedding tables,
              ``projection``, ``codebook0_head``, ``audio_head`` [n_cb - 1, decoder_dim, audio_vocab])
  * :361-404  generate_frame (masked sum of the 32 audio + 1 text embeddings, backbone, c0, then the depth decoder on a fresh cache:
              [last_h, c0_embed] at step 1, one position afterwards, ``decoder_h[:, -1] @ audio_head[i-1]``)
  * :406-425  _embed_audio / _embed_tokens
  * :767      sampler = make_sampler(temp=0.9, top_k=50) (lm/sample_utils.py) -- via oracle.sampling_ref with explicit Gumbel noise
Stacks: oracle.lm_ref.StackRef (interleaved RoPE, lm/models/llama.py:46-198 + sesame/attention.py:11-175).
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files for
CSM (imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) on a seeded tiny checkpoint, and
tests/test_reference_fixtures_cpu.py holds this oracle to the result -- ``SesameModel.generate_frame`` for three frames with a forcing sampler (backbone + depth decoder logits): 3e-5.
The reference's own tests pin shapes / token-rule cases only (reproduced in tests/test_oracle_golden.py); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import sampling_ref
from .lm_ref import StackConfig, StackRef

Tensor = torch.Tensor


@dataclass
class CSMConfig:
    backbone: StackConfig
    decoder: StackConfig
    audio_vocab_size: int = 2051
    audio_num_codebooks: int = 32
    text_vocab_size: int = 128256


def llama_stack(hidden, layers, heads, kv, dh, ff) -> StackConfig:
    return StackConfig(d_model=hidden, n_layers=layers, n_heads=heads, n_kv_heads=kv, head_dim=dh, d_ff=ff, norm="rms", norm_eps=1e-5,
                       rope_theta=500000.0, rope_interleaved=True, rope_llama3_factor=32.0, max_pos=2048, mlp="swiglu")


def csm_1b() -> CSMConfig:
    return CSMConfig(backbone=llama_stack(2048, 16, 32, 8, 64, 8192), decoder=llama_stack(1024, 4, 8, 2, 128, 8192))


class CSMRef:
    def __init__(self, weights: Dict[str, Tensor], cfg: CSMConfig, dtype=torch.float32, param_dtype=torch.bfloat16):
        self.cfg = cfg
        self.w = {k: v.to(param_dtype).to(dtype) for k, v in weights.items()}
        sub = lambda p: {k[len(p):]: v for k, v in weights.items() if k.startswith(p)}  # noqa: E731
        self.backbone = StackRef(sub("backbone."), cfg.backbone, dtype, param_dtype)
        self.decoder = StackRef(sub("decoder."), cfg.decoder, dtype, param_dtype)
        self.backbone_cache = self.backbone.make_cache()

    def reset_caches(self):
        self.backbone_cache = self.backbone.make_cache()

    def _embed_audio(self, codebook: int, tokens: Tensor) -> Tensor:
        return self.w["audio_embeddings.weight"][tokens + codebook * self.cfg.audio_vocab_size]

    def _embed_tokens(self, tokens: Tensor) -> Tensor:
        c = self.cfg
        text = self.w["text_embeddings.weight"][tokens[:, :, -1]][:, :, None, :]
        offs = torch.arange(c.audio_num_codebooks) * c.audio_vocab_size
        audio = self.w["audio_embeddings.weight"][tokens[:, :, :-1] + offs.reshape(1, 1, -1)]
        return torch.cat([audio, text], dim=-2)

    def generate_frame(self, tokens: Tensor, tokens_mask: Tensor, *, temperature=0.9, top_k=50, gumbel=None, forced=None, trace=None) -> Tensor:
        """tokens int [B, S, n_cb + 1], tokens_mask bool [B, S, n_cb + 1]; gumbel [n_cb, B, audio_vocab] -> sample int [B, n_cb]."""
        c = self.cfg
        embeds = self._embed_tokens(tokens)
        h = (embeds * tokens_mask[..., None].to(embeds.dtype)).sum(dim=2)
        h = self.backbone(h, self.backbone_cache)
        last_h = h[:, -1, :]
        c0_logits = F.linear(last_h, self.w["codebook0_head.weight"])
        if trace is not None:
            trace.append(c0_logits)

        def draw(logits, i):
            s = sampling_ref.sample(logits, None if gumbel is None else gumbel[i], temperature=temperature, top_k=top_k, top_p=1.0,
                                    repetition_penalty=1.0)
            return s if forced is None else forced[:, i]

        c0 = draw(c0_logits, 0)
        curr_h = torch.stack([last_h, self._embed_audio(0, c0)], dim=1)
        samples = [c0]
        cache = self.decoder.make_cache()  # reset decoder cache for new frame
        for i in range(1, c.audio_num_codebooks):
            dh = self.decoder(F.linear(curr_h, self.w["projection.weight"]), cache)
            ci_logits = dh[:, -1, :] @ self.w["audio_head"][i - 1]
            if trace is not None:
                trace.append(ci_logits)
            ci = draw(ci_logits, i)
            curr_h = self._embed_audio(i, ci)[:, None, :]
            samples.append(ci)
        return torch.stack(samples, dim=1)

    def generate(self, prompt_tokens: Tensor, prompt_mask: Tensor, max_frames: int, **kw):
        """The frame loop of Model.generate (sesame.py:813-846) for one prompt: returns frames [B, n, n_cb]; stops on an all-zero frame."""
        self.reset_caches()
        toks, mask = prompt_tokens, prompt_mask
        gum = kw.pop("gumbel", None)
        forced = kw.pop("forced", None)
        frames, traces = [], []
        for f in range(max_frames):
            tr = [] if kw.get("record") else None
            s = self.generate_frame(toks, mask, temperature=kw.get("temperature", 0.9), top_k=kw.get("top_k", 50),
                                    gumbel=None if gum is None else gum[f], forced=None if forced is None else forced[:, f], trace=tr)
            if bool((s == 0).all()):
                break
            frames.append(s)
            traces.append(tr)
            B = s.shape[0]
            toks = torch.cat([s, torch.zeros(B, 1, dtype=s.dtype)], dim=1)[:, None, :]
            mask = torch.cat([torch.ones_like(s, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool)], dim=1)[:, None, :]
        return dict(frames=torch.stack(frames, dim=1) if frames else torch.zeros(prompt_tokens.shape[0], 0, self.cfg.audio_num_codebooks, dtype=torch.long),
                    trace=traces)
