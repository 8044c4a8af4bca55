"""PyTorch-CPU restatement of the Qwen3-TTS frame loop: talker step, first-codebook sampling, 15-step code predictor, next-input
embedding (TEST ORACLE, not product).

Follows ``tts/models/qwen3_tts`` of the reference:
  * ``talker.py:403-500, 766-822``  Qwen3TTSTalkerModel / ForConditionalGeneration (stack + ``codec_head``; ``text_projection`` ResizeMLP :333-363)
  * ``talker.py:615-764``           CodePredictorModel / Qwen3TTSTalkerCodePredictor (``small_to_mtp_projection`` when the talker width differs
                                    from the predictor's, per-step ``lm_head[i]`` and ``codec_embedding[i]``)
  * ``qwen3_tts.py:941-983``        _predict_code_tokens (step 0 feeds [last_hidden, embed(code0)], later steps one position; fresh cache per frame)
  * ``qwen3_tts.py:985-1015``       _codec_embeds_for_tokens / _next_batch_input_embeds (text embed or tts_pad + sum of 16 codec embeds)
  * ``qwen3_tts.py:1860-1935``      the batched generation loop (finished rows emit EOS, history of non-finished rows feeds the repetition
                                    penalty, suppress ids [vocab-1024, vocab) except EOS :927-933)
Sampling is ``oracle.sampling_ref`` with explicit Gumbel noise.  Prompt construction (:359-604) is tokenizer-side host logic and not part of
this restatement: the loop starts from prefill embeddings.
Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files for
the Qwen3-TTS talker (imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) on a seeded tiny checkpoint, and
tests/test_reference_fixtures_cpu.py holds this oracle to the result -- talker stack (prefill + cached steps), ``codec_head``, ``text_projection`` and the code predictor stepped like ``_predict_code_tokens`` on forced codes: 2e-5.
The reference's own tests pin shapes / token-rule cases only (reproduced in tests/test_oracle_golden.py); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import sampling_ref
from .lm_ref import StackConfig, StackRef

Tensor = torch.Tensor


def talker_stack_config(c) -> StackConfig:
    return StackConfig(d_model=c.hidden_size, n_layers=c.num_hidden_layers, n_heads=c.num_attention_heads, n_kv_heads=c.num_key_value_heads,
                       head_dim=c.head_dim, d_ff=c.intermediate_size, norm="rms", norm_eps=c.rms_norm_eps, qk_norm=True, rope_theta=c.rope_theta,
                       max_pos=min(c.max_position_embeddings, 8192), attn_bias=c.attention_bias, mlp="swiglu")


def canonical(w: Dict[str, Tensor], prefix: str, n_layers: int) -> Dict[str, Tensor]:
    m = {"self_attn.q_proj": "wq", "self_attn.k_proj": "wk", "self_attn.v_proj": "wv", "self_attn.o_proj": "wo", "mlp.gate_proj": "w_gate",
         "mlp.up_proj": "w_up", "mlp.down_proj": "w_down", "input_layernorm": "attn_norm", "post_attention_layernorm": "mlp_norm",
         "self_attn.q_norm": "q_norm", "self_attn.k_norm": "k_norm"}
    out = {}
    for i in range(n_layers):
        for src, dst in m.items():
            for suf in ("weight", "bias"):
                k = f"{prefix}layers.{i}.{src}.{suf}"
                if k in w:
                    out[f"layers.{i}.{dst}.{suf}"] = w[k]
    out["final_norm.weight"] = w[prefix + "norm.weight"]
    return out


class Qwen3TalkerRef:
    def __init__(self, weights: Dict[str, Tensor], cfg, dtype=torch.float32, param_dtype=torch.bfloat16):
        self.cfg = cfg
        cp = cfg.code_predictor_config
        self.w = {k: v.to(param_dtype).to(dtype) for k, v in weights.items()}
        self.talker = StackRef(canonical(weights, "model.", cfg.num_hidden_layers), talker_stack_config(cfg), dtype, param_dtype)
        self.cp = StackRef(canonical(weights, "code_predictor.model.", cp.num_hidden_layers), talker_stack_config(cp), dtype, param_dtype)
        self.n_groups = cfg.num_code_groups

    def text_projection(self, x: Tensor) -> Tensor:
        """ResizeMLP (talker.py:333-363), hidden_act silu."""
        h = F.silu(F.linear(x, self.w["text_projection.linear_fc1.weight"], self.w["text_projection.linear_fc1.bias"]))
        return F.linear(h, self.w["text_projection.linear_fc2.weight"], self.w["text_projection.linear_fc2.bias"])

    def suppress_tokens(self) -> List[int]:
        c = self.cfg
        return [i for i in range(c.vocab_size - 1024, c.vocab_size) if i != c.codec_eos_token_id]

    def predict_codes(self, first: Tensor, hidden_last: Tensor, *, temperature, top_k, top_p, gumbel=None, forced=None, trace=None):
        """qwen3_tts.py:941-983.  first [B] long, hidden_last [B, H] -> codes [B, n_groups]."""
        cache = self.cp.make_cache()
        toks = [first]
        for i in range(self.n_groups - 1):
            if i == 0:
                x = torch.stack([hidden_last, self.w["model.codec_embedding.weight"][first]], dim=1)
            else:
                x = self.w[f"code_predictor.model.codec_embedding.{i - 1}.weight"][toks[-1]][:, None, :]
            if "code_predictor.small_to_mtp_projection.weight" in self.w:
                x = F.linear(x, self.w["code_predictor.small_to_mtp_projection.weight"], self.w["code_predictor.small_to_mtp_projection.bias"])
            h = self.cp(x, cache)
            logits = F.linear(h[:, -1], self.w[f"code_predictor.lm_head.{i}.weight"])
            if trace is not None:
                trace.append(logits)
            nxt = sampling_ref.sample(logits, None if gumbel is None else gumbel[i], temperature=temperature, top_k=top_k, top_p=top_p,
                                      repetition_penalty=1.0)
            if forced is not None:
                nxt = forced[:, i + 1]
            toks.append(nxt)
        return torch.stack(toks, dim=1)

    def codec_embeds(self, codes: Tensor) -> Tensor:
        """qwen3_tts.py:985-992: sum of the 16 codec embeddings, [B, n_groups] -> [B, H]."""
        e = self.w["model.codec_embedding.weight"][codes[:, 0]]
        for i in range(1, codes.shape[1]):
            e = e + self.w[f"code_predictor.model.codec_embedding.{i - 1}.weight"][codes[:, i]]
        return e

    def next_input(self, trailing: Tensor, trailing_idx: Tensor, tts_pad: Tensor, codes: Tensor, pad_when_index_clamped: bool) -> Tensor:
        """``_next_batch_input_embeds`` (qwen3_tts.py:993-1015): the next position's input [B, 1, H] = trailing text at ``trailing_idx`` (or ``tts_pad`` once
        the text is exhausted) + the summed embeddings of the frame's codes.  The batched loop pads from the LAST trailing position on
        (``pad_when_index_clamped``), the single-utterance loop (:1388-1394) after it."""
        B, Tt = trailing.shape[0], trailing.shape[1]
        clamped = torch.clamp(trailing_idx, max=Tt - 1)
        text = trailing[torch.arange(B), clamped]
        exhausted = (clamped >= Tt - 1) if pad_when_index_clamped else (trailing_idx >= Tt)
        text = torch.where(exhausted[:, None], tts_pad.reshape(1, -1).expand_as(text), text)
        return (text + self.codec_embeds(codes))[:, None, :]

    def generate(self, prefill: Tensor, trailing: Tensor, tts_pad: Tensor, max_frames: int, *, temperature=0.9, top_k=50, top_p=1.0,
                 repetition_penalty=1.05, gumbel0=None, gumbel_cp=None, forced_codes=None, record=False, pad_when_index_clamped=True):
        """prefill [B, L, H] embeddings; trailing [B, Tt, H] text embeddings consumed one per frame, then ``tts_pad`` [1, 1, H].
        gumbel0 [frames, B, V], gumbel_cp [frames, n_groups-1, B, Vcp] (None => arg-max).  Returns dict(codes [B, frames, n_groups],
        finished_at [B] (frame index of the EOS, or -1), trace)."""
        c = self.cfg
        B = prefill.shape[0]
        cache = self.talker.make_cache()
        finished = torch.zeros(B, dtype=torch.bool)
        finished_at = torch.full((B,), -1, dtype=torch.long)
        hist: List[List[int]] = [[] for _ in range(B)]
        trailing_idx = torch.zeros(B, dtype=torch.long)
        x = prefill
        out, trace = [], []
        sup = self.suppress_tokens()
        for f in range(max_frames):
            h = self.talker(x, cache)
            last = h[:, -1]
            logits = F.linear(last, self.w["codec_head.weight"])
            tr = [logits] if record else None
            tok = sampling_ref.sample(logits, None if gumbel0 is None else gumbel0[f], temperature=temperature, top_k=top_k, top_p=top_p,
                                      repetition_penalty=repetition_penalty, generated=hist, suppress_tokens=sup)
            if forced_codes is not None:
                tok = forced_codes[:, f, 0]
            tok = torch.where(finished, torch.full_like(tok, c.codec_eos_token_id), tok)
            newly = tok == c.codec_eos_token_id
            finished_at = torch.where(newly & ~finished, torch.full_like(finished_at, f), finished_at)
            finished = finished | newly
            codes = self.predict_codes(tok, last, temperature=temperature, top_k=top_k, top_p=top_p,
                                       gumbel=None if gumbel_cp is None else gumbel_cp[f],
                                       forced=None if forced_codes is None else forced_codes[:, f], trace=tr)
            x = self.next_input(trailing, trailing_idx, tts_pad, codes, pad_when_index_clamped)
            trailing_idx = trailing_idx + (~finished).long()
            for b in range(B):
                if not bool(finished[b]):
                    hist[b].append(int(tok[b]))
            out.append(codes)
            if record:
                trace.append(tr)
            if bool(finished.all()) and forced_codes is None:
                break
        return dict(codes=torch.stack(out, dim=1), finished_at=finished_at, trace=trace)
