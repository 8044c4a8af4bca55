"""Qwen3-TTS talker + code predictor frame loop on MI355X: host schedule over the HIP kernels.

Mirrors the batched generation loop of the reference (``tts/models/qwen3_tts/qwen3_tts.py:1860-1935``) and the modules under it
(``talker.py:230-822``): per 80 ms frame one talker step, the first-codebook sampling chain (suppress -> repetition penalty -> temperature
-> top-k -> top-p -> categorical), 15 code-predictor steps on a fresh KV cache, and the next input embedding (text embed or tts_pad + sum of
the 16 codec embeddings).  Every step is 1 row per sequence: Linear layers run as the HBM-bound ``mi355_gemv`` on row-major bf16 weights
(SwiGLU fused), attention is the KV-streaming kernel, the whole sampling chain is ONE kernel per token, and the per-frame bookkeeping
(finished mask, repetition-penalty history, trailing-text index) lives in device tensors -- the reference's per-frame ``mx.eval`` +
``.tolist()`` round trip (qwen3_tts.py:1911-1914) is gone; completion is polled every ``poll`` frames.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .... import ops
from ....lm.stack import Lin, StackConfig, TransformerStack, linear, make_lin
from ....ops import ACT_SILU
from .config import Qwen3TTSTalkerConfig


def talker_stack_config(c) -> StackConfig:
    return StackConfig(d_model=c.hidden_size, n_layers=c.num_hidden_layers, n_heads=c.num_attention_heads, n_kv_heads=c.num_key_value_heads,
                       head_dim=c.head_dim, d_ff=c.intermediate_size, norm="rms", norm_eps=c.rms_norm_eps, qk_norm=True, rope_theta=c.rope_theta,
                       max_pos=min(c.max_position_embeddings, 8192), attn_bias=c.attention_bias, mlp="swiglu")


def canonical(w: Dict[str, torch.Tensor], prefix: str, n_layers: int) -> Dict[str, torch.Tensor]:
    """``<prefix>layers.N.self_attn.q_proj`` ... (talker.py module paths after sanitize :825-837) -> canonical stack names."""
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


def tiny_talker_config() -> Qwen3TTSTalkerConfig:
    """Structurally identical (GQA, q/k norm, RoPE, SwiGLU, talker width != predictor width => small_to_mtp_projection, 4 code groups)."""
    from .config import Qwen3TTSTalkerCodePredictorConfig

    cp = Qwen3TTSTalkerCodePredictorConfig(vocab_size=96, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                           num_key_value_heads=1, head_dim=64, num_code_groups=4)
    return Qwen3TTSTalkerConfig(code_predictor_config=cp, vocab_size=1200, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1, head_dim=128, num_code_groups=4, text_hidden_size=192,
                                text_vocab_size=500, codec_eos_token_id=1150)


def make_talker_weights(cfg: Qwen3TTSTalkerConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic talker + code-predictor parameters (reference module paths, bf16-representable float32)."""
    import math

    from ....lm.synthetic import make_stack_weights

    g = torch.Generator().manual_seed(seed)
    cp = cfg.code_predictor_config
    w: Dict[str, torch.Tensor] = {}

    def r16(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def rnd(*shape, std):
        return r16(torch.randn(*shape, generator=g) * std)

    inv = {"wq": "self_attn.q_proj", "wk": "self_attn.k_proj", "wv": "self_attn.v_proj", "wo": "self_attn.o_proj", "w_gate": "mlp.gate_proj",
           "w_up": "mlp.up_proj", "w_down": "mlp.down_proj", "attn_norm": "input_layernorm", "mlp_norm": "post_attention_layernorm",
           "q_norm": "self_attn.q_norm", "k_norm": "self_attn.k_norm"}
    for prefix, c, sd in (("model.", cfg, seed * 7 + 1), ("code_predictor.model.", cp, seed * 7 + 2)):
        sw = make_stack_weights(talker_stack_config(c), seed=sd, gain=1.5)
        for k, v in sw.items():
            if k.startswith("final_norm"):
                w[prefix + "norm." + k.split(".", 1)[1]] = v
            else:
                _, i, name, suf = k.split(".")
                w[f"{prefix}layers.{i}.{inv[name]}.{suf}"] = v
    H = cfg.hidden_size
    w["model.codec_embedding.weight"] = rnd(cfg.vocab_size, H, std=0.5)
    w["model.text_embedding.weight"] = rnd(cfg.text_vocab_size, cfg.text_hidden_size, std=0.5)
    w["text_projection.linear_fc1.weight"] = rnd(cfg.text_hidden_size, cfg.text_hidden_size, std=1.0 / math.sqrt(cfg.text_hidden_size))
    w["text_projection.linear_fc1.bias"] = rnd(cfg.text_hidden_size, std=0.02)
    w["text_projection.linear_fc2.weight"] = rnd(H, cfg.text_hidden_size, std=1.0 / math.sqrt(cfg.text_hidden_size))
    w["text_projection.linear_fc2.bias"] = rnd(H, std=0.02)
    w["codec_head.weight"] = rnd(cfg.vocab_size, H, std=4.0 / math.sqrt(H))
    if cp.hidden_size != H:
        w["code_predictor.small_to_mtp_projection.weight"] = rnd(cp.hidden_size, H, std=1.0 / math.sqrt(H))
        w["code_predictor.small_to_mtp_projection.bias"] = rnd(cp.hidden_size, std=0.02)
    for i in range(cfg.num_code_groups - 1):
        w[f"code_predictor.model.codec_embedding.{i}.weight"] = rnd(cp.vocab_size, H, std=0.5)
        w[f"code_predictor.lm_head.{i}.weight"] = rnd(cp.vocab_size, cp.hidden_size, std=4.0 / math.sqrt(cp.hidden_size))
    return w


class Qwen3Talker:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: Qwen3TTSTalkerConfig, device="cuda:0", precision: int = 2):
        ops.require_gpu()
        self.cfg = cfg
        cp = cfg.code_predictor_config
        self.device = torch.device(device)
        self.precision = precision
        dev = self.device
        w = {k: v.detach().to(torch.bfloat16).to(torch.float32).cpu() for k, v in weights.items()}
        self.talker = TransformerStack(canonical(w, "model.", cfg.num_hidden_layers), talker_stack_config(cfg), device=dev, precision=precision)
        self.cp = TransformerStack(canonical(w, "code_predictor.model.", cp.num_hidden_layers), talker_stack_config(cp), device=dev, precision=precision)
        self.codec_head = make_lin(w["codec_head.weight"], None, dev)
        self.lm_heads = [make_lin(w[f"code_predictor.lm_head.{i}.weight"], None, dev) for i in range(cfg.num_code_groups - 1)]
        self.mtp: Optional[Lin] = None
        if "code_predictor.small_to_mtp_projection.weight" in w:
            self.mtp = make_lin(w["code_predictor.small_to_mtp_projection.weight"], w["code_predictor.small_to_mtp_projection.bias"], dev)
        self.fc1 = make_lin(w["text_projection.linear_fc1.weight"], w["text_projection.linear_fc1.bias"], dev)
        self.fc2 = make_lin(w["text_projection.linear_fc2.weight"], w["text_projection.linear_fc2.bias"], dev)
        self.text_embedding = w["model.text_embedding.weight"].to(dev)
        # one stacked codec-embedding table: slot 0 = talker codec_embedding, slots 1.. = code_predictor.codec_embedding[i]
        tabs = [w["model.codec_embedding.weight"]] + [w[f"code_predictor.model.codec_embedding.{i}.weight"] for i in range(cfg.num_code_groups - 1)]
        offs, r = [], 0
        for t in tabs:
            offs.append(r)
            r += t.shape[0]
        self.codec_table = torch.cat(tabs, 0).contiguous().to(dev)
        self.codec_offs = torch.tensor(offs, dtype=torch.int32, device=dev)
        self.codec_offs_host = offs
        sup = torch.zeros(cfg.vocab_size)
        sup[[i for i in range(cfg.vocab_size - 1024, cfg.vocab_size) if i != cfg.codec_eos_token_id]] = -float("inf")
        self.suppress_mask = sup.to(dev)  # qwen3_tts.py:927-933

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def embed_text(self, ids: torch.Tensor) -> torch.Tensor:
        """text_projection(text_embedding(ids)) (talker.py:333-363, 786-793): int [B, L] -> [B, L, hidden]."""
        ids = ids.to(self.device, torch.int32).contiguous()
        B, L = ids.shape
        e = self._f(B, L, self.cfg.text_hidden_size)
        ops.gather_rows(self.text_embedding, ids, e)
        h = self._f(B, L, self.cfg.text_hidden_size)
        linear(e, self.fc1, h, post_act=ACT_SILU, precision=self.precision)
        out = self._f(B, L, self.cfg.hidden_size)
        linear(h, self.fc2, out, precision=self.precision)
        return out

    def embed_codes(self, codes: torch.Tensor) -> torch.Tensor:
        """Sum over the code groups of the codec embeddings of whole frames (group 0: the talker's ``codec_embedding``, group i >= 1:
        ``code_predictor.codec_embedding[i - 1]``): int ``[B, T, G]`` -> ``[B, T, hidden]`` -- the reference-clip half of an in-context prompt
        (qwen3_tts.py:701-709), one gather-and-sum launch."""
        ids = codes.to(self.device, torch.int32).contiguous()
        B, T, G = ids.shape
        assert G == self.cfg.num_code_groups, (G, self.cfg.num_code_groups)
        out = self._f(B, T, self.cfg.hidden_size)
        ops.embed_sum(self.codec_table, ids, out, slot_offset=self.codec_offs)
        return out

    def _logits(self, h_last: torch.Tensor, head: Lin, norm=None) -> torch.Tensor:
        """h_last [B, 1, C] -> [B, V_padded] fp32 (``norm``: the producing stack's deferred final norm, applied in the GEMV prologue)."""
        B = h_last.shape[0]
        V = head.rm.n
        out = self._f(B, 1, ops.round_up(V, 4))
        linear(h_last, head, out[:, :, :V], precision=self.precision, norm=norm)
        return out[:, 0, :]

    def generate(self, prefill: torch.Tensor, trailing: torch.Tensor, tts_pad: torch.Tensor, max_frames: int, **kw):
        """The frame loop run to its end: the last item of ``generate_iter`` (dict codes / finished_at / trace)."""
        out = None
        for out in self.generate_iter(prefill, trailing, tts_pad, max_frames, **kw):
            pass
        return out

    def generate_iter(self, prefill: torch.Tensor, trailing: torch.Tensor, tts_pad: torch.Tensor, max_frames: int, *, temperature: float = 0.9,
                      top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.05, gumbel0: Optional[torch.Tensor] = None,
                      gumbel_cp: Optional[torch.Tensor] = None, forced_codes: Optional[torch.Tensor] = None, record: bool = False, poll: int = 16,
                      left_pad: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, pad_when_index_clamped: bool = True,
                      chunk: int = 0):
        """The frame loop as a generator.  ``chunk`` > 0 (one sequence: qwen3_tts.py:1426-1465 ``stream=True``): every ``chunk`` frames a dict
        ``block`` = int64 [1, n, G] (the frames generated since the last block, the EOS frame excluded) is yielded WHILE the loop runs, so the caller
        decodes and hands on audio before the next frame exists (one host read-back per block; ``self.frames_generated`` = frames computed so far).
        The LAST item is always the final dict (``codes`` / ``finished_at`` / ``trace``) ``generate`` returns.

        The frame loop (qwen3_tts.py:1860-1935).  ``prefill`` [B, L, H] may be LEFT-padded (``left_pad`` int [B] = padding positions of each
        row, qwen3_tts.py:536-560): padded keys are invisible and positions count from the first real token, exactly what the reference's
        attention-mask / cumsum position ids do (talker.py:443-470).  Sampling noise: explicit Gumbel tensors (``gumbel0`` / ``gumbel_cp``, the
        parity tests), else -- for temperature > 0 -- Gumbel noise drawn on the device from ``generator`` (``mx.random.categorical`` in the
        reference, qwen3_tts.py:805-925: a different RNG stream, the same distribution); neither = greedy.
        ``pad_when_index_clamped``: the batched loop's rule for the trailing text (qwen3_tts.py:993-1015, 1894-1900): the LAST trailing position
        already reads as tts_pad (True, ``batch_generate``); the single-utterance loop feeds every trailing position and pads after it (False,
        ``generate`` :1388-1394)."""
        cfg = self.cfg
        cp = cfg.code_predictor_config
        dev = self.device
        G = cfg.num_code_groups
        x = prefill.to(dev, torch.float32).contiguous().clone()
        trailing = trailing.to(dev, torch.float32).contiguous()
        pad = tts_pad.to(dev, torch.float32).reshape(1, -1)
        B, _, H = x.shape
        Tt = trailing.shape[1]
        cache = self.talker.make_cache()
        cp_cache = self.cp.make_cache()
        k_start = None if left_pad is None else left_pad.to(dev, torch.int32).contiguous()
        if x.shape[1] + max_frames > self.talker.cos.shape[0]:
            raise ValueError(f"prompt ({x.shape[1]} positions) + max_frames ({max_frames}) exceeds the talker's {self.talker.cos.shape[0]} rotary positions")
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        finished_at = torch.full((B,), -1, dtype=torch.int64, device=dev)
        hist = torch.full((B, max_frames + 1), -1, dtype=torch.int32, device=dev)
        hist_len = torch.zeros(B, dtype=torch.int32, device=dev)
        trailing_idx = torch.zeros(B, dtype=torch.int64, device=dev)
        codes_all = torch.zeros((B, max_frames, G), dtype=torch.int32, device=dev)
        ar = torch.arange(B, device=dev)
        forced = None if forced_codes is None else forced_codes.to(dev, torch.int32)
        V0, Vc = cfg.vocab_size, cp.vocab_size

        draw = generator is not None and temperature > 0

        def noise(t, V):
            if t is None:
                if not draw:
                    return None
                e = torch.empty((B, ops.round_up(V, 4)), dtype=torch.float32, device=dev).exponential_(generator=generator)
                return -torch.log(e)  # -log(Exp(1)) is Gumbel(0, 1)
            n = torch.zeros((B, ops.round_up(V, 4)), dtype=torch.float32, device=dev)
            n[:, :V] = t.to(dev, torch.float32)
            return n

        trace: List[list] = []
        frames = 0
        sent = 0
        self.frames_generated = 0
        assert chunk == 0 or B == 1, "chunked (streaming) frame loop: one sequence"
        for f in range(max_frames):
            h = self.talker(x, cache, k_start=k_start)
            last = h[:, -1:, :].contiguous()
            logits = self._logits(last, self.codec_head)
            tr = [logits[:, :V0].clone()] if record else None
            row = codes_all[:, f, :]
            ops.sample(logits, row[:, 0], V=V0, suppress_mask=self.suppress_mask, history=hist, hist_len=hist_len,
                       repetition_penalty=repetition_penalty, temperature=temperature, top_k=top_k, top_p=top_p,
                       gumbel=noise(None if gumbel0 is None else gumbel0[f], V0), done=finished, done_token=cfg.codec_eos_token_id)
            if forced is not None:   # teacher forcing; a negative entry keeps the step's own selection (partial forcing: tests re-synchronise at knife edges)
                row[:, 0] = torch.where(finished.bool(), torch.full_like(forced[:, f, 0], cfg.codec_eos_token_id),
                                        torch.where(forced[:, f, 0] >= 0, forced[:, f, 0], row[:, 0]))
            tok = row[:, 0]
            newly = tok == cfg.codec_eos_token_id
            finished_at = torch.where(newly & (finished == 0), torch.full_like(finished_at, f), finished_at)
            finished = finished | newly.to(torch.int32)
            # ---- code predictor: 15 steps on a fresh cache (qwen3_tts.py:941-983)
            for c in cp_cache:
                c.reset()
            for i in range(G - 1):
                if i == 0 and B <= self.cp.max_decode_rows:
                    # step 0 feeds two positions, [last_hidden, embed(code0)] (qwen3_tts.py:961-966).  With a causal stack and a KV cache
                    # that is exactly two single-position steps, and single-position steps run on the GEMV / KV-streaming path instead
                    # of a 2-row MFMA tile; only the second position's output is used.
                    x0 = last.clone()
                    if self.mtp is not None:
                        xp = self._f(B, 1, cp.hidden_size)
                        linear(x0, self.mtp, xp, precision=self.precision)
                        x0 = xp
                    self.cp(x0, cp_cache)
                    xin = self._f(B, 1, H)
                    ops.embed_sum(self.codec_table, row[:, 0:1].unsqueeze(1), xin)
                elif i == 0:
                    xin = self._f(B, 2, H)
                    xin[:, 0:1, :] = last
                    ops.embed_sum(self.codec_table, row[:, 0:1].unsqueeze(1), xin[:, 1:2, :])
                else:
                    xin = self._f(B, 1, H)
                    ops.embed_sum(self.codec_table, row[:, i:i + 1].unsqueeze(1), xin, slot_offset=self.codec_offs[i:i + 1])
                if self.mtp is not None:
                    xp = self._f(B, xin.shape[1], cp.hidden_size)
                    linear(xin, self.mtp, xp, precision=self.precision)
                    xin = xp
                if xin.shape[1] == 1 and B <= self.cp.max_decode_rows and self.cp.native_decode and (B > 8 or self.cp.cfg.d_model <= 2048):
                    # the code predictor's final RMSNorm runs inside the lm_head GEMV (fused-norm prologue): one launch less per code group
                    hc = self.cp(xin, cp_cache, defer_final_norm=True)
                    lg = self._logits(hc, self.lm_heads[i], norm=self.cp.final_norm_arg())
                else:
                    hc = self.cp(xin, cp_cache)
                    lg = self._logits(hc[:, -1:, :].contiguous(), self.lm_heads[i])
                if record:
                    tr.append(lg[:, :Vc].clone())
                ops.sample(lg, row[:, i + 1], V=Vc, temperature=temperature, top_k=top_k, top_p=top_p,
                           gumbel=noise(None if gumbel_cp is None else gumbel_cp[f][i], Vc))
                if forced is not None:
                    row[:, i + 1] = torch.where(forced[:, f, i + 1] >= 0, forced[:, f, i + 1], row[:, i + 1])
            # ---- next input: text embed (or tts_pad once the trailing text is exhausted) + sum of the 16 codec embeddings
            clamped = torch.clamp(trailing_idx, max=Tt - 1)
            text = trailing[ar, clamped]
            exhausted = (clamped >= Tt - 1) if pad_when_index_clamped else (trailing_idx >= Tt)
            text = torch.where(exhausted[:, None], pad.expand_as(text), text)
            nx = self._f(B, 1, H)
            ops.embed_sum(self.codec_table, row.unsqueeze(1), nx, slot_offset=self.codec_offs, add=text[:, None, :].contiguous())
            x = nx
            alive = finished == 0
            trailing_idx = trailing_idx + alive.to(torch.int64)
            hist[ar, hist_len.long()] = torch.where(alive, tok, hist[ar, hist_len.long()])
            hist_len = hist_len + alive.to(torch.int32)
            frames = f + 1
            self.frames_generated = frames
            if record:
                trace.append(tr)
            if chunk and frames - sent >= chunk:   # streaming: hand the finished frames on now (one read-back per block)
                fa0 = int(finished_at[0])
                end = frames if fa0 < 0 else fa0
                if end > sent:
                    yield dict(block=codes_all[:, sent:end].to(torch.int64), first_frame=sent, last=fa0 >= 0 or frames == max_frames)
                sent = end
                if fa0 >= 0:
                    break
                continue
            if not chunk and forced is None and frames % poll == 0 and bool((finished != 0).all()):  # the only host round trip of the loop
                break
        codes = codes_all[:, :frames].to(torch.int64)
        if chunk:   # the tail of a streamed sequence
            fa0 = int(finished_at[0])
            end = frames if fa0 < 0 else fa0
            if end > sent:
                yield dict(block=codes_all[:, sent:end].to(torch.int64), first_frame=sent, last=True)
        if forced is None:
            fa = finished_at.cpu()
            if bool((fa >= 0).all()):
                codes = codes[:, : int(fa.max()) + 1]  # the reference stops at the frame where the last sequence emits EOS
        yield dict(codes=codes, finished_at=finished_at, trace=trace)


class Qwen3TalkerSlots:
    """Device-resident state of a continuous-batching session (``tts/models/qwen3_tts/continuous_batching.py:37-360``) over SLOT KV caches.

    The reference keeps one KV cache per request and, at EVERY step, merges the active requests' caches into a left-padded batch cache
    (``KVCache.merge``) and splits it again afterwards (``extract``): O(KV) copies twice per generated frame.  Here a request owns a
    slot = one row of every layer's ``[slots, capacity, 2 * kv_heads * dh]`` buffer for its whole life: admission prefills the newcomers as one
    left-padded batch and files each prompt's keys / values into its slot once; every later step appends in place -- the native step runner takes
    the per-slot lengths (``mi355_stack_desc.slot_lens_k``), so sequences of different lengths share one launch sequence without any padding,
    merging or extraction.  Slots without a request are carried as finished rows (their sampler emits EOS, nothing of their state moves).
    The per-frame arithmetic is ``Qwen3Talker.generate``'s (one talker step, the first-codebook sampling chain with the request's own history,
    15 code-predictor steps on a fresh cache, next input = trailing text or tts_pad + the 16 codec embeddings)."""

    def __init__(self, eng: "Qwen3Talker", n_slots: int, max_frames: int, *, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                 repetition_penalty: float = 1.05, generator: Optional[torch.Generator] = None):
        assert 1 <= n_slots <= eng.talker.max_decode_rows, f"at most {eng.talker.max_decode_rows} slots per step"
        self.eng, self.S, self.max_frames = eng, n_slots, max_frames
        self.kw = dict(temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty)
        self.generator = generator
        dev, cfg = eng.device, eng.cfg
        S, H, G = n_slots, cfg.hidden_size, cfg.num_code_groups
        self.lens = torch.zeros(S, dtype=torch.int32, device=dev)        # cached positions of each slot
        self.lens_k = torch.ones(S, dtype=torch.int32, device=dev)       # lens + 1: what the step runner reads
        self.x_next = torch.zeros((S, 1, H), dtype=torch.float32, device=dev)
        self.trailing = torch.zeros((S, 1, H), dtype=torch.float32, device=dev)
        self.trailing_idx = torch.zeros(S, dtype=torch.int64, device=dev)
        self.pad = torch.zeros((1, H), dtype=torch.float32, device=dev)
        self.hist = torch.full((S, max_frames + 1), -1, dtype=torch.int32, device=dev)
        self.hist_len = torch.zeros(S, dtype=torch.int32, device=dev)
        self.finished = torch.ones(S, dtype=torch.int32, device=dev)     # a free slot is a finished row
        self.codes = torch.zeros((S, max_frames, G), dtype=torch.int32, device=dev)
        self.fidx = torch.zeros(S, dtype=torch.int64, device=dev)        # frames generated per slot
        self.cache = eng.talker.make_cache()
        self.occupied = torch.zeros(S, dtype=torch.int32, device=dev)    # 1 while the slot holds a request (free slots stay at position 0)
        self.hlens = [0] * S                                             # host copy of lens for the occupied slots (capacity / table bounds)
        self.limit = [0] * S                                             # frames a slot's request may still generate in total
        self.max_len = 0                                                 # max(hlens)
        self._cp_caches: Dict[int, list] = {}
        self._ar = torch.arange(S, device=dev)

    # ------------------------------------------------------------------ storage
    def _ensure_capacity(self, positions: int):
        """Every layer's slot buffer holds ``positions`` rows per slot (grown in steps of 256, contents kept)."""
        need = ops.round_up(max(positions, 1), 256)
        for c in self.cache:
            if c.kv is None:
                c.kv = torch.zeros((self.S, need, c.width), dtype=c.dtype, device=c.device)
            elif c.kv.shape[1] < need:
                new = torch.zeros((self.S, need, c.width), dtype=c.dtype, device=c.device)
                new[:, : c.kv.shape[1]] = c.kv
                c.kv = new

    def _cp_cache(self, B: int):
        if B not in self._cp_caches:
            self._cp_caches[B] = self.eng.cp.make_cache()
        return self._cp_caches[B]

    def _noise(self, B: int, V: int):
        if self.generator is None or self.kw["temperature"] <= 0:
            return None
        e = torch.empty((B, ops.round_up(V, 4)), dtype=torch.float32, device=self.eng.device).exponential_(generator=self.generator)
        return -torch.log(e)

    # ------------------------------------------------------------------ one frame for B rows (the body of Qwen3Talker.generate's loop)
    def _frame(self, last: torch.Tensor, hist, hist_len, finished, trailing, trailing_idx):
        """``last`` [B, 1, H] final-normed talker output.  Updates hist / hist_len / finished / trailing_idx in place for the rows still alive.
        Returns (codes of the frame [B, G] int32, next input embedding [B, 1, H], alive mask before this frame's EOS took effect)."""
        eng, cfg = self.eng, self.eng.cfg
        cp = cfg.code_predictor_config
        B, H, G = last.shape[0], cfg.hidden_size, cfg.num_code_groups
        V0, Vc = cfg.vocab_size, cp.vocab_size
        kw = self.kw
        ar = self._ar[:B]
        row = torch.zeros((B, G), dtype=torch.int32, device=eng.device)
        logits = eng._logits(last, eng.codec_head)
        ops.sample(logits, row[:, 0], V=V0, suppress_mask=eng.suppress_mask, history=hist, hist_len=hist_len, repetition_penalty=kw["repetition_penalty"],
                   temperature=kw["temperature"], top_k=kw["top_k"], top_p=kw["top_p"], gumbel=self._noise(B, V0), done=finished,
                   done_token=cfg.codec_eos_token_id)
        tok = row[:, 0]
        finished |= (tok == cfg.codec_eos_token_id).to(torch.int32)
        cp_cache = self._cp_cache(B)
        for c in cp_cache:
            c.reset()
        tall = B <= eng.cp.max_decode_rows
        for i in range(G - 1):
            if i == 0 and tall:
                x0 = last.clone()
                if eng.mtp is not None:
                    xp = eng._f(B, 1, cp.hidden_size)
                    linear(x0, eng.mtp, xp, precision=eng.precision)
                    x0 = xp
                eng.cp(x0, cp_cache)
                xin = eng._f(B, 1, H)
                ops.embed_sum(eng.codec_table, row[:, 0:1].unsqueeze(1), xin)
            elif i == 0:
                xin = eng._f(B, 2, H)
                xin[:, 0:1, :] = last
                ops.embed_sum(eng.codec_table, row[:, 0:1].unsqueeze(1), xin[:, 1:2, :])
            else:
                xin = eng._f(B, 1, H)
                ops.embed_sum(eng.codec_table, row[:, i:i + 1].unsqueeze(1), xin, slot_offset=eng.codec_offs[i:i + 1])
            if eng.mtp is not None:
                xp = eng._f(B, xin.shape[1], cp.hidden_size)
                linear(xin, eng.mtp, xp, precision=eng.precision)
                xin = xp
            if xin.shape[1] == 1 and tall and eng.cp.native_decode and (B > 8 or eng.cp.cfg.d_model <= 2048):
                hc = eng.cp(xin, cp_cache, defer_final_norm=True)
                lg = eng._logits(hc, eng.lm_heads[i], norm=eng.cp.final_norm_arg())
            else:
                hc = eng.cp(xin, cp_cache)
                lg = eng._logits(hc[:, -1:, :].contiguous(), eng.lm_heads[i])
            ops.sample(lg, row[:, i + 1], V=Vc, temperature=kw["temperature"], top_k=kw["top_k"], top_p=kw["top_p"], gumbel=self._noise(B, Vc))
        # next input: the request's next trailing-text position (tts_pad once it is exhausted; rows are right-padded with tts_pad, so the
        # per-request rule of continuous_batching.py:266-283 and the padded-batch rule of qwen3_tts.py:993-1015 coincide) + the 16 codec embeddings
        Tt = trailing.shape[1]
        clamped = torch.clamp(trailing_idx, max=Tt - 1)
        text = trailing[ar, clamped]
        text = torch.where((trailing_idx >= Tt)[:, None], self.pad.expand_as(text), text)
        nx = eng._f(B, 1, H)
        ops.embed_sum(eng.codec_table, row.unsqueeze(1), nx, slot_offset=eng.codec_offs, add=text[:, None, :].contiguous())
        alive = finished == 0
        trailing_idx += alive.to(torch.int64)
        hist[ar, hist_len.long()] = torch.where(alive, tok, hist[ar, hist_len.long()])
        hist_len += alive.to(torch.int32)
        return row, nx, alive

    # ------------------------------------------------------------------ admission (continuous_batching.py:109-189)
    def admit(self, slots: List[int], input_embeds: torch.Tensor, left_pad: List[int], trailing: torch.Tensor, tts_pad: torch.Tensor) -> List[bool]:
        """Prefills the newcomers as ONE left-padded batch, files every prompt into its slot, samples their first frame.  ``trailing`` [n, Tt, H]
        right-padded with ``tts_pad``.  Returns, per newcomer, whether its first token was already EOS."""
        eng, dev = self.eng, self.eng.device
        n, L = input_embeds.shape[0], input_embeds.shape[1]
        assert n == len(slots) == len(left_pad) and len(set(slots)) == n and all(0 <= s < self.S for s in slots)
        x = input_embeds.to(dev, torch.float32).contiguous().clone()
        trailing = trailing.to(dev, torch.float32)
        self.pad = tts_pad.to(dev, torch.float32).reshape(1, -1)
        plens = [L - int(p) for p in left_pad]
        rows = eng.talker.cos.shape[0]
        if max(plens) >= rows:
            raise ValueError(f"prompt of {max(plens)} positions does not fit the talker's {rows} rotary positions")
        self._ensure_capacity(max(max(plens), self.max_len) + 2)
        tmp = eng.talker.make_cache()
        ks = torch.tensor([int(p) for p in left_pad], dtype=torch.int32, device=dev) if n > 1 or left_pad[0] else None
        h = eng.talker(x, tmp, k_start=ks)
        last = h[:, -1:, :].contiguous()
        for c, t in zip(self.cache, tmp):   # each prompt's keys / values into its slot, ONCE (positions count from the first real token)
            for r, s in enumerate(slots):
                c.kv[s, : plens[r]] = t.kv[r, int(left_pad[r]):L]
        sl = torch.tensor(slots, dtype=torch.long, device=dev)
        hist = torch.full((n, self.max_frames + 1), -1, dtype=torch.int32, device=dev)
        hist_len = torch.zeros(n, dtype=torch.int32, device=dev)
        fin = torch.zeros(n, dtype=torch.int32, device=dev)
        tidx = torch.zeros(n, dtype=torch.int64, device=dev)
        row, nx, alive = self._frame(last, hist, hist_len, fin, trailing, tidx)
        Tt = trailing.shape[1]
        if Tt > self.trailing.shape[1]:   # widen the slot rows, padding with tts_pad (what the exhausted rule reads)
            wide = self.pad.expand(self.S, Tt, -1).clone()
            wide[:, : self.trailing.shape[1]] = self.trailing
            self.trailing = wide
        self.trailing[sl] = self.pad.expand(n, self.trailing.shape[1], -1).clone()
        self.trailing[sl, :Tt] = trailing
        self.lens[sl] = torch.tensor(plens, dtype=torch.int32, device=dev)
        self.x_next[sl] = nx
        self.hist[sl] = hist
        self.hist_len[sl] = hist_len
        self.finished[sl] = fin
        self.trailing_idx[sl] = tidx
        self.codes[sl, 0] = row
        self.fidx[sl] = alive.to(torch.int64)
        self.occupied[sl] = 1
        for r, sidx in enumerate(slots):
            self.hlens[sidx] = plens[r]
            self.limit[sidx] = min(self.max_frames, rows - plens[r])   # a request stops where the rotary tables end (the reference's are unbounded)
        self.max_len = max(self.hlens)
        return [bool(v) for v in fin.cpu().tolist()]

    # ------------------------------------------------------------------ one step of every occupied slot (continuous_batching.py:191-239)
    def advance(self, n_rows: int) -> List[bool]:
        """One talker step for slots [0, n_rows) (free slots among them ride along as finished rows).  Returns the finished flag of each row."""
        eng = self.eng
        n = n_rows
        torch.add(self.lens[:n], 1, out=self.lens_k[:n])
        self._ensure_capacity(self.max_len + 2)
        for c in self.cache:
            c.offset = self.max_len
        x = self.x_next[:n].clone()
        h = eng.talker.decode_step(x, self.cache, slot_lens_k=self.lens_k)
        row, nx, alive = self._frame(h, self.hist[:n], self.hist_len[:n], self.finished[:n], self.trailing[:n], self.trailing_idx[:n])
        ar = self._ar[:n]
        f = torch.clamp(self.fidx[:n], max=self.max_frames - 1)
        self.codes[ar, f] = torch.where(alive[:, None], row, self.codes[ar, f])
        self.fidx[:n] += alive.to(torch.int64)
        # every occupied row appended one position (a row that just emitted EOS too: its request leaves with this step)
        self.lens[:n] += self.occupied[:n]
        self.x_next[:n] = nx
        for sidx in range(n):
            if self.hlens[sidx] > 0:
                self.hlens[sidx] += 1
        self.max_len = max(self.hlens)
        return [bool(v) for v in self.finished[:n].cpu().tolist()]

    def frames(self, slot: int) -> int:
        return int(self.fidx[slot])

    def take_codes(self, slot: int, n_frames: int) -> torch.Tensor:
        """The ``n_frames`` generated code frames of a slot ([n, G] int64) -- the slot can be released afterwards."""
        return self.codes[slot, :n_frames].to(torch.int64).clone()

    def release(self, slot: int):
        self.finished[slot] = 1
        self.occupied[slot] = 0
        self.hlens[slot] = 0
        self.max_len = max(self.hlens)
        self.lens[slot] = 0
        self.fidx[slot] = 0
        self.hist_len[slot] = 0
        self.trailing_idx[slot] = 0
