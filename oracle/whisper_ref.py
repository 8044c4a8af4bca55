"""PyTorch-CPU / numpy restatement of the Whisper hot path (TEST ORACLE, not product).

Follows, block by block (reference = /root/reference/mlx_audio, Blaizzy/mlx-audio v0.5.0):
  * ``stt/models/whisper/whisper.py:329-335``  sinusoids
  * ``stt/models/whisper/whisper.py:338-385``  MultiHeadAttention (q,k scaled by d_h^-0.25 each, K without bias,
                                               additive causal mask, softmax(precise=True))
  * ``stt/models/whisper/whisper.py:388-416``  ResidualAttentionBlock (pre-LN, GELU(erf) MLP 4x)
  * ``stt/models/whisper/whisper.py:419-448``  AudioEncoder (conv k3 p1 + GELU, conv k3 s2 p1 + GELU, + sinusoids)
  * ``stt/models/whisper/whisper.py:451-498``  TextDecoder (token + learned positional embedding, tied logits)
  * ``stt/models/whisper/decoding.py:302-330`` GreedyDecoder.update / finalize
  * ``stt/models/whisper/decoding.py:333-443`` SuppressBlank, SuppressTokens, ApplyTimestampRules
  * ``stt/models/whisper/decoding.py:588-632`` DecodingTask._main_loop
  * ``stt/models/whisper/audio.py:41-82``      log_mel_spectrogram (via oracle.dsp_ref.whisper_log_mel)

Precision model: parameters hold fp16-representable values (the checkpoint dtype of Whisper in the reference,
``Model(dims, dtype=mx.float16)``), arithmetic is float32 (``dtype=torch.float64`` gives a higher-precision truth).
The reference itself rounds every activation to fp16; that rounding noise (2^-11 relative per op) is the reason the
logit tolerance of the parity tests is 2e-3 and token equality is asserted only where the oracle's top-2 logit
margin exceeds the tolerance.

Parity status: **pinned to the reference's own modules** (round 2): tests/golden/make_reference_fixtures.py runs the reference's source files for
Whisper (imported from /root/reference, unmodified, over the numpy stand-in for MLX in tests/golden/mlx_shim.py) on a seeded tiny checkpoint, and
tests/test_reference_fixtures_cpu.py holds this oracle to the result -- ``Model`` (encoder, decoder + KV cache) and ``DecodingTask`` (greedy, all three logit filters, no-speech probability) in float32: features / logits 2e-5, decoded tokens of both decoding modes exact.
The reference's own tests pin shapes / token-rule cases only (reproduced in tests/test_oracle_golden.py); MLX's kernels are not exercised by the stand-in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class Dims:
    """ModelDimensions (whisper.py:280-291)."""
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 768
    n_audio_head: int = 12
    n_audio_layer: int = 12
    n_vocab: int = 51865
    n_text_ctx: int = 448
    n_text_state: int = 768
    n_text_head: int = 12
    n_text_layer: int = 12


@dataclass
class TokenizerSpec:
    """The special-token ids of the multilingual Whisper vocabulary that the decode loop needs (tokenizer.py)."""
    eot: int = 50257
    sot: int = 50258
    translate: int = 50358
    transcribe: int = 50359
    sot_lm: int = 50360
    sot_prev: int = 50361
    no_speech: int = 50362
    no_timestamps: Optional[int] = 50363
    timestamp_begin: int = 50364
    language_token: int = 50259  # <|en|>
    blank_ids: Tuple[int, ...] = (220,)  # tokenizer.encode(" ")
    non_speech_tokens: Tuple[int, ...] = ()

    @property
    def sot_sequence(self) -> Tuple[int, ...]:
        return (self.sot, self.language_token, self.transcribe)

    @property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return self.sot_sequence + (self.no_timestamps,)


def sinusoids(length: int, channels: int, max_timescale: float = 10000) -> Tensor:
    """whisper.py:329-335 (float32 like mx.arange / mx.exp)."""
    assert channels % 2 == 0
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    st = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def gelu(x: Tensor) -> Tensor:
    return F.gelu(x)  # nn.gelu = exact erf form


class WhisperRef:
    def __init__(self, weights: Dict[str, Tensor], dims: Dims, dtype=torch.float32, param_dtype=torch.float16, cross_kv_dtype=torch.float16):
        """``cross_kv_dtype``: the cross-attention K / V are computed once per window and CACHED in the model dtype (whisper.py:360-365: fp16 for
        the released checkpoints); the restatement keeps float32 activations everywhere else, so that one rounding is applied explicitly where
        the reference stores them (None = keep float32)."""
        self.dims = dims
        self.dtype = dtype
        self.cross_kv_dtype = cross_kv_dtype
        # the encoder's self-attention reads its K / V operands in the same 16-bit type (in the reference every activation is fp16: this
        # restatement keeps float32 everywhere except the K / V operands of attention, which are the values the reference would hold in fp16 too)
        self.enc_kv_dtype = cross_kv_dtype
        self.w = {k: v.to(param_dtype).to(dtype) for k, v in weights.items()}
        # whisper.py:434: sinusoids(...).astype(dtype) -- the positional table is rounded to the model dtype
        self.enc_pos = sinusoids(dims.n_audio_ctx, dims.n_audio_state).to(param_dtype).to(dtype)

    # ------------------------------------------------------------------ blocks
    def _lin(self, x: Tensor, name: str, bias: bool = True) -> Tensor:
        return F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"] if bias else None)

    def _ln(self, x: Tensor, name: str) -> Tensor:
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], 1e-5)

    def _mha(self, pfx: str, n_head: int, x: Tensor, xa: Optional[Tensor], mask: Optional[Tensor], kv_cache):
        """whisper.py:347-385."""
        q = self._lin(x, pfx + ".query")
        if xa is None:
            k = self._lin(x, pfx + ".key", bias=False)
            v = self._lin(x, pfx + ".value")
            if pfx.startswith("encoder.") and self.enc_kv_dtype is not None:
                k, v = k.to(self.enc_kv_dtype).to(self.dtype), v.to(self.enc_kv_dtype).to(self.dtype)
            if kv_cache is not None:
                k = torch.cat([kv_cache[0], k], dim=1)
                v = torch.cat([kv_cache[1], v], dim=1)
        elif kv_cache is None:
            k = self._lin(xa, pfx + ".key", bias=False)
            v = self._lin(xa, pfx + ".value")
            if self.cross_kv_dtype is not None:
                k, v = k.to(self.cross_kv_dtype).to(self.dtype), v.to(self.cross_kv_dtype).to(self.dtype)
        else:
            k, v = kv_cache
        B, n_ctx, n_state = q.shape
        scale = (n_state // n_head) ** -0.25
        qh = q.reshape(B, n_ctx, n_head, -1).permute(0, 2, 1, 3) * scale
        kh = k.reshape(B, k.shape[1], n_head, -1).permute(0, 2, 3, 1) * scale
        vh = v.reshape(B, v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        qk = qh @ kh
        if mask is not None:
            qk = qk + mask[:n_ctx, :n_ctx]
        w = torch.softmax(qk, dim=-1)
        out = (w @ vh).permute(0, 2, 1, 3).reshape(B, n_ctx, n_state)
        return self._lin(out, pfx + ".out"), (k, v)

    def _block(self, pfx: str, n_head: int, x: Tensor, xa: Optional[Tensor], mask: Optional[Tensor], kv_cache, cross: bool):
        """whisper.py:405-416."""
        kv, cross_kv = kv_cache if kv_cache else (None, None)
        y, kv = self._mha(pfx + ".attn", n_head, self._ln(x, pfx + ".attn_ln"), None, mask, kv)
        x = x + y
        if cross:
            y, cross_kv = self._mha(pfx + ".cross_attn", n_head, self._ln(x, pfx + ".cross_attn_ln"), xa, None, cross_kv)
            x = x + y
        x = x + self._lin(gelu(self._lin(self._ln(x, pfx + ".mlp_ln"), pfx + ".mlp1")), pfx + ".mlp2")
        return x, (kv, cross_kv)

    def _conv(self, x: Tensor, name: str, stride: int) -> Tensor:
        """nn.Conv1d on NLC input with MLX weights (C_out, K, C_in), padding 1."""
        w = self.w[name + ".weight"].permute(0, 2, 1)
        return F.conv1d(x.transpose(1, 2), w, self.w[name + ".bias"], stride=stride, padding=1).transpose(1, 2)

    def encoder(self, mel: Tensor, return_layers: bool = False):
        """whisper.py:438-448.  mel [B, 3000, n_mels] -> [B, n_audio_ctx, n_audio_state]."""
        d = self.dims
        x = mel.to(self.dtype)
        x = gelu(self._conv(x, "encoder.conv1", 1))
        x = gelu(self._conv(x, "encoder.conv2", 2))
        assert tuple(x.shape[1:]) == tuple(self.enc_pos.shape), "incorrect audio shape"
        x = x + self.enc_pos
        layers = [x]
        for i in range(d.n_audio_layer):
            x, _ = self._block(f"encoder.blocks.{i}", d.n_audio_head, x, None, None, None, cross=False)
            layers.append(x)
        x = self._ln(x, "encoder.ln_post")
        return (x, layers) if return_layers else x

    def decoder(self, tokens: Tensor, xa: Tensor, kv_cache=None):
        """whisper.py:476-498.  tokens int [B, n] -> logits [B, n, n_vocab], kv_cache."""
        d = self.dims
        offset = kv_cache[0][0][0].shape[1] if kv_cache else 0
        emb = self.w["decoder.token_embedding.weight"]
        x = emb[tokens] + self.w["decoder.positional_embedding"][offset:offset + tokens.shape[-1]]
        n = d.n_text_ctx
        # nn.MultiHeadAttention.create_additive_causal_mask: large negative above the diagonal
        mask = torch.triu(torch.full((n, n), -1e9, dtype=self.dtype), diagonal=1)
        if kv_cache is None:
            kv_cache = [None] * d.n_text_layer
        for i in range(d.n_text_layer):
            # NOTE whisper.py:373-379: the mask slice is [:n_ctx, :n_ctx] with n_ctx = query length; with a kv cache
            # (n_ctx = 1) that is mask[0:1, 0:1] = 0 broadcast over all keys, i.e. no masking -- correct for one token
            x, kv_cache[i] = self._block(f"decoder.blocks.{i}", d.n_text_head, x, xa, mask, kv_cache[i], cross=True)
        x = self._ln(x, "decoder.ln")
        return x @ emb.T, kv_cache

    # ------------------------------------------------------------------ decoding
    def decode(self, mel: Tensor, tok: TokenizerSpec, *, sample_len: Optional[int] = None, without_timestamps: bool = False,
               suppress_blank: bool = True, suppress_tokens: Optional[Sequence[int]] = None,
               max_initial_timestamp: Optional[float] = 1.0, forced_tokens: Optional[Tensor] = None, audio_features=None,
               record: bool = False, initial_tokens: Optional[Sequence[int]] = None, temperature: float = 0.0, gumbel=None):
        """DecodingTask.run / _main_loop (decoding.py:588-632, 634-700).

        ``initial_tokens``: the initial sequence when a prompt / prefix is given (decoding.py:525-551; default = the sot sequence).
        ``temperature`` > 0 with ``gumbel`` (callable step -> [B, V] Gumbel(0, 1) noise): ``categorical(logits / T)`` (decoding.py:266-269)
        drawn as ``argmax(logits / T + g)``, so that a test can hand both implementations the same noise.

        ``forced_tokens`` [B, steps] (optional): teacher forcing -- the tokens appended at each step are taken from
        here instead of the arg-max (the filters and log-probs are still evaluated), which lets a test compare the
        per-step filtered logits of two implementations on identical contexts.
        Returns dict(tokens [B, n], sum_logprobs [B], no_speech_probs [B], trace=[per-step dict]).
        """
        d = self.dims
        xa = self.encoder(mel) if audio_features is None else audio_features.to(self.dtype)
        B = xa.shape[0]
        sot_sequence = tok.sot_sequence_including_notimestamps if without_timestamps else tok.sot_sequence
        initial = tuple(sot_sequence) if initial_tokens is None else tuple(int(t) for t in initial_tokens)
        sample_begin = len(initial)
        sot_index = initial.index(tok.sot)
        sample_len = sample_len or d.n_text_ctx // 2
        filters = []
        if suppress_blank:
            filters.append(SuppressBlankRef(tok, sample_begin, d.n_vocab))
        if suppress_tokens is not None:
            filters.append(SuppressTokensRef(suppress_tokens, d.n_vocab))
        if not without_timestamps:
            precision = 30.0 / d.n_audio_ctx
            idx = round(max_initial_timestamp / precision) if max_initial_timestamp else None
            filters.append(ApplyTimestampRulesRef(tok, sample_begin, idx))
        dec = GreedyDecoderRef(tok.eot)
        tokens = torch.tensor([list(initial)] * B, dtype=torch.long)
        sum_logprobs = torch.zeros(B, dtype=torch.float32)
        kv = None
        trace: List[dict] = []
        no_speech = None
        for i in range(sample_len):
            inputs = tokens if i == 0 else tokens[:, -1:]
            if tokens.shape[-1] > d.n_text_ctx:
                break
            pre_logits, kv = self.decoder(inputs, xa, kv)
            pre_logits = pre_logits.to(torch.float32)
            if i == 0:
                probs = torch.softmax(pre_logits[:, sot_index], dim=-1)
                no_speech = probs[:, tok.no_speech]
            logits = pre_logits[:, -1]
            raw = logits.clone()
            for f in filters:
                logits = f.apply(logits, tokens)
            forced = None if forced_tokens is None else forced_tokens[:, i]
            if forced is None and temperature > 0:
                forced = (logits / temperature + gumbel(i).to(logits.dtype)).argmax(dim=-1)
            tokens, completed, sum_logprobs = dec.update(tokens, logits, sum_logprobs, forced)
            if record:
                trace.append(dict(raw=raw, filtered=logits.clone()))
            if completed and forced_tokens is None:
                break
        return dict(tokens=tokens, sum_logprobs=sum_logprobs, no_speech_probs=no_speech, trace=trace,
                    sample_begin=sample_begin, audio_features=xa)


class GreedyDecoderRef:
    """decoding.py:302-330 at temperature 0."""

    def __init__(self, eot: int):
        self.eot = eot

    def update(self, tokens: Tensor, logits: Tensor, sum_logprobs: Tensor, forced: Optional[Tensor] = None):
        next_tokens = logits.argmax(dim=-1) if forced is None else forced.to(torch.long)
        logprobs = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        current = logprobs[torch.arange(logprobs.shape[0]), next_tokens]
        alive = (tokens[:, -1] != self.eot)
        sum_logprobs = sum_logprobs + torch.where(alive, current, torch.zeros_like(current))
        eot_mask = tokens[:, -1] == self.eot
        next_tokens = torch.where(eot_mask, torch.full_like(next_tokens, self.eot), next_tokens)
        tokens = torch.cat([tokens, next_tokens[:, None]], dim=-1)
        completed = bool(torch.all(tokens[:, -1] == self.eot))
        return tokens, completed, sum_logprobs

    def finalize(self, tokens: Tensor, sum_logprobs: Tensor):
        return F.pad(tokens, (0, 1), value=self.eot), sum_logprobs


class SuppressBlankRef:
    """decoding.py:349-359."""

    def __init__(self, tok: TokenizerSpec, sample_begin: int, n_vocab: int):
        self.sample_begin = sample_begin
        mask = np.zeros(n_vocab, np.float32)
        mask[list(tok.blank_ids) + [tok.eot]] = -np.inf
        self.mask = torch.from_numpy(mask)

    def apply(self, logits: Tensor, tokens: Tensor) -> Tensor:
        if tokens.shape[1] == self.sample_begin:
            return logits + self.mask
        return logits


class SuppressTokensRef:
    """decoding.py:362-369."""

    def __init__(self, suppress_tokens: Sequence[int], n_vocab: int):
        mask = np.zeros(n_vocab, np.float32)
        mask[list(suppress_tokens)] = -np.inf
        self.mask = torch.from_numpy(mask)

    def apply(self, logits: Tensor, tokens: Tensor) -> Tensor:
        return logits + self.mask


class ApplyTimestampRulesRef:
    """decoding.py:372-443, transcribed statement by statement (including the index-vs-value slip at :410-419)."""

    def __init__(self, tok: TokenizerSpec, sample_begin: int, max_initial_timestamp_index: Optional[int]):
        self.tokenizer = tok
        self.sample_begin = sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index

    def apply(self, logits: Tensor, tokens: Tensor) -> Tensor:
        tk = self.tokenizer
        mask = np.zeros(tuple(logits.shape), np.float32)
        if tk.no_timestamps is not None:
            mask[:, tk.no_timestamps] = -np.inf
        toks = tokens.tolist()
        for k in range(len(toks)):
            seq = toks[k][self.sample_begin:]
            last_was_timestamp = len(seq) >= 1 and seq[-1] >= tk.timestamp_begin
            penultimate_was_timestamp = len(seq) < 2 or seq[-2] >= tk.timestamp_begin
            if last_was_timestamp:
                if penultimate_was_timestamp:
                    mask[k, tk.timestamp_begin:] = -np.inf
                else:
                    mask[k, : tk.eot] = -np.inf
            timestamps = [i for i, v in enumerate(seq) if v > tk.timestamp_begin]
            if len(timestamps) > 0:
                last_timestamp = timestamps[-1]
                if not last_timestamp or penultimate_was_timestamp:
                    last_timestamp += 1
                mask[k, tk.timestamp_begin: last_timestamp] = -np.inf
        if len(toks[0]) == self.sample_begin:
            mask[:, : tk.timestamp_begin] = -np.inf
            if self.max_initial_timestamp_index is not None:
                last_allowed = tk.timestamp_begin + self.max_initial_timestamp_index
                mask[:, last_allowed + 1:] = -np.inf
        mask_t = torch.from_numpy(mask)
        logprobs = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        timestamp_logprob = torch.logsumexp(logprobs[:, tk.timestamp_begin:], dim=-1, keepdim=True)
        max_text_token_logprob = logprobs[:, : tk.timestamp_begin].max(dim=-1, keepdim=True).values
        kill = timestamp_logprob > max_text_token_logprob
        mask_t[:, : tk.timestamp_begin] = torch.where(kill, torch.full_like(mask_t[:, : tk.timestamp_begin], -np.inf),
                                                      mask_t[:, : tk.timestamp_begin])
        return logits + mask_t
