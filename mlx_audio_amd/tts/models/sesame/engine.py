"""CSM-1B frame generation on MI355X: host schedule over the HIP kernels.

Mirrors ``SesameModel.generate_frame`` and the frame loop of ``Model.generate`` (``tts/models/sesame/sesame.py:361-425, 813-846``):
  * the masked sum of the 32 audio-codebook embeddings + the text embedding is ONE ``embed_sum`` launch over a stacked table (masked slots
    carry id -1), instead of a [B, S, 33, D] gather + multiply + reduce;
  * backbone (Llama-1B shapes) and depth decoder (Llama-100M shapes) are ``lm.stack.TransformerStack`` with the Llama-3 scaled, interleaved
    RoPE of sesame/attention.py:41-105 (tables built on the host in float32 exactly as ``rope_init`` does);
  * every decoder step is 1-2 rows per sequence: GEMV on row-major bf16 weights, KV-streaming attention; the 31 ``audio_head[i]`` matrices
    ([decoder_dim, vocab], applied as ``h @ W``) are stored transposed once at load so that they are GEMV row-major images;
  * sampling (temperature 0.9, top-k 50: make_sampler, sesame.py:767) is one kernel per token with caller-supplied Gumbel noise;
  * the all-zero-frame EOS test (sesame.py:828, a host sync per frame in the reference) is polled every ``poll`` frames.
BASELINE config[4] names fp8 MFMA GEMMs for this model; at <= 8 rows per step the GEMMs are weight-bandwidth bound GEMVs, so the gain of
fp8 is the halved weight stream, not MFMA rate -- the fp8 weight image is the next step, bf16 is what is measured here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .... import ops
from ....lm.stack import StackConfig, TransformerStack, linear, make_lin
from ....lm.synthetic import make_stack_weights


@dataclass
class CSMConfig:
    backbone: StackConfig
    decoder: StackConfig
    audio_vocab_size: int = 2051
    audio_num_codebooks: int = 32
    text_vocab_size: int = 128256


def llama_stack(hidden, layers, heads, kv, dh, ff) -> StackConfig:
    """create_llama_model_args (sesame.py:204-299): rms eps 1e-5, rope theta 5e5, Llama-3 scaling factor 32, 2048 positions."""
    return StackConfig(d_model=hidden, n_layers=layers, n_heads=heads, n_kv_heads=kv, head_dim=dh, d_ff=ff, norm="rms", norm_eps=1e-5,
                       rope_theta=500000.0, rope_interleaved=True, rope_llama3_factor=32.0, max_pos=2048, mlp="swiglu")


def csm_1b() -> CSMConfig:
    return CSMConfig(backbone=llama_stack(2048, 16, 32, 8, 64, 8192), decoder=llama_stack(1024, 4, 8, 2, 128, 8192))


def tiny_csm() -> CSMConfig:
    return CSMConfig(backbone=llama_stack(256, 2, 4, 2, 64, 512), decoder=llama_stack(128, 2, 2, 1, 64, 256), audio_vocab_size=67,
                     audio_num_codebooks=4, text_vocab_size=300)


def make_csm_weights(cfg: CSMConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(torch.bfloat16).to(torch.float32)

    w = {"backbone." + k: v for k, v in make_stack_weights(cfg.backbone, seed=seed * 5 + 1, gain=1.5).items()}
    w.update({"decoder." + k: v for k, v in make_stack_weights(cfg.decoder, seed=seed * 5 + 2, gain=1.5).items()})
    D, Dd = cfg.backbone.d_model, cfg.decoder.d_model
    w["text_embeddings.weight"] = rnd(cfg.text_vocab_size, D, std=0.5)
    w["audio_embeddings.weight"] = rnd(cfg.audio_vocab_size * cfg.audio_num_codebooks, D, std=0.5)
    w["projection.weight"] = rnd(Dd, D, std=1.0 / math.sqrt(D))
    w["codebook0_head.weight"] = rnd(cfg.audio_vocab_size, D, std=4.0 / math.sqrt(D))
    w["audio_head"] = rnd(cfg.audio_num_codebooks - 1, Dd, cfg.audio_vocab_size, std=4.0 / math.sqrt(Dd))
    return w


class CSMEngine:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg: CSMConfig, device="cuda:0", precision: int = 2):
        ops.require_gpu()
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        self.fuse_embed = True  # one sequence: the depth decoder's input embedding lookup runs inside the projection GEMV (False: embed_sum + GEMV)
        # Cache policy of the weight streams in one-sequence steps (mi355_gemv_args.w_policy): the 2.4 GB backbone and the 32 heads (each read once per
        # frame) stream past the caches; the 220 MB depth decoder runs 31 times per frame and keeps the default policy.  set_stream_policy() applies it.
        self.head_policy = 1
        dev = self.device
        w = {k: v.detach().to(torch.bfloat16).to(torch.float32).cpu() for k, v in weights.items()}
        self.backbone = TransformerStack(w, cfg.backbone, device=dev, precision=precision, prefix="backbone.")
        self.decoder = TransformerStack(w, cfg.decoder, device=dev, precision=precision, prefix="decoder.")
        self.backbone.w_policy, self.decoder.w_policy = 1, 0   # measured: 5.35 -> 5.10 ms per frame (profiles/r4_bench_csm_*_call6.json); nt on the decoder too: slower
        self.projection = make_lin(w["projection.weight"], None, dev)
        self.c0_head = make_lin(w["codebook0_head.weight"], None, dev)
        self.heads = [make_lin(w["audio_head"][i].t().contiguous(), None, dev) for i in range(cfg.audio_num_codebooks - 1)]
        # stacked embedding table: 32 audio codebooks (already offset by codebook * vocab in the checkpoint layout), then the text table
        self.table = torch.cat([w["audio_embeddings.weight"], w["text_embeddings.weight"]], 0).contiguous().to(dev)
        nb = cfg.audio_num_codebooks
        self.slot_offs = torch.tensor([i * cfg.audio_vocab_size for i in range(nb)] + [nb * cfg.audio_vocab_size], dtype=torch.int32, device=dev)
        self.backbone_cache = self.backbone.make_cache()
        self.decoder_cache = self.decoder.make_cache()

    def set_stream_policy(self, backbone: int = 1, heads: int = 1, decoder: int = 0):
        self.backbone.w_policy, self.decoder.w_policy, self.head_policy = int(backbone), int(decoder), int(heads)
        for st in (self.backbone, self.decoder):
            st._native = None   # descriptors are rebuilt with the new policy

    def reset_caches(self):
        for c in self.backbone_cache:
            c.reset()

    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _logits(self, h_last, head, norm=None):
        B = h_last.shape[0]
        V = head.rm.n
        out = self._f(B, 1, ops.round_up(V, 4))
        linear(h_last, head, out[:, :, :V], precision=self.precision, norm=norm, w_policy=self.head_policy)
        return out[:, 0, :]

    def generate_frame(self, tokens: torch.Tensor, tokens_mask: torch.Tensor, *, temperature: float = 0.9, top_k: int = 50,
                       gumbel: Optional[torch.Tensor] = None, forced: Optional[torch.Tensor] = None, trace: Optional[list] = None,
                       out: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """tokens int [B, S, n_cb + 1], tokens_mask bool [B, S, n_cb + 1] -> sample int32 [B, n_cb] (device)."""
        cfg = self.cfg
        dev = self.device
        nb, V = cfg.audio_num_codebooks, cfg.audio_vocab_size
        B, S, _ = tokens.shape
        ids = torch.where(tokens_mask.to(dev), tokens.to(dev, torch.int32), torch.full((), -1, dtype=torch.int32, device=dev)).contiguous()
        D = cfg.backbone.d_model
        h = self._f(B, S, D)
        ops.embed_sum(self.table, ids, h, slot_offset=self.slot_offs)
        h = self.backbone(h, self.backbone_cache)
        last = h[:, -1:, :].contiguous()
        sample = out if out is not None else torch.zeros((B, nb), dtype=torch.int32, device=dev)
        Vp = ops.round_up(V, 4)

        def noise(i):
            if gumbel is None:
                if generator is None or temperature <= 0:
                    return None  # greedy
                # make_sampler(temp, top_k) -> mx.random.categorical in the reference (sesame.py:767): Gumbel-max with device-drawn noise here
                return -torch.log(torch.empty((B, Vp), dtype=torch.float32, device=dev).exponential_(generator=generator))
            n = torch.zeros((B, Vp), dtype=torch.float32, device=dev)
            n[:, :V] = gumbel[i].to(dev, torch.float32)
            return n

        def draw(logits, i):
            if trace is not None:
                trace.append(logits[:, :V].clone())
            ops.sample(logits, sample[:, i], V=V, temperature=temperature, top_k=top_k, gumbel=noise(i))
            if forced is not None:   # teacher forcing; a negative entry keeps the step's own selection
                fo = forced[:, i].to(dev, torch.int32)
                sample[:, i] = torch.where(fo >= 0, fo, sample[:, i])

        draw(self._logits(last, self.c0_head), 0)
        cache = self.decoder_cache  # reset for every frame (sesame.py:385-387): offsets back to 0, buffers reused
        for c in cache:
            c.reset()
        Dd = cfg.decoder.d_model
        for i in range(1, nb):
            if i == 1 and B <= self.decoder.max_decode_rows:
                # [last_h, c0_embed] (sesame.py:380) as two single-position steps through the causal depth decoder: same result, and single
                # positions run on the GEMV / KV-streaming path; only the second position's output is used
                p0 = self._f(B, 1, Dd)
                linear(last, self.projection, p0, precision=self.precision)
                self.decoder(p0, cache)
                if B == 1 and self.fuse_embed:
                    cur = None
                else:
                    cur = self._f(B, 1, D)
                    ops.embed_sum(self.table, sample[:, 0:1].unsqueeze(1), cur, slot_offset=self.slot_offs[0:1])
            elif i == 1:
                cur = self._f(B, 2, D)
                cur[:, 0:1, :] = last
                ops.embed_sum(self.table, sample[:, 0:1].unsqueeze(1), cur[:, 1:2, :], slot_offset=self.slot_offs[0:1])
            elif B == 1 and self.fuse_embed:
                cur = None
            else:
                cur = self._f(B, 1, D)
                ops.embed_sum(self.table, sample[:, i - 1:i].unsqueeze(1), cur, slot_offset=self.slot_offs[i - 1:i])
            p = self._f(B, cur.shape[1] if cur is not None else 1, Dd)
            if cur is None:  # one sequence: embedding lookup of the code just sampled fused into the projection GEMV (mi355_gemv_args.x_ids)
                ops.gemv(self.table, self.projection.rm, p[:, 0, :], x_ids=sample[0, i - 1:i], x_id_offset=(i - 1) * V)   # (4 MB, re-read 31 x per frame: default policy)
            else:
                linear(cur, self.projection, p, precision=self.precision)
            if p.shape[1] == 1 and B <= self.decoder.max_decode_rows and self.decoder.native_decode:
                # the depth decoder's final RMSNorm runs inside the head GEMV (its fused-norm prologue): one launch less per codebook
                dh = self.decoder(p, cache, defer_final_norm=True)
                draw(self._logits(dh, self.heads[i - 1], norm=self.decoder.final_norm_arg()), i)
            else:
                dh = self.decoder(p, cache)
                draw(self._logits(dh[:, -1:, :].contiguous(), self.heads[i - 1]), i)
        return sample

    def generate_chunks(self, prompt_tokens: torch.Tensor, prompt_mask: torch.Tensor, max_frames: int, *, chunk: int, temperature: float = 0.9,
                        top_k: int = 50, gumbel: Optional[torch.Tensor] = None, forced: Optional[torch.Tensor] = None,
                        generator: Optional[torch.Generator] = None):
        """The frame loop as a generator (sesame.py:813-860 with ``stream=True``): yields int64 [B, n, n_cb] blocks of the frames generated since the last
        yield, every ``chunk`` frames, WHILE the loop runs -- the caller decodes and hands on a block before the next frame is computed.  The
        all-zero EOS frame ends the loop and is not yielded; one host read-back per block (the reference reads back every frame, sesame.py:828).
        ``self.frames_generated`` counts the frames computed so far (what a test reads to see that a block left before the loop ended)."""
        cfg, dev = self.cfg, self.device
        self.reset_caches()
        max_pos = self.backbone.cos.shape[0]
        if prompt_tokens.shape[1] + max_frames > max_pos:  # sesame.py:817-820
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_pos - max_frames}")
        nb = cfg.audio_num_codebooks
        B = prompt_tokens.shape[0]
        frames = torch.zeros((B, max_frames, nb), dtype=torch.int32, device=dev)
        toks, mask = prompt_tokens, prompt_mask
        next_mask = torch.cat([torch.ones(B, 1, nb, dtype=torch.bool), torch.zeros(B, 1, 1, dtype=torch.bool)], dim=2).to(dev)
        sent = 0
        self.frames_generated = 0
        for f in range(max_frames):
            s = self.generate_frame(toks, mask, temperature=temperature, top_k=top_k, gumbel=None if gumbel is None else gumbel[f],
                                    forced=None if forced is None else forced[:, f], out=frames[:, f, :], generator=generator)
            self.frames_generated = f + 1
            toks = torch.cat([s, torch.zeros((B, 1), dtype=torch.int32, device=dev)], dim=1)[:, None, :]
            mask = next_mask
            if f + 1 - sent >= chunk or f + 1 == max_frames:
                blk = frames[:, sent:f + 1].to(torch.int64)
                zero = (blk == 0).all(dim=2).all(dim=0).cpu()
                if bool(zero.any()):
                    first = int(torch.nonzero(zero)[0])
                    if first > 0:
                        yield blk[:, :first]
                    return
                yield blk
                sent = f + 1

    def generate(self, prompt_tokens: torch.Tensor, prompt_mask: torch.Tensor, max_frames: int, *, temperature: float = 0.9, top_k: int = 50,
                 gumbel: Optional[torch.Tensor] = None, forced: Optional[torch.Tensor] = None, record: bool = False, poll: int = 16,
                 generator: Optional[torch.Generator] = None):
        """Frame loop (sesame.py:813-846): frames int64 [B, n, n_cb]; stops at the first all-zero frame (EOS)."""
        cfg = self.cfg
        dev = self.device
        self.reset_caches()
        max_pos = self.backbone.cos.shape[0]
        if prompt_tokens.shape[1] + max_frames > max_pos:  # sesame.py:817-820
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_pos - max_frames}")
        nb = cfg.audio_num_codebooks
        B = prompt_tokens.shape[0]
        frames = torch.zeros((B, max_frames, nb), dtype=torch.int32, device=dev)
        toks, mask = prompt_tokens, prompt_mask
        next_mask = torch.cat([torch.ones(B, 1, nb, dtype=torch.bool), torch.zeros(B, 1, 1, dtype=torch.bool)], dim=2).to(dev)
        traces: List[list] = []
        n = 0
        for f in range(max_frames):
            tr = [] if record else None
            s = self.generate_frame(toks, mask, temperature=temperature, top_k=top_k, gumbel=None if gumbel is None else gumbel[f],
                                    forced=None if forced is None else forced[:, f], trace=tr, out=frames[:, f, :], generator=generator)
            traces.append(tr)
            toks = torch.cat([s, torch.zeros((B, 1), dtype=torch.int32, device=dev)], dim=1)[:, None, :]
            mask = next_mask
            n = f + 1
            if forced is None and n % poll == 0 and bool((frames[:, :n] == 0).all(dim=(0, 2)).any()):  # the only host round trip
                break
        fr = frames[:, :n].to(torch.int64)
        zero = (fr == 0).all(dim=2).all(dim=0).cpu()
        if bool(zero.any()):
            first = int(torch.nonzero(zero)[0])
            fr, traces = fr[:, :first], traces[:first]
        return dict(frames=fr, trace=traces)
