"""Qwen3-TTS behind the reference's model protocol (``Model(config)``, ``sanitize``, ``post_load_hook``, ``generate``, ``batch_generate`` --
``tts/models/qwen3_tts/qwen3_tts.py:168-2937``), computing on MI355X through ``Qwen3Talker`` (talker + code predictor frame loop) and
``Qwen3TTSSpeechTokenizer`` (codec decoder).

Same as the reference: constructor / config records, checkpoint key handling, the chat-template prompt construction
(``_prepare_generation_inputs`` :326-484) and its left-padded batch form (``_prepare_batch_inputs`` :486-604), the sampling defaults, the
generator protocol and every field of ``GenerationResult`` / ``BatchGenerationResult``, chunked codec decode (15-frame chunks + 5 frames of
left context, :1050-1083).

Voice cloning (round 3): ``ref_audio`` alone adds the clip's x-vector to the codec prefix (:383-384); ``ref_audio`` + ``ref_text`` is the in-context
path (``_prepare_icl_generation_inputs`` :606-803, ``_generate_icl`` :2200-2510, the shared-reference batch :1724-2045): the clip's codes from the
speech tokenizer's ENCODER, its transcript and the x-vector of the ECAPA speaker encoder (``speaker_encoder.py``) in the prefill, the reference codes
in front of the generated ones at decode time, the reference's share of the waveform cut off proportionally.
Not in this build: streaming chunk decode with carried codec state (``stream=True`` decodes each chunk behind left context instead, which is also
what the reference's ``batch_generate(stream=True)`` does).
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Generator, List, Optional, Tuple, Union

import torch

from ..base import BatchGenerationResult, GenerationResult
from .config import ModelConfig, Qwen3TTSTokenizerConfig, filter_dict_for_dataclass
from .speech_tokenizer import Qwen3TTSSpeechTokenizer, check_array_shape_qwen3


def format_duration(seconds: float) -> str:
    """HH:MM:SS.mmm (qwen3_tts.py:160-165)."""
    return f"{int(seconds // 3600):02d}:{int((seconds % 3600) // 60):02d}:{seconds % 60:06.3f}"


@dataclass
class Qwen3BatchInputs:
    """``qwen3_tts.py:36-44``."""
    input_embeds: torch.Tensor
    trailing_text_hidden: torch.Tensor
    tts_pad_embed: torch.Tensor
    attention_mask: torch.Tensor
    left_padding: List[int]
    prefill_lens: List[int]
    trailing_lens: List[int]
    ref_codes: Optional[torch.Tensor] = None


class Model:
    def __init__(self, config: ModelConfig, device: str = "cuda", precision: int = 2):
        self.config = config
        self._sample_rate = config.sample_rate
        self.device = device
        self.precision = precision
        self.talker = None            # Qwen3Talker (engine), built by load_weights
        self.speaker_encoder = None   # Qwen3TTSSpeakerEncoder (engine), built by load_weights when a Base checkpoint carries speaker_encoder.*
        self._icl_cache: Dict = {}
        self.speech_tokenizer: Optional[Qwen3TTSSpeechTokenizer] = None
        self.tokenizer = None
        self.generate_config = None
        tc = config.talker_config
        self.supported_speakers = list(tc.spk_id.keys()) if tc.spk_id else []
        self.supported_languages = ["auto"] + [k for k in (tc.codec_language_id or {}) if "dialect" not in k]
        self.model_path = None

    # ------------------------------------------------------------------ protocol
    @property
    def sample_rate(self) -> int:
        return self._sample_rate

    @property
    def model_type(self) -> str:
        return "qwen3_tts"

    def eval(self):
        return self

    def supports_tts_batch(self, *, stream: bool = False, voice: Optional[str] = None, instruct: Optional[str] = None, ref_audio=None,
                           ref_text: Optional[str] = None, speed: Optional[float] = 1.0, pitch: Optional[float] = 1.0, **kwargs) -> bool:
        """``qwen3_tts.py:215-252``."""
        del kwargs
        if stream or speed not in (None, 1.0) or pitch not in (None, 1.0):
            return False
        kind = getattr(self.config, "tts_model_type", "base")
        if ref_audio is not None or ref_text is not None:
            return (kind == "base" and ref_audio is not None and ref_text is not None and voice is None and instruct is None
                    and self.speech_tokenizer is not None and self.speech_tokenizer.has_encoder)
        if kind not in {"base", "custom_voice"}:
            return False
        if kind == "base" and instruct:
            return False
        if kind == "custom_voice" and not voice:
            return False
        return True

    def supports_tts_continuous_batch(self, **kwargs) -> bool:
        """``qwen3_tts.py:254-257``."""
        if kwargs.get("ref_audio") is not None or kwargs.get("ref_text") is not None:
            return False
        return self.supports_tts_batch(**kwargs)

    def create_tts_batch_session(self, options, **kw):
        """``qwen3_tts.py:1114-1120``: the step-wise session the serving shell drives (``add`` / ``cancel`` / ``step``), here on slot KV caches."""
        from .continuous_batching import Qwen3TTSBatchSession

        return Qwen3TTSBatchSession(self, options, **kw)

    def load_speech_tokenizer(self, speech_tokenizer: Qwen3TTSSpeechTokenizer):
        self.speech_tokenizer = speech_tokenizer

    def load_generate_config(self, generate_config: dict):
        self.generate_config = generate_config

    def get_supported_speakers(self) -> List[str]:
        return self.supported_speakers

    def get_supported_languages(self) -> List[str]:
        return self.supported_languages

    def model_quant_predicate(self, path: str, module) -> bool:
        return not any(p in path for p in ("codec_embedding", "text_embedding", "speech_tokenizer", "speaker_encoder"))

    # ------------------------------------------------------------------ checkpoint handling
    @staticmethod
    def sanitize(weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """``qwen3_tts.py:2914-2937``: drop ``position_ids``, PyTorch conv layouts (out, in, K) -> (out, K, in) for every 3-D "conv" weight
        (and ``speaker_encoder.fc``) unless the shape heuristic says it already is."""
        out = {}
        for k, v in weights.items():
            if "position_ids" in k:
                continue
            if ("conv" in k or "speaker_encoder.fc" in k) and "weight" in k and v.dim() == 3:
                v = v if check_array_shape_qwen3(v) else v.permute(0, 2, 1).contiguous()
            out[k] = v
        return out

    def load_weights(self, weights, strict: bool = True):
        """``weights``: sanitized dict / pair list with the reference's module paths (``talker.model.layers...``, ``talker.codec_head...``,
        ``talker.code_predictor...``, ``speaker_encoder...``); builds the device engine of the talker."""
        from .talker import Qwen3Talker

        w = dict(weights)
        tw = {k[len("talker."):]: v for k, v in w.items() if k.startswith("talker.")}
        if not tw:
            raise ValueError("Qwen3-TTS checkpoint has no talker.* parameters")
        try:
            self.talker = Qwen3Talker(tw, self.config.talker_config, device=self.device, precision=self.precision)
        except KeyError as e:
            raise ValueError(f"Qwen3-TTS checkpoint is missing parameter talker.{e.args[0]}") from e
        # the reference builds the speaker encoder for every Base model (qwen3_tts.py:179-182); here it exists when the checkpoint carries its parameters
        sw = {k[len("speaker_encoder."):]: v for k, v in w.items() if k.startswith("speaker_encoder.")}
        if sw and getattr(self.config, "tts_model_type", "base") == "base":
            from .speaker_encoder import Qwen3TTSSpeakerEncoder

            self.speaker_encoder = Qwen3TTSSpeakerEncoder(self.config.speaker_encoder_config, device=self.device, precision=self.precision)
            self.speaker_encoder.load_weights(sw, strict=strict)
        return self

    @classmethod
    def post_load_hook(cls, model: "Model", model_path) -> "Model":
        """``qwen3_tts.py:2818-2911``: text tokenizer (``AutoTokenizer`` on the model directory), speech tokenizer from ``speech_tokenizer/``
        (its own config.json + safetensors), generation_config.json."""
        model_path = Path(model_path)
        if model.tokenizer is None:
            try:
                from transformers import AutoTokenizer

                model.tokenizer = AutoTokenizer.from_pretrained(str(model_path))
            except Exception as e:  # same behaviour as the reference: warn, fail later in generate()
                print(f"Warning: Could not load tokenizer: {e}")
        st_path = model_path / "speech_tokenizer"
        if st_path.exists():
            from safetensors.torch import load_file

            with open(st_path / "config.json") as f:
                d = json.load(f)
            tcfg = Qwen3TTSTokenizerConfig(**filter_dict_for_dataclass(Qwen3TTSTokenizerConfig, d))
            st = Qwen3TTSSpeechTokenizer(tcfg, device=model.device, precision=model.precision)
            tw: Dict[str, torch.Tensor] = {}
            for wf in sorted(st_path.glob("*.safetensors")):
                tw.update(load_file(str(wf)))
            if tw:
                st.load_weights(Qwen3TTSSpeechTokenizer.sanitize(tw))
                model.load_speech_tokenizer(st)
        gen = model_path / "generation_config.json"
        if gen.exists():
            with open(gen) as f:
                model.load_generate_config(json.load(f))
        return model

    # ------------------------------------------------------------------ prompt construction
    def _text_embed(self, ids: List[int]) -> torch.Tensor:
        return self.talker.embed_text(torch.tensor([ids], dtype=torch.int32))

    def _codec_embed(self, ids: List[int]) -> torch.Tensor:
        idx = torch.tensor(ids, dtype=torch.long, device=self.talker.device)
        return self.talker.codec_table[idx][None]  # slot 0 of the stacked table = the talker's codec_embedding

    def _codes_embed(self, codes: torch.Tensor) -> torch.Tensor:
        """Whole frames ``[1, T, num_code_groups]`` -> the sum of their codec embeddings ``[1, T, H]`` (qwen3_tts.py:701-709)."""
        return self.talker.embed_codes(codes)

    def extract_speaker_embedding(self, audio, sr: int = 24000) -> torch.Tensor:
        """``qwen3_tts.py:285-324``: 24 kHz samples ``[n]`` (or ``[B, n]``) -> x-vector ``[B, enc_dim]``: the fused mel front end
        (``dsp.mel_spectrogram``: n_fft 1024, hop 256, 128 Slaney mels up to 12 kHz) then the ECAPA encoder, all on the device."""
        if sr != 24000:
            raise ValueError("Only 24kHz audio is supported for speaker embedding extraction")
        if self.speaker_encoder is None:
            raise ValueError("Speaker encoder not available for this model type")
        from ....dsp import mel_spectrogram

        mels = mel_spectrogram(audio, n_fft=1024, num_mels=128, sample_rate=24000, hop_size=256, win_size=1024, fmin=0, fmax=12000)
        return self.speaker_encoder(mels)

    def _prepare_generation_inputs(self, text: str, language: str = "auto", speaker: Optional[str] = None, ref_audio=None,
                                   ref_text: Optional[str] = None, instruct: Optional[str] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """``qwen3_tts.py:326-484``: (input_embeds [1, L, H], trailing_text_hidden [1, T, H], tts_pad_embed [1, 1, H])."""
        if self.tokenizer is None:
            raise ValueError("Tokenizer not loaded. Call post_load_hook first.")
        cfg = self.config.talker_config
        chat = f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"
        text_embed = self._text_embed(list(self.tokenizer.encode(chat)))
        tts = self._text_embed([self.config.tts_bos_token_id, self.config.tts_eos_token_id, self.config.tts_pad_token_id])
        tts_bos, tts_eos, tts_pad = tts[:, 0:1], tts[:, 1:2], tts[:, 2:3]
        speaker_embed = None
        if ref_audio is not None and self.speaker_encoder is not None:   # x-vector cloning without a transcript (:383-384)
            speaker_embed = self.extract_speaker_embedding(ref_audio).to(torch.float32)
        elif speaker and speaker.lower() in (cfg.spk_id or {}):
            sid = cfg.spk_id[speaker.lower()]
            speaker_embed = self._codec_embed([sid[0] if isinstance(sid, (list, tuple)) else sid])
        language_id = None
        if language.lower() != "auto" and cfg.codec_language_id and language.lower() in cfg.codec_language_id:
            language_id = cfg.codec_language_id[language.lower()]
        if language.lower() in ("chinese", "auto") and speaker and (cfg.spk_is_dialect or {}).get(speaker.lower()):
            dialect = cfg.spk_is_dialect[speaker.lower()]
            if dialect in (cfg.codec_language_id or {}):
                language_id = cfg.codec_language_id[dialect]
        if language_id is None:
            prefill = [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id]
        else:
            prefill = [cfg.codec_think_id, cfg.codec_think_bos_id, language_id, cfg.codec_think_eos_id]
        codec_embed = self._codec_embed(prefill)
        suffix = self._codec_embed([cfg.codec_pad_id, cfg.codec_bos_id])
        codec_embed = torch.cat([codec_embed] + ([speaker_embed.reshape(1, 1, -1)] if speaker_embed is not None else []) + [suffix], dim=1)
        instruct_embed = None
        if instruct:
            instruct_embed = self._text_embed(list(self.tokenizer.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n")))
        role = text_embed[:, :3]
        pad_count = codec_embed.shape[1] - 2
        combined = torch.cat([tts_pad.expand(1, pad_count, -1), tts_bos], dim=1) + codec_embed[:, :-1]
        parts = ([instruct_embed] if instruct_embed is not None else []) + [role, combined, text_embed[:, 3:4] + codec_embed[:, -1:]]
        input_embeds = torch.cat(parts, dim=1)
        trailing = torch.cat([text_embed[:, 4:-5], tts_eos], dim=1)
        return input_embeds.contiguous(), trailing.contiguous(), tts_pad.contiguous()

    def _prepare_icl_generation_inputs(self, text: str, ref_audio, ref_text: str, language: str = "auto"):
        """``qwen3_tts.py:606-803`` (the official ``generate_icl_prompt`` in its non-streaming form): (input_embeds, trailing_text_hidden = tts_pad,
        tts_pad_embed, ref_codes [1, groups, ref_time]).  Prefill = role (3 tokens) | codec prefix (think / no-think, language id, x-vector, pad, bos)
        under tts_pad ... tts_bos | all text (reference transcript + target text + tts_eos) over codec_pad | codec_bos + the reference clip's frames
        (sum of the codec embeddings of all groups) over tts_pad.  Codes and transcript ids of a clip are cached by (transcript, clip fingerprint)."""
        if self.tokenizer is None:
            raise ValueError("Tokenizer not loaded. Call post_load_hook first.")
        cfg = self.config.talker_config
        ref_audio = torch.as_tensor(ref_audio)
        key = (ref_text, (int(ref_audio.numel()), float(ref_audio.sum())))
        ref_codes, ref_text_ids = self._icl_cache.get(key, (None, None))
        audio_for_spk = ref_audio
        if ref_codes is None:
            a = ref_audio[None, None, :] if ref_audio.dim() == 1 else (ref_audio[None, :] if ref_audio.dim() == 2 else ref_audio)
            ref_codes = self.speech_tokenizer.encode(a)                          # [1, groups, ref_time]
            ref_text_ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{ref_text}<|im_end|>\n"))[3:-2]
            self._icl_cache[key] = (ref_codes, ref_text_ids)
        target_ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"))
        text_ids = target_ids[3:-5]
        tts = self._text_embed([self.config.tts_bos_token_id, self.config.tts_eos_token_id, self.config.tts_pad_token_id])
        tts_bos, tts_eos, tts_pad = tts[:, 0:1], tts[:, 1:2], tts[:, 2:3]
        text_embed = torch.cat([self._text_embed(list(ref_text_ids) + text_ids), tts_eos], dim=1)
        codes_t = torch.as_tensor(ref_codes).permute(0, 2, 1)                    # [1, ref_time, groups]
        codec_icl = torch.cat([self._codec_embed([cfg.codec_bos_id]), self._codes_embed(codes_t)], dim=1)
        icl = torch.cat([text_embed + self._codec_embed([cfg.codec_pad_id]), codec_icl + tts_pad], dim=1)
        language_id = None
        if language.lower() != "auto" and cfg.codec_language_id and language.lower() in cfg.codec_language_id:
            language_id = cfg.codec_language_id[language.lower()]
        speaker_embed = self.extract_speaker_embedding(audio_for_spk) if self.speaker_encoder is not None else None   # ICL still uses the x-vector (:743-746)
        if language_id is None:
            prefill = [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id]
        else:
            prefill = [cfg.codec_think_id, cfg.codec_think_bos_id, language_id, cfg.codec_think_eos_id]
        prefix = torch.cat([self._codec_embed(prefill)] + ([speaker_embed.to(torch.float32).reshape(1, 1, -1)] if speaker_embed is not None else [])
                           + [self._codec_embed([cfg.codec_pad_id, cfg.codec_bos_id])], dim=1)
        role = self._text_embed(target_ids[:3])
        combined = torch.cat([tts_pad.expand(1, prefix.shape[1] - 2, -1), tts_bos], dim=1) + prefix[:, :-1]
        input_embeds = torch.cat([role, combined, icl], dim=1)
        return input_embeds.contiguous(), tts_pad.contiguous(), tts_pad.contiguous(), ref_codes

    def _prepare_batch_inputs(self, texts: List[str], language: str = "auto", speakers: Optional[List[Optional[str]]] = None,
                              instructs: Optional[List[Optional[str]]] = None, ref_audio=None, ref_text: Optional[str] = None,
                              return_metadata: bool = False):
        """``qwen3_tts.py:486-604``: per-sequence inputs, input_embeds LEFT-padded with zeros, trailing text RIGHT-padded with tts_pad."""
        use_icl = ref_audio is not None and ref_text is not None
        embeds, trailings, pad, shared_ref_codes = [], [], None, None
        for i, t in enumerate(texts):
            if use_icl:   # one shared reference clip for the whole batch
                e, tr, p, rc = self._prepare_icl_generation_inputs(t, ref_audio=ref_audio, ref_text=ref_text, language=language)
                shared_ref_codes = rc if shared_ref_codes is None else shared_ref_codes
            else:
                e, tr, p = self._prepare_generation_inputs(t, language=language, speaker=speakers[i] if speakers else None,
                                                           instruct=instructs[i] if instructs else None)
            embeds.append(e)
            trailings.append(tr)
            pad = p if pad is None else pad
        H = embeds[0].shape[-1]
        plens = [e.shape[1] for e in embeds]
        mp = max(plens)
        left = [mp - n for n in plens]
        dev = embeds[0].device
        x = torch.cat([torch.cat([torch.zeros((1, l, H), device=dev), e], dim=1) for e, l in zip(embeds, left)], dim=0)
        mask = torch.cat([torch.cat([torch.zeros((1, l), device=dev), torch.ones((1, n), device=dev)], dim=1) for n, l in zip(plens, left)], dim=0)
        tlens = [t.shape[1] for t in trailings]
        mt = max(tlens)
        tr = torch.cat([torch.cat([t, pad.expand(1, mt - t.shape[1], H)], dim=1) for t in trailings], dim=0)
        bi = Qwen3BatchInputs(input_embeds=x.contiguous(), trailing_text_hidden=tr.contiguous(), tts_pad_embed=pad, attention_mask=mask,
                              left_padding=left, prefill_lens=plens, trailing_lens=tlens, ref_codes=shared_ref_codes)
        return bi if return_metadata else (bi.input_embeds, bi.trailing_text_hidden, bi.tts_pad_embed, bi.attention_mask)

    # ------------------------------------------------------------------ decode
    def _decode_generated_codes(self, codes: torch.Tensor, *, decode_chunk: int = 15, decode_ctx: int = 5) -> torch.Tensor:
        """codes int [T, num_code_groups] of ONE sequence -> waveform [samples] with bounded decoder memory (``qwen3_tts.py:1050-1083``): chunks
        of ``decode_chunk`` frames with ``decode_ctx`` frames of left context whose audio is trimmed."""
        dec = self.speech_tokenizer.decoder
        if codes.numel() == 0:
            return torch.zeros(0, dtype=torch.float32, device=dec.device)
        up = dec.total_upsample
        tr = codes.t()[None].contiguous()  # [1, groups, T]
        parts, start, n = [], 0, tr.shape[-1]
        while start < n:
            end = min(start + decode_chunk, n)
            ctx = decode_ctx if start > decode_ctx else start
            wav = dec(tr[..., start - ctx:end]).squeeze(1)[0]
            parts.append(wav[ctx * up:] if ctx > 0 else wav)
            start = end
        return torch.cat(parts) if len(parts) > 1 else parts[0]

    def _decode_icl_generated_codes(self, codes: torch.Tensor, ref_codes: torch.Tensor) -> torch.Tensor:
        """``qwen3_tts.py:1085-1112``: generated frames ``[T, groups]`` decoded BEHIND the reference clip's frames (the codec decoder is causal: the
        clip is the acoustic left context of the new audio), then the clip's share of the samples -- ``ref_len / total_len`` of them -- is cut off."""
        dec_dev = self.speech_tokenizer.decoder.device
        if codes.numel() == 0:
            return torch.zeros(0, dtype=torch.float32, device=dec_dev)
        ref_t = torch.as_tensor(ref_codes).permute(0, 2, 1).to(codes.device, codes.dtype)     # [1, ref_len, groups]
        full = torch.cat([ref_t, codes[None]], dim=1).contiguous()
        ref_len, total_len = int(ref_t.shape[1]), int(full.shape[1])
        audio, lengths = self.speech_tokenizer.decode(full)
        audio = audio[0]
        valid = int(lengths[0])
        if 0 < valid < audio.shape[0]:
            audio = audio[:valid]
        cut = int(ref_len / max(total_len, 1) * audio.shape[0])
        if 0 < cut < audio.shape[0]:
            audio = audio[cut:]
        return audio

    def _result(self, audio: torch.Tensor, segment_idx: int, token_count: int, elapsed: float, **extra) -> GenerationResult:
        samples = int(audio.shape[0])
        dur = samples / self.sample_rate
        return GenerationResult(
            audio=audio, samples=samples, sample_rate=self.sample_rate, segment_idx=segment_idx, token_count=token_count,
            audio_duration=format_duration(dur), real_time_factor=dur / elapsed if elapsed > 0 else 0,
            prompt={"tokens": token_count, "tokens-per-sec": token_count / elapsed if elapsed > 0 else 0},
            audio_samples={"samples": samples, "samples-per-sec": samples / elapsed if elapsed > 0 else 0},
            processing_time_seconds=elapsed, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0, **extra)

    def _frame_loop(self, input_embeds, trailing, tts_pad, max_tokens, *, temperature, top_k, top_p, repetition_penalty, left_pad=None, seed=None,
                    **engine_kw):
        gen = None
        if temperature > 0 and "gumbel0" not in engine_kw:
            gen = torch.Generator(device=self.talker.device)
            gen.manual_seed(int(seed) if seed is not None else int(torch.seed() % (2 ** 31)))
        cap = self.talker.talker.cos.shape[0] - input_embeds.shape[1]
        if engine_kw.get("chunk"):   # streaming: the generator itself (blocks of frames while the loop runs, then the final dict)
            return self.talker.generate_iter(input_embeds, trailing, tts_pad, min(max_tokens, cap), temperature=temperature, top_k=top_k, top_p=top_p,
                                             repetition_penalty=repetition_penalty, left_pad=left_pad, generator=gen, **engine_kw)
        return self.talker.generate(input_embeds, trailing, tts_pad, min(max_tokens, cap), temperature=temperature, top_k=top_k, top_p=top_p,
                                    repetition_penalty=repetition_penalty, left_pad=left_pad, generator=gen, **engine_kw)

    def _stream_blocks(self, blocks, segment_idx: int, prime_codes: Optional[torch.Tensor] = None):
        """qwen3_tts.py:1426-1465: every block of new frames goes through ``decoder.streaming_step`` (conv buffers + transformer KV cache carried from
        block to block: speech_tokenizer.py:882-930) and leaves as a ``GenerationResult`` WHILE the frame loop runs.  ``prime_codes`` [n, G]: frames
        that precede the stream (the reference clip of in-context cloning): decoded into the state, their audio dropped."""
        dec = self.speech_tokenizer.decoder
        st = dec.new_stream(1)
        if prime_codes is not None and prime_codes.shape[0] > 0:
            dec.streaming_step(prime_codes.t()[None].contiguous(), st)
        t0 = time.time()
        for item in blocks:
            if "block" not in item:
                continue
            blk = item["block"][0]
            wav = dec.streaming_step(blk.t()[None].contiguous(), st).squeeze(1)[0]
            torch.cuda.synchronize()
            yield self._result(wav, segment_idx, int(blk.shape[0]), time.time() - t0, is_streaming_chunk=True, is_final_chunk=bool(item.get("last")))
            t0 = time.time()

    # ------------------------------------------------------------------ generate
    def generate(self, text: str, voice: Optional[str] = None, instruct: Optional[str] = None, temperature: float = 0.9, speed: float = 1.0,
                 lang_code: str = "auto", ref_audio=None, ref_text: Optional[str] = None, split_pattern: str = "\n", max_tokens: int = 4096,
                 verbose: bool = False, stream: bool = False, streaming_interval: float = 2.0, streaming_context_size: int = 25, top_k: int = 50,
                 top_p: float = 1.0, repetition_penalty: float = 1.05, **kwargs) -> Generator[GenerationResult, None, None]:
        """``qwen3_tts.py:1122-1575``: one ``GenerationResult`` per text segment (``stream=True``: one per ``streaming_interval`` of audio plus the
        final chunk).  Routing by ``config.tts_model_type`` as in the reference: ``voice_design`` needs ``instruct``, ``custom_voice`` needs
        ``voice``; ``base`` accepts an optional preset ``voice``.  Engine-level keyword arguments (``seed``, ``gumbel0``, ``gumbel_cp``,
        ``forced_codes``) pass through for deterministic runs; other unknown keyword arguments are ignored like the reference does."""
        if self.talker is None:
            raise RuntimeError("Model has no weights: call load_weights() (or mlx_audio_amd.tts.utils.load_model)")
        if self.speech_tokenizer is None:
            raise ValueError("Speech tokenizer not loaded")
        if ref_audio is not None and isinstance(ref_audio, (str, Path)):
            from ....utils import load_audio

            ref_audio = load_audio(str(ref_audio), sample_rate=self.sample_rate)
        kind = getattr(self.config, "tts_model_type", "base")
        if kind == "voice_design" and not instruct:
            raise ValueError("VoiceDesign model requires 'instruct' to describe the voice (e.g., 'A cheerful young female voice with high pitch')")
        if kind == "custom_voice" and not voice:
            raise ValueError(f"CustomVoice model requires 'voice' (speaker name) (e.g., {self.supported_speakers})")
        if kind == "base":
            instruct = None
        engine_kw = {k: kwargs[k] for k in ("gumbel0", "gumbel_cp", "forced_codes") if k in kwargs}
        if kind == "base" and ref_audio is not None and ref_text is not None and self.speech_tokenizer.has_encoder:
            # in-context cloning (:1227-1250); the stronger repetition penalty keeps long reference prefills from degenerating
            yield from self._generate_icl(text, ref_audio, ref_text, language=lang_code, temperature=temperature, max_tokens=max_tokens, top_k=top_k,
                                          top_p=top_p, repetition_penalty=max(repetition_penalty, 1.5), stream=stream, streaming_interval=streaming_interval,
                                          streaming_context_size=streaming_context_size, seed=kwargs.get("seed"), **engine_kw)
            return
        if kind == "base" and voice is not None and voice.lower() not in [s.lower() for s in self.supported_speakers]:
            raise ValueError(f"Voice '{voice}' is not supported by this Base model. Base models have no built-in preset voices — clone a voice by "
                             "passing ref_audio and ref_text instead."
                             + (f" Available preset voices: {self.supported_speakers}" if self.supported_speakers else ""))
        clip = ref_audio if kind == "base" else None   # only the Base route hands the clip on (x-vector cloning, :1290-1297)
        segments = [s.strip() for s in text.split(split_pattern) if s.strip()] if split_pattern else [text]
        for segment_idx, seg in enumerate(segments):
            t0 = time.time()
            x, trailing, pad = self._prepare_generation_inputs(seg, language=lang_code, speaker=voice if kind != "voice_design" else None, ref_audio=clip,
                                                               ref_text=ref_text if kind == "base" else None, instruct=instruct)
            if stream:
                blocks = self._frame_loop(x, trailing, pad, max_tokens, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                                          seed=kwargs.get("seed"), pad_when_index_clamped=False, chunk=max(1, int(streaming_interval * 12.5)), **engine_kw)
                yield from self._stream_blocks(blocks, segment_idx)
                continue
            out = self._frame_loop(x, trailing, pad, max_tokens, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                                   seed=kwargs.get("seed"), pad_when_index_clamped=False, **engine_kw)
            codes = out["codes"][0]
            fa = int(out["finished_at"][0])
            codes = codes[:fa] if fa >= 0 else codes  # the EOS frame itself is not decoded (qwen3_tts.py:1408-1412)
            if codes.shape[0] == 0:
                continue
            audio, lengths = self.speech_tokenizer.decode(codes[None])
            audio = audio[0]
            valid = int(lengths[0])
            if 0 < valid < audio.shape[0]:
                audio = audio[:valid]
            torch.cuda.synchronize()
            yield self._result(audio, segment_idx, int(codes.shape[0]), time.time() - t0)

    def _generate_icl(self, text: str, ref_audio, ref_text: str, language: str = "auto", temperature: float = 0.9, max_tokens: int = 4096, top_k: int = 50,
                      top_p: float = 1.0, repetition_penalty: float = 1.5, stream: bool = False, streaming_interval: float = 2.0,
                      streaming_context_size: int = 25, seed=None, prime_stream_with_reference: bool = False,
                      **engine_kw) -> Generator[GenerationResult, None, None]:
        """``qwen3_tts.py:2200-2510``: the whole text as ONE segment behind the in-context prompt; the frame loop is the engine's (prefill of the
        prompt, then a talker step + 15 code-predictor steps per frame); decode behind the reference codes and cut them off.  ``stream=True``
        yields chunks of the NEW audio only, decoded by ``streaming_step`` on a FRESH decoder state fed with the generated codes alone -- the
        reference's behaviour (qwen3_tts.py:2266-2444: ``reset_streaming_state()`` then ``streaming_step`` per chunk, no priming).
        ``prime_stream_with_reference=True`` is a deliberate DEVIATION kept behind this flag: the state is first run over the reference clip's
        codes (their audio dropped), so the first streamed chunk starts with the conv / attention context the one-shot decode gives it."""
        t0 = time.time()
        x, trailing, pad, ref_codes = self._prepare_icl_generation_inputs(text, ref_audio=ref_audio, ref_text=ref_text, language=language)
        if stream:
            blocks = self._frame_loop(x, trailing, pad, max_tokens, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                                      seed=seed, pad_when_index_clamped=False, chunk=max(1, int(streaming_interval * 12.5)), **engine_kw)
            rc = None if ref_codes is None else torch.as_tensor(ref_codes)[0].t()   # [1, groups, ref_time] -> frames [ref_time, groups]
            yield from self._stream_blocks(blocks, 0, prime_codes=rc if prime_stream_with_reference else None)
            return
        out = self._frame_loop(x, trailing, pad, max_tokens, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                               seed=seed, pad_when_index_clamped=False, **engine_kw)
        codes = out["codes"][0]
        fa = int(out["finished_at"][0])
        codes = codes[:fa] if fa >= 0 else codes
        if codes.shape[0] == 0:
            return
        audio = self._decode_icl_generated_codes(codes, ref_codes)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        yield self._result(audio, 0, int(codes.shape[0]), time.time() - t0)

    def generate_custom_voice(self, text: str, speaker: str, language: str = "auto", instruct: Optional[str] = None, **kw):
        """``qwen3_tts.py:2062-2137``."""
        if getattr(self.config, "tts_model_type", "base") != "custom_voice":
            raise ValueError("generate_custom_voice needs a CustomVoice checkpoint")
        yield from self.generate(text, voice=speaker, instruct=instruct, lang_code=language, **kw)

    def generate_voice_design(self, text: str, instruct: str, language: str = "auto", **kw):
        """``qwen3_tts.py:2139-2198``."""
        if getattr(self.config, "tts_model_type", "base") != "voice_design":
            raise ValueError("generate_voice_design needs a VoiceDesign checkpoint")
        yield from self.generate(text, instruct=instruct, lang_code=language, **kw)

    @staticmethod
    def _same_shared_ref_value(left, right) -> bool:
        if isinstance(left, (str, Path)) and isinstance(right, (str, Path)):
            return str(left) == str(right)
        return left is right

    def _normalize_shared_batch_refs(self, batch_size: int, *, ref_audio=None, ref_text: Optional[str] = None, ref_audios=None, ref_texts=None):
        """``qwen3_tts.py:1582-1649``: ONE shared (clip, transcript) pair for a whole batch, given directly or as per-item lists that must all name
        the same reference; both halves or neither."""
        def shared_from_list(name, values):
            if values is None:
                return None
            if len(values) != batch_size:
                raise ValueError(f"{name} length ({len(values)}) must match texts length ({batch_size})")
            present = [v for v in values if v is not None]
            if not present:
                return None
            if len(present) != batch_size:
                raise ValueError(f"Qwen3-TTS batch_generate requires {name} for every text when using reference cloning")
            shared = present[0]
            for v in present[1:]:
                if not self._same_shared_ref_value(shared, v):
                    raise ValueError(f"Qwen3-TTS batch_generate currently supports only one shared {name[:-1]} across the whole batch")
            return shared

        list_audio, list_text = shared_from_list("ref_audios", ref_audios), shared_from_list("ref_texts", ref_texts)
        if list_audio is not None:
            if ref_audio is not None and not self._same_shared_ref_value(ref_audio, list_audio):
                raise ValueError("ref_audio and ref_audios must refer to the same shared reference")
            ref_audio = list_audio
        if list_text is not None:
            if ref_text is not None and ref_text != list_text:
                raise ValueError("ref_text and ref_texts must refer to the same shared reference")
            ref_text = list_text
        if ref_audio is None and ref_text is None:
            return None, None
        if ref_audio is None or ref_text is None:
            raise ValueError("Qwen3-TTS batch reference cloning requires both ref_audio and ref_text")
        if isinstance(ref_audio, (str, Path)):
            from ....utils import load_audio

            ref_audio = load_audio(str(ref_audio), sample_rate=self.sample_rate)
        return ref_audio, ref_text

    def batch_generate(self, texts: List[str], voices: Optional[List[Optional[str]]] = None, instructs: Optional[List[Optional[str]]] = None,
                       ref_audio=None, ref_text: Optional[str] = None, ref_audios=None, ref_texts=None, temperature: float = 0.9,
                       lang_code: str = "auto", max_tokens: int = 4096, top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.05,
                       stream: bool = False, streaming_interval: float = 2.0, streaming_context_size: int = 25, verbose: bool = False,
                       **kwargs) -> Generator[BatchGenerationResult, None, None]:
        """All texts in ONE batched frame loop (``qwen3_tts.py:1651-2060``): left-padded prompts, finished rows emit EOS, per-sequence chunked
        codec decode; yields one ``BatchGenerationResult`` per sequence in input order."""
        if self.speech_tokenizer is None:
            raise ValueError("Speech tokenizer not loaded")
        if not texts:
            return
        for name, lst in (("voices", voices), ("instructs", instructs)):
            if lst is not None and len(lst) != len(texts):
                raise ValueError(f"{name} length ({len(lst)}) must match texts length ({len(texts)})")
        ref_audio, ref_text = self._normalize_shared_batch_refs(len(texts), ref_audio=ref_audio, ref_text=ref_text, ref_audios=ref_audios, ref_texts=ref_texts)
        use_icl = ref_audio is not None and ref_text is not None
        caps = [max_tokens] * len(texts)
        if use_icl:   # one shared reference clip in front of every sequence (:1724-1741, :1823-1827)
            if not self.speech_tokenizer.has_encoder:
                raise ValueError("Qwen3-TTS batch reference cloning requires a speech tokenizer encoder")
            if any(v is not None for v in (voices or [])):
                raise ValueError("Qwen3-TTS batch reference cloning does not support voices")
            if any(v is not None for v in (instructs or [])):
                raise ValueError("Qwen3-TTS batch reference cloning does not support instructs")
            repetition_penalty = max(repetition_penalty, 1.5)
            caps = [min(max_tokens, max(75, len(self.tokenizer.encode(t)) * 6)) for t in texts]
        t0 = time.time()
        bi = self._prepare_batch_inputs(texts, language=lang_code, speakers=voices, instructs=instructs, ref_audio=ref_audio, ref_text=ref_text,
                                        return_metadata=True)
        if stream:
            yield from self._batch_generate_stream(bi, caps, t0, temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty,
                                                   streaming_interval=streaming_interval, seed=kwargs.get("seed"), use_icl=use_icl,
                                                   slots=kwargs.get("slots"), codes_log=kwargs.get("codes_log"))
            return
        left = torch.tensor(bi.left_padding, dtype=torch.int32) if len(texts) > 1 else None
        engine_kw = {k: kwargs[k] for k in ("gumbel0", "gumbel_cp", "forced_codes") if k in kwargs}
        out = self._frame_loop(bi.input_embeds, bi.trailing_text_hidden, bi.tts_pad_embed, max(caps), temperature=temperature, top_k=top_k, top_p=top_p,
                               repetition_penalty=repetition_penalty, left_pad=left, seed=kwargs.get("seed"), **engine_kw)
        torch.cuda.synchronize()
        elapsed = time.time() - t0
        fa = out["finished_at"].cpu()
        for b in range(len(texts)):
            n = int(fa[b]) if int(fa[b]) >= 0 else out["codes"].shape[1]
            n = min(n, caps[b])   # a row that reaches its own budget is finished there (rows never influence each other)
            if n == 0:
                continue
            audio = (self._decode_icl_generated_codes(out["codes"][b, :n], bi.ref_codes) if use_icl else self._decode_generated_codes(out["codes"][b, :n]))
            yield BatchGenerationResult(audio=audio, sequence_idx=b, samples=int(audio.shape[0]), sample_rate=self.sample_rate, token_count=n,
                                        audio_duration=format_duration(audio.shape[0] / self.sample_rate), processing_time_seconds=elapsed,
                                        peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9)

    def _batch_generate_stream(self, bi: Qwen3BatchInputs, caps: List[int], t0: float, *, temperature: float, top_k: int, top_p: float,
                               repetition_penalty: float, streaming_interval: float, seed=None, use_icl: bool = False, slots=None,
                               codes_log: Optional[dict] = None):
        """The streaming half of ``batch_generate`` (``qwen3_tts.py:1845-1853, 1935-2010``): every sequence owns a slot of the step-wise engine
        (``Qwen3TalkerSlots``: one prefill for the whole batch, then one talker step + the code-predictor steps per frame for all rows); whenever a
        sequence has ``int(streaming_interval * 12.5)`` new frames they are decoded behind up to 25 frames of its own left context (none for its first
        chunk), the context's samples are cut off and the chunk is yielded; what is left when every row is finished (EOS, its frame budget, the end
        of the rotary tables) goes out flagged ``is_final_chunk`` -- like the reference, a sequence that ends exactly on a chunk boundary gets no
        final flag, except in the in-context batch when the step that exhausts the last open budget also completes a chunk (the reference leaves
        its loop before the emission there, :1926-1933).  ``slots`` / ``codes_log``: test hooks (a scripted engine; sequence index -> all frames)."""
        B = bi.input_embeds.shape[0]
        if slots is None:
            from .talker import Qwen3TalkerSlots

            if B > self.talker.talker.max_decode_rows:
                raise NotImplementedError(f"batch_generate(stream=True) steps at most {self.talker.talker.max_decode_rows} sequences at once")
            gen = None
            if temperature > 0:
                gen = torch.Generator(device=self.talker.device)
                gen.manual_seed(int(seed) if seed is not None else int(torch.seed() % (2 ** 31)))
            rows = self.talker.talker.cos.shape[0]
            slots = Qwen3TalkerSlots(self.talker, B, max(1, min(max(caps), rows - 1)), temperature=temperature, top_k=top_k, top_p=top_p,
                                     repetition_penalty=repetition_penalty, generator=gen)
        chunk = max(1, int(streaming_interval * 12.5))
        ctx_max = 25                                   # the vocoder's left context (qwen3_tts.py:1846-1847: the argument is not consulted)
        up = self.speech_tokenizer.decode_upsample_rate
        frames, decoded = [0] * B, [0] * B

        def emit(b: int, final: bool):
            new = frames[b] - decoded[b]
            ctx = 0 if decoded[b] == 0 else min(ctx_max, decoded[b])
            codes = slots.take_codes(b, frames[b])[decoded[b] - ctx:]
            wav = self.speech_tokenizer.decoder.chunked_decode(codes.t()[None].contiguous()).squeeze(1)[0]
            if ctx > 0 and ctx * up < wav.shape[0]:
                wav = wav[ctx * up:]
            decoded[b] = frames[b]
            extra = dict(is_final_chunk=True) if final else {}
            return BatchGenerationResult(audio=wav, sequence_idx=b, samples=int(wav.shape[0]), sample_rate=self.sample_rate, token_count=new,
                                         audio_duration=format_duration(wav.shape[0] / self.sample_rate), processing_time_seconds=time.time() - t0,
                                         peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0,
                                         is_streaming_chunk=True, **extra)

        fin = slots.admit(list(range(B)), bi.input_embeds, bi.left_padding, bi.trailing_text_hidden, bi.tts_pad_embed)
        done = [False] * B
        for b in range(B):
            frames[b] = 0 if fin[b] else 1
            done[b] = bool(fin[b]) or frames[b] >= min(caps[b], slots.limit[b])
        while True:
            if not (use_icl and all(done)):
                for b in range(B):
                    if frames[b] - decoded[b] >= chunk:
                        yield emit(b, False)
            if all(done):
                break
            fin = slots.advance(B)
            for b in range(B):
                if done[b]:
                    continue
                if not fin[b]:
                    frames[b] += 1
                done[b] = bool(fin[b]) or frames[b] >= min(caps[b], slots.limit[b])
        for b in range(B):
            if frames[b] > decoded[b]:
                yield emit(b, True)
        if codes_log is not None:
            for b in range(B):
                codes_log[b] = slots.take_codes(b, frames[b])
        for b in range(B):
            slots.release(b)

    @classmethod
    def from_pretrained(cls, path: Union[str, Path]) -> "Model":
        from ...utils import load

        return load(path)
