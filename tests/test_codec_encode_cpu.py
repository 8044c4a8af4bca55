"""Codec ENCODE oracles against the reference's own modules: tests/golden/make_reference_fixtures.py ``codec_encode`` runs
``DAC.encode`` / ``SNAC.encode`` / ``Encodec.encode`` of /root/reference over the MLX stand-in on seeded checkpoints; the restatements in ``oracle/`` must
reproduce every code and the float tensors at float32 rounding level.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def dac_model_weights(fx):
    from mlx_audio_amd.codec.models.descript import make_dac_encoder_weights, make_dac_weights

    c = json.loads(str(fx["config"]))
    w = make_dac_weights(c["decoder_dim"], c["decoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_size"], c["codebook_dim"], seed=int(fx["seed_w"]))
    w.update(make_dac_encoder_weights(c["encoder_dim"], c["encoder_rates"], c["latent_dim"], c["n_codebooks"], c["codebook_dim"], seed=int(fx["seed_w"])))
    return c, w


def test_dac_encode_oracle_reproduces_the_reference_modules():
    """``DAC.encode`` (dac.py:184-192): encoder output, every code of every codebook, latents, z_q, both losses; ``n_quantizers = 2``; and the
    encode half of ``DAC.__call__`` (preprocess pads to a whole number of hops first: one more frame)."""
    from oracle.dac_ref import DACDecoderRef, DACEncoderRef

    fx = np.load(os.path.join(GOLD, "ref_dac_encode.npz"))
    c, w = dac_model_weights(fx)
    ref = DACEncoderRef(w, c["encoder_rates"], c["n_codebooks"])
    audio = torch.from_numpy(fx["audio"])
    enc = ref.encoder(audio)
    assert tuple(enc.shape) == fx["enc"].shape and rel_max(enc.numpy(), fx["enc"]) < 2e-5
    z, codes, latents, commit, cbl, margins = ref.quantize(enc, return_margins=True)
    assert np.array_equal(codes.numpy(), fx["codes"]), int((codes.numpy() != fx["codes"]).sum())
    assert rel_max(latents.numpy(), fx["latents"]) < 2e-5 and rel_max(z.numpy(), fx["z"]) < 2e-5
    assert abs(float(commit) - float(fx["commitment_loss"])) < 1e-5 * float(fx["commitment_loss"])
    assert abs(float(cbl) - float(fx["codebook_loss"])) < 1e-5 * float(fx["codebook_loss"])
    assert float(margins.min()) >= 0.0 and tuple(margins.shape) == tuple(codes.shape)
    z2, codes2, latents2, commit2, _ = ref.quantize(enc, 2)
    assert np.array_equal(codes2.numpy(), fx["codes_nq2"]) and rel_max(latents2.numpy(), fx["latents_nq2"]) < 2e-5
    assert rel_max(z2.numpy(), fx["z_nq2"]) < 2e-5 and abs(float(commit2) - float(fx["commitment_loss_nq2"])) < 1e-5 * float(fx["commitment_loss_nq2"])
    # __call__: right-pad to a multiple of hop_length (dac.py:173-182), encode, decode
    hop = int(np.prod(c["encoder_rates"]))
    pad = (-audio.shape[-1]) % hop
    zc, cc, *_ = ref.encode(torch.nn.functional.pad(audio, (0, pad)))
    assert np.array_equal(cc.numpy(), fx["call_codes"]) and rel_max(zc.numpy(), fx["call_z"]) < 2e-5
    dec = DACDecoderRef(w, c["decoder_rates"], c["n_codebooks"]).decode(zc).numpy()
    assert dec.shape == fx["call_audio"].shape and rel_max(dec, fx["call_audio"]) < 5e-5


def test_dac_encode_host_schedule_dry_run():
    """The product's host schedule (``mlx_audio_amd/codec/models/descript/dac.py``: flattened stem conv, in-place residual units inside the zero-padded
    staging buffers, the strided convs as two taps over regrouped rows, per-codebook in_proj -> search -> negated-table residual update) over the CPU
    emulation of the operator contracts (tests/_ops_emu.py), against the reference's own run: views, pads and operand order are right before a GPU
    is spent.  (The kernels themselves are held by tests/test_codec_encode_gpu.py.)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.descript import DAC
    from oracle.dac_ref import DACEncoderRef

    fx = np.load(os.path.join(GOLD, "ref_dac_encode.npz"))
    c, w = dac_model_weights(fx)
    audio = torch.from_numpy(fx["audio"])
    with _ops_emu.patched():
        eng = DAC(**c, weights=w, device="cpu")
        enc, st = eng.encoder(audio, return_stages=True)
        _, est = DACEncoderRef(w, c["encoder_rates"], c["n_codebooks"]).encoder(audio, return_stages=True)
        for k in est:
            assert rel_max(st[k].numpy(), est[k].numpy()) < 1e-5, k
        assert rel_max(enc.numpy(), fx["enc"]) < 1e-5
        z, codes, latents, commit, cbl, margins = eng.quantizer(torch.from_numpy(fx["enc"]), return_margins=True)
        assert np.array_equal(codes.numpy(), fx["codes"]) and codes.dtype == torch.int64
        assert rel_max(z.numpy(), fx["z"]) < 1e-5 and rel_max(latents.numpy(), fx["latents"]) < 1e-5
        assert abs(float(commit) - float(fx["commitment_loss"])) < 1e-5 * float(fx["commitment_loss"]) and float(cbl) == float(commit)
        assert float(margins.min()) > 0 and tuple(margins.shape) == tuple(codes.shape)
        z2, codes2, latents2, commit2, _ = eng.quantizer(torch.from_numpy(fx["enc"]), 2)
        assert np.array_equal(codes2.numpy(), fx["codes_nq2"]) and rel_max(z2.numpy(), fx["z_nq2"]) < 1e-5 and rel_max(latents2.numpy(), fx["latents_nq2"]) < 1e-5
        out = eng(audio, c["sample_rate"])
        assert np.array_equal(out["codes"].numpy(), fx["call_codes"]) and rel_max(out["z"].numpy(), fx["call_z"]) < 1e-5
        assert out["audio"].shape == fx["call_audio"].shape and rel_max(out["audio"].numpy(), fx["call_audio"]) < 2e-5
        # use_rvq = False: encoder -> decoder without the quantizer (the reference then leaves codes / latents / losses unbound and raises at its return
        # statement, dac.py:218-239; here they are None); return_loss = True: a scalar
        raw = eng(audio, c["sample_rate"], use_rvq=False)
        assert raw["codes"] is None and raw["latents"] is None and tuple(raw["z"].shape) == fx["call_z"].shape and raw["audio"].shape == fx["call_audio"].shape
        assert rel_max(raw["z"].numpy(), eng.encoder(eng.preprocess(audio, c["sample_rate"])).numpy()) == 0.0
        loss = eng(audio, c["sample_rate"], return_loss=True)
        assert loss.dim() == 0 and float(loss) > 0 and np.isfinite(float(loss))
    import mlx_audio_amd.ops as real_ops
    assert real_ops.conv_gemm.__module__ == "mlx_audio_amd.ops"   # the emulation is gone after the block


def snac_model_weights(fx):
    from mlx_audio_amd.codec.models.snac import make_snac_encoder_weights, make_snac_weights

    c = json.loads(str(fx["config"]))
    latent = c["encoder_dim"] * 2 ** len(c["encoder_rates"])
    w = make_snac_weights(latent, c["decoder_dim"], c["decoder_rates"], c["vq_strides"], c["codebook_size"], c["codebook_dim"], True, c["depthwise"], seed=int(fx["seed_w"]))
    w.update(make_snac_encoder_weights(c["encoder_dim"], c["encoder_rates"], latent, c["vq_strides"], c["codebook_dim"], c["depthwise"], seed=int(fx["seed_w"])))
    return c, w


def test_snac_encode_oracle_reproduces_the_reference_modules():
    """``SNAC.encode`` (snac.py:96-102): padded length, encoder output, every code of the three levels (strides 4 / 2 / 1), z_q -- depthwise and dense."""
    from oracle.snac_ref import SNACEncoderRef

    for kind in ("dw", "dense"):
        fx = np.load(os.path.join(GOLD, f"ref_snac_encode_{kind}.npz"))
        c, w = snac_model_weights(fx)
        ref = SNACEncoderRef(w, c["encoder_rates"], c["vq_strides"], depthwise=c["depthwise"])
        audio = torch.from_numpy(fx["audio"])
        padded = ref.preprocess(audio)
        assert padded.shape[-1] == int(fx["padded_len"])
        z = ref.encoder(padded)
        assert tuple(z.shape) == fx["z"].shape and rel_max(z.numpy(), fx["z"]) < 2e-5
        z_q, codes, margins = ref.quantize(z, return_margins=True)
        for i, cd in enumerate(codes):
            assert np.array_equal(cd.numpy(), fx[f"codes{i}"]), (kind, i)
            assert tuple(margins[i].shape) == tuple(cd.shape) and float(margins[i].min()) >= 0.0
        assert rel_max(z_q.numpy(), fx["z_q"]) < 2e-5
        assert all(np.array_equal(a.numpy(), fx[f"codes{i}"]) for i, a in enumerate(ref.encode(audio)))
        assert "expected input" in str(fx["call_error"])   # the reference's SNAC.__call__ cannot run (decoder handed [B, D, T]): nothing to pin there


def test_snac_encode_host_schedule_dry_run():
    """The product's SNAC encode schedule (staging buffers, depthwise / dense units, strided convs over regrouped rows, average pool folded into the
    in_proj tap, negated-table residual update with repeated ids) over tests/_ops_emu.py, against the reference's own run."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.snac import SNAC
    from oracle.snac_ref import SNACEncoderRef

    for kind in ("dw", "dense"):
        fx = np.load(os.path.join(GOLD, f"ref_snac_encode_{kind}.npz"))
        c, w = snac_model_weights(fx)
        audio = torch.from_numpy(fx["audio"])
        with _ops_emu.patched():
            eng = SNAC(**c, weights=w, device="cpu")
            padded = eng.preprocess(audio)
            assert padded.shape[-1] == int(fx["padded_len"])
            z, st = eng.encoder(padded, return_stages=True)
            _, est = SNACEncoderRef(w, c["encoder_rates"], c["vq_strides"], depthwise=c["depthwise"]).encoder(padded, return_stages=True)
            for k in est:
                assert rel_max(st[k].numpy(), est[k].numpy()) < 1e-5, (kind, k)
            assert rel_max(z.numpy(), fx["z"]) < 1e-5
            z_q, codes, margins = eng.quantizer(torch.from_numpy(fx["z"]), return_margins=True)
            for i, cd in enumerate(codes):
                assert np.array_equal(cd.numpy(), fx[f"codes{i}"]) and cd.dtype == torch.int64, (kind, i)
                assert float(margins[i].min()) > 0
            assert rel_max(z_q.numpy(), fx["z_q"]) < 1e-5
            assert all(np.array_equal(a.numpy(), fx[f"codes{i}"]) for i, a in enumerate(eng.encode(audio)))
            g = torch.Generator().manual_seed(1)
            nz = [torch.randn(2, 1, c["decoder_dim"] // 2 ** (i + 1), generator=g) for i in range(len(c["decoder_rates"]))]
            hat, codes2 = eng(audio, noises=nz)
            assert hat.shape[0] == 2 and hat.shape[2] == 1 and torch.isfinite(hat).all() and all(torch.equal(a, b) for a, b in zip(codes, codes2))
            with pytest.raises(ValueError, match="whole number of stride"):
                eng.quantizer(torch.zeros(1, z.shape[1], 7))


def encodec_model_weights(fx):
    from mlx_audio_amd.codec.models.encodec import make_encodec_encoder_weights, make_encodec_weights

    c = json.loads(str(fx["config"]))
    w = make_encodec_weights(c, seed=int(fx["seed_w"]))
    w.update(make_encodec_encoder_weights(c, seed=int(fx["seed_w"])))
    return c, w


def test_encodec_encode_oracle_reproduces_the_reference_modules():
    """``Encodec.encode`` (encodec.py:585-650): mono causal one-chunk, and stereo non-causal with loudness normalisation and overlapping chunks; every
    code at both bandwidths, the scales, the encoder's embeddings, and the decode of those codes."""
    from oracle.encodec_ref import EncodecRef

    for tag in ("mono", "stereo"):
        fx = np.load(os.path.join(GOLD, f"ref_encodec_encode_{tag}.npz"))
        c, w = encodec_model_weights(fx)
        ref = EncodecRef(w, c)
        x, m = torch.from_numpy(fx["inputs"]), torch.from_numpy(fx["masks"])
        chunk = ref.chunk_length or x.shape[1]
        emb = ref.encoder(x[:, :chunk])
        assert tuple(emb.shape) == fx["embeddings_chunk0_unnormalised"].shape and rel_max(emb.numpy(), fx["embeddings_chunk0_unnormalised"]) < 3e-5, tag
        for bw in c["target_bandwidths"]:
            codes, scales = ref.encode(x, m, bandwidth=bw)
            assert np.array_equal(codes.numpy(), fx[f"codes_bw{bw}"]), (tag, bw, int((codes.numpy() != fx[f"codes_bw{bw}"]).sum()))
            if c["normalize"]:
                assert rel_max(torch.stack(scales).numpy(), fx[f"scales_bw{bw}"]) < 1e-6
            else:
                assert all(s is None for s in scales)
        audio = ref.decode(codes, scales, m).numpy()
        assert audio.shape == fx["decoded"].shape and rel_max(audio, fx["decoded"]) < 5e-5, tag


def test_encodec_encode_host_schedule_dry_run():
    """The product's EnCodec encode schedule (flattened stem over the reflect-padded samples, resnet blocks, strided convs over the padded rows regrouped,
    LSTM, one rvq_encode launch per chunk, normalisation and the chunk loop) over tests/_ops_emu.py, against the reference's own run."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.encodec import Encodec
    from oracle.encodec_ref import EncodecRef

    for tag in ("mono", "stereo"):
        fx = np.load(os.path.join(GOLD, f"ref_encodec_encode_{tag}.npz"))
        c, w = encodec_model_weights(fx)
        x, m = torch.from_numpy(fx["inputs"]), torch.from_numpy(fx["masks"])
        with _ops_emu.patched():
            eng = Encodec(c, weights=w, device="cpu")
            chunk = eng.chunk_length or x.shape[1]
            emb, st = eng._encoder(x[:, :chunk], return_stages=True)
            _, est = EncodecRef(w, c).encoder(x[:, :chunk], return_stages=True)
            for k in est:
                assert rel_max(st[k].numpy(), est[k].numpy()) < 1e-5, (tag, k)
            assert rel_max(emb.numpy(), fx["embeddings_chunk0_unnormalised"]) < 3e-5
            for bw in c["target_bandwidths"]:
                codes, scales = eng.encode(x, m, bandwidth=bw)
                assert codes.dtype == torch.int64 and np.array_equal(codes.numpy(), fx[f"codes_bw{bw}"]), (tag, bw)
                if c["normalize"]:
                    assert rel_max(torch.stack(scales).numpy(), fx[f"scales_bw{bw}"]) < 1e-6
            cm, mg = eng.quantizer.encode(emb, c["target_bandwidths"][-1], return_margins=True)
            assert tuple(mg.shape) == tuple(cm.shape) and float(mg.min()) >= 0
            audio = eng.decode(codes, scales, m)
            assert rel_max(audio.numpy(), fx["decoded"]) < 5e-5
            with pytest.raises(ValueError, match="doesn't support the bandwidth"):
                eng.encode(x, m, bandwidth=7.0)
            if eng.chunk_length is not None:
                with pytest.raises(ValueError, match="not properly padded"):
                    eng.encode(x[:, :-1], m[:, :-1])


def test_vocos_encodec_features_dry_run():
    """``EncodecFeatures`` (codec/models/vocos/vocos.py:54-116) over the emulated operators: ``get_encodec_codes`` returns the reference run's codes in the
    [nq, 1, T] layout, ``get_features_from_codes`` is the sum of the codebook rows; hub names without a local model fail loudly."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.encodec import Encodec
    from mlx_audio_amd.codec.models.vocos import EncodecFeatures

    fx = np.load(os.path.join(GOLD, "ref_encodec_encode_mono.npz"))
    c, w = encodec_model_weights(fx)
    with _ops_emu.patched():
        enc = Encodec(c, weights=w, device="cpu")
        fe = EncodecFeatures(bandwidths=c["target_bandwidths"], encodec=enc)
        assert fe.num_q == 6 and tuple(fe.codebook_weights.shape) == (4 * c["codebook_size"], c["codebook_dim"])
        codes = fe.get_encodec_codes(torch.from_numpy(fx["raw"][:, 0]), bandwidth_id=[1])
        want = fx["codes_bw60.0"]                                  # [chunks = 1, B = 1, nq, T]
        assert tuple(codes.shape) == (want.shape[2], 1, want.shape[3]) and np.array_equal(codes[:, 0].numpy(), want[0, 0])
        feats = fe(torch.from_numpy(fx["raw"][:, 0]), bandwidth_id=torch.tensor([1]))
        rows = sum(w[f"quantizer.layers.{i}.codebook.embed"][codes[i, 0]] for i in range(codes.shape[0]))
        assert tuple(feats.shape) == (1, want.shape[3], c["codebook_dim"]) and rel_max(feats[0].numpy(), rows.numpy()) < 1e-6
        with pytest.raises(ValueError, match="bandwidth_id"):
            fe(torch.zeros(100))
    with pytest.raises(FileNotFoundError, match="not reachable"):
        EncodecFeatures()
    with pytest.raises(ValueError, match="Unsupported encodec_model"):
        EncodecFeatures(encodec_model="encodec_16khz")


def test_encodec_from_pretrained_local_directory(tmp_path):
    """``Encodec.from_pretrained`` (encodec.py:710-737) on a LOCAL directory (``config.json`` with extra keys + ``model.safetensors``): the model and the
    ``preprocess_audio`` partial it returns reproduce the reference run's codes; ``EncodecFeatures(encodec_model=<that directory>)`` builds on it; a path
    that does not exist is a ``FileNotFoundError`` (no hub here)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from safetensors.torch import save_file

    from mlx_audio_amd.codec.models.encodec import Encodec
    from mlx_audio_amd.codec.models.vocos import EncodecFeatures

    fx = np.load(os.path.join(GOLD, "ref_encodec_encode_stereo.npz"))
    c, w = encodec_model_weights(fx)
    d = tmp_path / "encodec-local"
    d.mkdir()
    with open(d / "config.json", "w") as f:
        json.dump(dict(c, transformers_version="4.x", torch_dtype="float32"), f)     # keys outside EncodecConfig are dropped, like filter_dataclass_fields
    save_file({k: v.contiguous() for k, v in w.items()}, str(d / "model.safetensors"))
    with _ops_emu.patched():
        model, processor = Encodec.from_pretrained(str(d), device="cpu")
        assert model.chunk_length == int(c["chunk_length_s"] * c["sampling_rate"]) and model.channels == 2
        inputs, masks = processor(torch.from_numpy(fx["raw"]))
        assert tuple(inputs.shape) == fx["inputs"].shape and np.array_equal(masks.numpy(), fx["masks"])
        bw = c["target_bandwidths"][-1]
        codes, scales = model.encode(inputs, masks, bandwidth=bw)
        assert np.array_equal(codes.numpy(), fx[f"codes_bw{bw}"])
        fe = EncodecFeatures(encodec_model=str(d), bandwidths=c["target_bandwidths"], device="cpu")
        assert fe.encodec.chunk_length == model.chunk_length and fe.num_q >= 1
    with pytest.raises(FileNotFoundError):
        Encodec.from_pretrained(str(tmp_path / "missing"), device="cpu")


def test_snac_preprocess_pads_to_the_attention_window_too():
    """snac.py:67-86: the pad unit is hop_length x lcm(vq_strides) -- and the LocalMHA window joins the least common multiple when the model has one (the
    44 kHz model: hop 441, strides 8 / 4 / 2 / 1, window 32 -> 441 x 32 samples, not 441 x 8)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.snac import SNAC, make_snac_weights

    with _ops_emu.patched():
        w = make_snac_weights(64, 64, [2, 2], [8, 4, 2, 1], 16, 8, True, True, seed=0)
        plain = SNAC(sampling_rate=44100, encoder_dim=16, encoder_rates=[3, 3, 7, 7], latent_dim=64, decoder_dim=64, decoder_rates=[2, 2], attn_window_size=None,
                     codebook_size=16, codebook_dim=8, vq_strides=[8, 4, 2, 1], weights=w, device="cpu")
        assert plain.hop_length == 441 and plain.preprocess(torch.zeros(1, 1, 1000)).shape[-1] == 441 * 8
        plain.attn_window_size = 32
        assert plain.preprocess(torch.zeros(1, 1, 1000)).shape[-1] == 441 * 32
        plain.attn_window_size = 12
        assert plain.preprocess(torch.zeros(1, 1, 1000)).shape[-1] == 441 * 24


def test_dac_and_snac_from_pretrained_local_directories(tmp_path):
    """``DAC.from_pretrained`` (dac.py:251-270) and ``SNAC.from_config / from_pretrained`` (snac.py:177-201) on LOCAL directories (``config.json`` =
    the constructor's keyword arguments + ``model.safetensors`` in the reference's parameter names): the loaded models reproduce the reference run's codes."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from safetensors.torch import save_file

    from mlx_audio_amd.codec.models.descript import DAC
    from mlx_audio_amd.codec.models.snac import SNAC

    fx = np.load(os.path.join(GOLD, "ref_dac_encode.npz"))
    c, w = dac_model_weights(fx)
    d = tmp_path / "dac"
    d.mkdir()
    json.dump(c, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in w.items()}, str(d / "model.safetensors"))
    fs = np.load(os.path.join(GOLD, "ref_snac_encode_dense.npz"))
    sc, sw = snac_model_weights(fs)
    s = tmp_path / "snac"
    s.mkdir()
    json.dump(sc, open(s / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in sw.items()}, str(s / "model.safetensors"))
    with _ops_emu.patched():
        dac = DAC.from_pretrained(str(d), device="cpu")
        _, codes, *_ = dac.encode(torch.from_numpy(fx["audio"]))
        assert np.array_equal(codes.numpy(), fx["codes"])
        snac = SNAC.from_pretrained(str(s), device="cpu")
        assert all(np.array_equal(a.numpy(), fs[f"codes{i}"]) for i, a in enumerate(snac.encode(torch.from_numpy(fs["audio"]))))
        fresh = SNAC.from_config(s / "config.json", device="cpu")       # random parameters, both halves
        assert fresh.enc is not None and fresh.quantizer.in_proj is not None and [tuple(x.shape) for x in fresh.encode(torch.zeros(1, 1, 500))] == [(1, 3), (1, 6), (1, 12)]   # 500 -> 576 samples = 12 frames of hop 48
    for cls in (DAC, SNAC):
        with pytest.raises(FileNotFoundError):
            cls.from_pretrained(str(tmp_path / "missing"), device="cpu")


def test_dac_compress_decompress_against_the_reference_run(tmp_path):
    """``CodecMixin.compress`` / ``decompress`` + ``DACFile`` (codec/models/descript/base.py:13-231) through the product's host code (emulated operators),
    against the reference's own run with a scripted audio reader (tests/golden/ref_dac_compress.npz): a signal of 2.3 windows (three chunks, the last one
    zero-padded; ``padding`` False inside, restored after) and one shorter than the window; what the reference's ``get_delay`` / ``get_output_length``
    return on this architecture (0 / identity: its module walk finds no ``nn.Conv1d``) is what the mirror assumes."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.descript import DAC, DACFile

    fx = np.load(os.path.join(GOLD, "ref_dac_compress.npz"))
    assert int(fx["delay"]) == 0 and int(fx["out_len_1000"]) == 1000 and bool(fx["padding_after"])
    c, w = dac_model_weights(fx)
    with _ops_emu.patched():
        eng = DAC(**c, weights=w, device="cpu")
        assert eng.delay == 0 and eng.get_output_length(1000) == 1000 and eng.padding is True
        for tag in ("long", "short"):
            sig = fx[f"{tag}_signal"]
            f = eng.compress((sig, c["sample_rate"]), win_duration=float(fx[f"{tag}_win"]))
            assert np.array_equal(f.codes.numpy(), fx[f"{tag}_codes"]) and f.chunk_length == int(fx[f"{tag}_chunk_length"]) and f.channels == 1
            assert abs(f.input_db - float(fx[f"{tag}_input_db"])) < 1e-4 and f.padding == bool(fx[f"{tag}_padding"]) and eng.padding is True
            assert abs(f.original_length - float(fx[f"{tag}_original_length"])) < 1e-12 and f.sample_rate == c["sample_rate"]
            rec = eng.decompress(f)
            assert tuple(rec.shape) == fx[f"{tag}_recons"].shape and rel_max(rec.numpy(), fx[f"{tag}_recons"]) < 2e-5
            path = f.save(tmp_path / f"{tag}_clip")                 # the suffix becomes .dac
            assert path.suffix == ".dac"
            g = DACFile.load(path)
            assert torch.equal(g.codes.long(), f.codes.long()) and g.chunk_length == f.chunk_length and g.padding == f.padding and g.dac_version == "1.0.0"
            assert rel_max(eng.decompress(str(path)).numpy(), fx[f"{tag}_recons"]) < 2e-5
        f2 = eng.compress((fx["long_signal"], c["sample_rate"]), win_duration=0.1, n_quantizers=2)
        assert np.array_equal(f2.codes.numpy(), fx["long_codes_nq2"])
        with pytest.raises(ValueError, match="does not match the sample rate"):
            eng.compress((fx["short_signal"], 8000))


def test_codec_decode_host_schedules_dry_run():
    """The DECODE schedules of DAC / SNAC / EnCodec (folded lookup tables, Snake / ELU prologues, polyphase transposed convs with their trims, depthwise convs,
    the noise blocks, the LSTM, the reflect paddings) over the emulated operators against the reference's own ``decode`` runs (tests/golden/ref_{dac,snac,
    encodec}_tiny.npz) -- the host logic of these rows on the CPU suite; the kernels themselves are held by tests/test_{dac,snac,encodec}_gpu.py."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ops_emu
    from mlx_audio_amd.codec.models.descript import DAC, make_dac_weights
    from mlx_audio_amd.codec.models.encodec import Encodec, make_encodec_weights
    from mlx_audio_amd.codec.models.snac import SNAC, make_snac_weights

    with _ops_emu.patched():
        fx = np.load(os.path.join(GOLD, "ref_dac_tiny.npz"))
        rates, dim, latent, nq, csize, cdim = [8, 5, 4, 2], 64, 32, 3, 128, 8
        dac = DAC(encoder_dim=16, encoder_rates=[2, 4, 5, 8], latent_dim=latent, decoder_dim=dim, decoder_rates=rates, n_codebooks=nq, codebook_size=csize, codebook_dim=cdim,
                  sample_rate=16000, weights=make_dac_weights(dim, rates, latent, nq, csize, cdim, seed=int(fx["seed_w"])), device="cpu")
        z, _, _ = dac.quantizer.from_codes(torch.from_numpy(fx["codes"]).long())
        assert rel_max(z.numpy(), fx["z"]) < 1e-5
        audio = dac.decode(z)
        assert audio.shape == fx["audio"].shape and rel_max(audio.numpy(), fx["audio"]) < 3e-5

        fs = np.load(os.path.join(GOLD, "ref_snac_tiny.npz"), allow_pickle=True)
        cfg = json.loads(str(fs["config"]))
        lat = cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
        snac = SNAC(**cfg, weights=make_snac_weights(lat, cfg["decoder_dim"], cfg["decoder_rates"], cfg["vq_strides"], cfg["codebook_size"], cfg["codebook_dim"], True, True,
                                                     seed=int(fs["seed_w"])), device="cpu")
        codes = [torch.from_numpy(fs[f"codes{i}"]).long() for i in range(len(cfg["vq_strides"]))]
        noises = [torch.from_numpy(fs[f"noise{i}"]) for i in range(int(fs["n_noise"]))]
        assert rel_max(snac.quantizer.from_codes(codes).numpy(), fs["z"]) < 1e-5
        a2 = snac.decode(codes, noises=noises)
        assert a2.shape == fs["audio"].shape and rel_max(a2.numpy(), fs["audio"]) < 3e-5

        fe = np.load(os.path.join(GOLD, "ref_encodec_tiny.npz"))
        ecfg = json.loads(str(fe["config"]))
        enc = Encodec(ecfg, weights=make_encodec_weights(ecfg, seed=int(fe["seed_w"])), device="cpu")
        ecodes = torch.from_numpy(fe["codes"]).long()
        assert rel_max(enc.quantizer.decode(ecodes[:, 0]).numpy(), fe["embeddings"]) < 1e-6   # (the emulation sums in float64; the kernel's float32 slot order is held bit-exact on the GPU)
        a3 = enc.decode(ecodes, [None])
        assert a3.shape == fe["audio"].shape and rel_max(a3.numpy(), fe["audio"]) < 5e-5
