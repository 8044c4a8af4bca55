"""CPU oracle for the mlx-audio hot path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the reported CPU baseline -- never as the thing that
is measured or shipped.  ``mlx_audio_amd`` must not import this package.

The reference (``/root/reference``, Blaizzy/mlx-audio v0.5.0) is 100 % Python on
top of the external ``mlx`` runtime (pinned ``mlx 0.31.2``, ``uv.lock:930-931``),
which is not installed in this image and is not vendored in the reference tree.
The oracle therefore *restates* the reference's algorithms in numpy (dsp,
fp64/fp32) and PyTorch-CPU fp32 (model blocks), each function citing the
reference file:line it follows.

Pinning status
--------------
* ``dsp_ref`` (windows / stft / istft / mel_filters / ISTFTCache / mel front
  ends), ``interp_ref`` and the weight-normed transposed convolution are pinned
  against every known-answer vector the reference's own tests hold for this
  path (``tests/golden/*.json``, transcribed with file:line citations; checked
  in ``tests/test_oracle_golden.py``).
* every model oracle -- ``kokoro_ref``, ``kitten_ref``, ``whisper_ref``, ``qwen3_talker_ref``,
  ``qwen3_codec_ref``, ``csm_ref``, ``mimi_ref``, ``dac_ref``, ``snac_ref``, ``vocos_ref`` (and through
  them ``lm_ref.StackRef``'s Qwen3 / Llama-3 / Mimi variants) -- is **pinned to the reference's own modules**
  since round 2.  The reference holds no golden audio / token / logit fixture for
  them and real MLX cannot be installed, so ``tests/golden/make_reference_fixtures.py``
  imports the reference's source files from ``/root/reference`` (unmodified) over a
  numpy stand-in for the MLX array library (``tests/golden/mlx_shim.py``), runs them
  on seeded synthetic checkpoints and commits what they compute
  (``tests/golden/ref_*.npz``); ``tests/test_reference_fixtures_cpu.py`` holds the
  oracles to those files (1e-6 .. 3e-5; integer paths exact).  The stand-in is itself
  checked against the reference's known-answer vectors before anything is written,
  and its ``load_weights`` reports parameter names the reference's modules own but
  the synthetic checkpoints lack (none on the decode paths).  Not covered: MLX's own
  kernels.  Two reference quirks the restatements had missed were found this way and
  are now reproduced: KittenTTS's 2F + 1-point coarse phase grid and SNAC's
  per-channel NoiseBlock noise.
"""
