"""Host-only helpers OUTSIDE the hot-path scope (SURVEY section 2.1 marks them out of scope): numpy code kept because a caller on the path's edge needs it
(``DACFile`` / ``compress`` normalise loudness before encoding; the reference exposes the same functions from ``mlx_audio.dsp``).  Nothing on the device
path imports this package; ``mlx_audio_amd.dsp`` resolves these names lazily."""
