"""Guard on GENERATED code (no GPU): no packed-fp32 VALU arithmetic in any conv kernel.

Round 5 found that hipcc's SLP vectoriser packs the fused instance-norm statistics of ``conv_epilogue_interior`` (csrc/conv_common.h) into
``v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32`` and that this packed code returns RANDOM statistics on gfx950 for launches of >= 128 tiles (identical
inputs, stored outputs correct; ``profiles/r5_diag_conv_stats_variants_call5.txt``, ``profiles/r6_slp_hazard_*.txt``).  The library is therefore built
with ``-fno-slp-vectorize`` (``mlx_audio_amd/build.py: COMMON_FLAGS``).  This test compiles every conv translation unit with the PRODUCTION flags and
fails if a packed-fp32 arithmetic instruction appears in a ``conv_ws4_kernel`` / ``conv_gemm_kernel`` code object -- a dropped flag, a new ROCm whose
other passes pack, or an intrinsic somebody adds to the epilogue would bring the corruption back silently; and it compiles ONE unit without the flag to
show the scan still sees the pattern where it exists."""
import concurrent.futures
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "mlx_audio_amd", "csrc")
CONV_UNITS = ["conv_gemm.hip", "conv_ws4.hip", "conv_ws4_p4.hip", "conv_ws4_p13.hip", "conv_ws4_p5.hip", "conv_ws4_fq.hip"]
PACKED = re.compile(r"^\s*(v_pk_(?:add|mul|fma)_f32)\b")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")


def _packed_ops(unit, flags, td):
    """{kernel symbol: count of packed-fp32 arithmetic instructions} for the conv kernels of one translation unit."""
    out = os.path.join(td, unit + ("." + "_".join(f.strip("-") for f in flags) if flags else ".plain") + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, unit), "-o", out],
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-600:]
    counts, name = {}, None
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1) if ("conv_ws4_kernel" in m.group(1) or "conv_gemm_kernel" in m.group(1)) else None
            if name:
                counts[name] = 0
            continue
        if name and PACKED.match(ln):
            counts[name] += 1
        if "s_endpgm" in ln:
            name = None
    return counts


def test_no_packed_fp32_arithmetic_in_the_conv_kernels():
    from mlx_audio_amd import build as B

    assert "-fno-slp-vectorize" in B.COMMON_FLAGS, "the production build no longer switches the SLP vectoriser off (mlx_audio_amd/build.py)"
    assert all(u in B.SOURCES for u in CONV_UNITS)
    with tempfile.TemporaryDirectory() as td:
        with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
            prod = {u: ex.submit(_packed_ops, u, list(B.COMMON_FLAGS) + list(B.EXTRA_FLAGS.get(u, [])), td) for u in CONV_UNITS}
            plain = ex.submit(_packed_ops, "conv_ws4_p13.hip", [], td)   # the smallest unit, WITHOUT the flag: the scan must see packed code there
            prod = {u: f.result() for u, f in prod.items()}
            plain = plain.result()
    kernels = sum(len(c) for c in prod.values())
    assert kernels >= 30, kernels                                  # the scan found the instantiations (not vacuous)
    bad = {u: {k: n for k, n in c.items() if n} for u, c in prod.items()}
    bad = {u: c for u, c in bad.items() if c}
    assert not bad, f"packed fp32 arithmetic in conv kernels under the production flags: {bad}"
    assert sum(plain.values()) > 0, "the SLP build of conv_ws4_p13.hip shows no packed fp32 code: the scan (or the compiler) changed -- re-derive this guard"
