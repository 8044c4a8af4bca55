"""Builds libmi355audio.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

    python -m mlx_audio_amd.build          # or: from mlx_audio_amd.build import build; build()

The shared library is written to ``mlx_audio_amd/lib/libmi355audio.so`` so that it travels with the
source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles for
gfx950 without a GPU, so this also runs in the CPU-only authoring container.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmi355audio.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "mi355audio.h")
SOURCES = ["api.cpp", "conv_gemm.hip", "conv_ws4.hip", "conv_ws4_p4.hip", "conv_ws4_p13.hip", "conv_ws4_p5.hip", "conv_ws4_p6.hip", "conv_ws4_p5_probe.hip", "conv_ws4_fq.hip", "conv_quant.hip", "norm.hip", "lstm.hip", "lstm_seq.hip", "attention.hip", "glue.hip", "source.hip", "fft.hip", "flash_attn.hip",
           "decode_rules.hip", "gemv.hip", "gemv_mfma.hip", "gemv_mfma_fp8.hip", "gemm_rows.hip", "rows_pipe.hip", "transformer.hip", "rvq.hip", "ecapa.hip", "sampler.hip", "stack_step.cpp"]
# per-file extra flags: source.hip mirrors the reference's fp32 op order one rounding at a time
EXTRA_FLAGS = {"source.hip": ["-ffp-contract=off"]}
# Every translation unit: no SLP vectorisation.  Under plain -O3 hipcc packs adjacent scalar fp32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32 /
# v_pk_fma_f32 with SGPR-pair operands; in the interior epilogue of conv_ws4_kernel that packed code produced RANDOM errors in the fused
# instance-norm statistics (the M2 term of a few 32-column fragments per launch, different ones every run; identical inputs, stored outputs
# correct): round 5, tools/diag_conv_stats.py, profiles/r5_diag_conv_stats_variants_call5.txt -- the same source compiled with
# -fno-slp-vectorize is exact on every run.  (MI355X_MICROARCH.md also prices packed fp32 beside MFMAs as slower than the scalar pair.)
COMMON_FLAGS = ["-fno-slp-vectorize"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _fingerprint() -> str:
    h = hashlib.sha256()
    for name in SOURCES + ["common.h", "conv_common.h", "conv_ws4.h", "fft_fast.h", "fft_tw.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    with open(HEADER, "rb") as f:
        h.update(f.read())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    h.update(repr(COMMON_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libmi355audio.stamp")
    fp = _fingerprint()
    if not force and os.path.exists(LIBPATH) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIBPATH
    hipcc = _hipcc()
    objs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *COMMON_FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs
    if verbose:
        print("[build]", " ".join(link), flush=True)
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(stamp, "w") as f:
        f.write(fp)
    return LIBPATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
