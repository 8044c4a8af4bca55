"""Regression check on GENERATED code (no GPU): the latency-bound decode kernels converted in round 4 must stay free of the serial-load pattern.

hipcc compiles a guarded load (``k < K ? *p : 0``, ``ptr ? ptr[n] : c``) to a branch around the load and -- when the value is used or copied inside the
branch -- an ``s_waitcnt vmcnt(0)`` behind it: N guarded loads are N serial memory round trips ("Round 4, second pass").
``tools/scan_serial_waits.py`` compiles a source file to gfx950 assembly and counts, per kernel, the ``vmcnt(0)`` waits with at most one vector-memory
load since the previous wait.  The bars below are the measured counts of the converted kernels plus a small allowance (their remaining hits are the
mutually exclusive epilogue variants and the tail of a loop); the kernels they replaced counted 22 (gemv1_splitk), 21 (gemv1_res) and 118 (rows_finish)."""
import importlib.util
import os
import shutil
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc (cross-compiles without a GPU) and c++filt")


def _scan(name):
    spec = importlib.util.spec_from_file_location("scan_serial_waits", os.path.join(ROOT, "tools", "scan_serial_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as td:
        rows = mod.scan(os.path.join(ROOT, "mlx_audio_amd", "csrc", name), td)
    assert rows, f"{name}: no kernels found in the assembly"
    return {sym: (serial, waits0, loads) for serial, waits0, loads, sym in rows}


def _worst(rows, needle):
    hit = {k: v for k, v in rows.items() if needle in k}
    assert hit, f"no kernel matching {needle}"
    return max(v[0] for v in hit.values()), hit


def test_row_epilogue_is_straight_line():
    rows = _scan("rows_pipe.hip")
    worst, hit = _worst(rows, "rows_finish_lean_kernel")
    assert worst <= 5, hit                                   # measured 3 (round 4); rows_finish_kernel itself: > 100
    assert all(v[2] >= 20 for v in hit.values()), hit        # ... while still holding its >= 20 loads: the check is not vacuous
    old, _ = _worst(rows, "18rows_finish_kernel")
    assert old > 50                                          # the scanner still sees the pattern where it exists


def test_one_row_gemv_kernels_are_straight_line():
    rows = _scan("gemv.hip")
    worst, hit = _worst(rows, "19gemv1_splitk_kernel")
    assert worst <= 6, hit                                   # measured 2-3; the first form (gemv1_splitk_old_kernel): 18-22
    worst, hit = _worst(rows, "19gemv1_stream_kernel")
    assert worst <= 20, hit                                  # measured 7 (one column) / 13-16 (two columns: GLU, rotary and plain epilogues are exclusive paths)
    old, _ = _worst(rows, "gemv1_splitk_old_kernel")
    assert old >= 15
