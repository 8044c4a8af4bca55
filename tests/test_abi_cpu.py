"""CPU-side checks of the C-ABI boundary (no GPU, no compute launches):
the shared library loads, exports every symbol include/mi355audio.h declares, its struct layouts match what
the ctypes layer generates from the header, compute entry points fail loudly (never fall back) when there
is no device, and the host-side weight packers produce the documented fragment order."""
import ctypes
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355audio.h")


@pytest.fixture(scope="module")
def lib():
    from mlx_audio_amd import _lib

    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from mlx_audio_amd import _lib

    names = _lib.declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mi355_abi_version() == _lib.ABI_VERSION
    # and nothing torch-typed crosses the boundary: the header is plain C
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", HEADER], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_struct_layout_matches_c(tmp_path):
    from mlx_audio_amd import _lib

    structs = _lib._STRUCT_DECLS
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for name, fields in structs.items():
        src.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in fields:
            src.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    want = dict(line.split() for line in out.strip().splitlines())
    for name in structs:
        st = _lib.STRUCTS[name]
        assert ctypes.sizeof(st) == int(want[name]), name
        for f, _ in structs[name]:
            assert getattr(st, f).offset == int(want[f"{name}.{f}"]), (name, f)


def test_no_silent_cpu_fallback():
    """Without a ROCm device the product path must raise, not compute something somewhere else."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mlx_audio_amd import _lib, ops

    with pytest.raises(_lib.Mi355Error):
        ops.require_gpu()
    # argument validation happens before any launch and reports through mi355_last_error()
    lib = _lib.load()
    st = _lib.STRUCTS["mi355_conv_gemm_args"]()
    rc = lib.mi355_conv_gemm(ctypes.byref(st), None)
    assert rc == -1 and b"null tensor" in lib.mi355_last_error()
    rc = lib.mi355_conv_gemm(None, None)
    assert rc == -1
    # the product package never imports the oracle
    pkg = os.path.join(ROOT, "mlx_audio_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def _bf16(v):
    return torch.tensor(v, dtype=torch.float32).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def test_pack_conv_weight_fragment_order(lib):
    """element ((((chunk*K + tap)*NTp + nt)*2 + kk)*64 + lane)*8 + j  <-  w[n = nt*32 + lane%32, tap, c = chunk*32 + kk*16 + (lane//32)*8 + j]"""
    rng = np.random.default_rng(0)
    for cout, k, cin in [(5, 3, 7), (130, 2, 33), (128, 1, 64)]:
        w = rng.standard_normal((cout, k, cin)).astype(np.float32)
        n = lib.mi355_packed_conv_weight_elems(cout, k, cin)
        chunks, ntp = (cin + 31) // 32, ((cout + 127) // 128) * 4
        assert n == chunks * k * ntp * 2 * 512
        out = np.empty(n, dtype=np.uint16)
        assert lib.mi355_pack_conv_weight_host(w.ctypes.data, cout, k, cin, out.ctypes.data) == 0
        out = out.reshape(chunks, k, ntp, 2, 64, 8)
        wb = _bf16(w)
        for _ in range(200):
            ch, tap, nt, kk, lane, j = (int(rng.integers(0, m)) for m in (chunks, k, ntp, 2, 64, 8))
            nn_, c = nt * 32 + lane % 32, ch * 32 + kk * 16 + (lane // 32) * 8 + j
            want = wb[nn_, tap, c] if (nn_ < cout and c < cin) else 0
            assert out[ch, tap, nt, kk, lane, j] == want


def test_polyphase_repack_equals_conv_transpose():
    """ops.pack_conv_transpose's stride-1 polyphase form reproduces conv_transpose1d (istftnet.py:128-170 path)."""
    from mlx_audio_amd.ops import _polyphase_weight

    g = torch.Generator().manual_seed(1)
    for cin, cout, k, s, L in [(6, 4, 20, 10, 9), (3, 5, 12, 6, 7), (2, 2, 4, 2, 5)]:
        p = (k - s) // 2
        w_t = torch.randn(cout, k, cin, generator=g, dtype=torch.float64)  # mx.conv_transpose1d layout
        x = torch.randn(1, L, cin, generator=g, dtype=torch.float64)
        ref = F.conv_transpose1d(x.transpose(1, 2), w_t.permute(2, 0, 1), stride=s, padding=p).transpose(1, 2)[0]
        w = _polyphase_weight(w_t, s)  # [s*cout, k/s, cin]
        kp = k // s
        y = F.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), padding=kp - 1).transpose(1, 2)[0]  # rows u = 0 .. L+kp-2
        lout = (L - 1) * s - 2 * p + k
        got = torch.zeros(lout, cout, dtype=torch.float64)
        for u in range(y.shape[0]):
            for r in range(s):
                n = u * s + r - p
                if 0 <= n < lout:
                    got[n] = y[u, r * cout:(r + 1) * cout]
        assert torch.allclose(got, ref, atol=1e-12)


def test_lstm_weight_packing(lib):
    H = 16
    rng = np.random.default_rng(3)
    wf, wb = rng.standard_normal((4 * H, H)).astype(np.float32), rng.standard_normal((4 * H, H)).astype(np.float32)
    out = np.empty(2 * 4 * H * H, dtype=np.uint16)
    assert lib.mi355_pack_lstm_wh_host(wf.ctypes.data, wb.ctypes.data, H, out.ctypes.data) == 0
    out = out.reshape(2, H // 8, 4 * H, 8)
    for d, w in enumerate((wf, wb)):
        want = _bf16(w).reshape(4 * H, H // 8, 8).transpose(1, 0, 2)
        assert np.array_equal(out[d], want)
    assert lib.mi355_pack_lstm_wh_host(wf.ctypes.data, wb.ctypes.data, 12, out.ctypes.data) == -1  # H % 8 != 0


def test_pack_conv_weight_fp16_rounding(lib):
    """MI355_W_F16 packing: IEEE round-to-nearest-even incl. subnormals (== torch's conversion), saturating at 65504."""
    rng = np.random.default_rng(7)
    cout, k, cin = 32, 1, 32
    mags = np.concatenate([rng.standard_normal(512), rng.standard_normal(256) * 1e-5, rng.standard_normal(128) * 1e-7,
                           np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25,
                                     6.1035e-5, 6.0976e-5]), rng.standard_normal(116) * 3e3]).astype(np.float32)
    w = mags.reshape(cout, k, cin)
    out = np.empty(lib.mi355_packed_conv_weight_elems(cout, k, cin), dtype=np.uint16)
    assert lib.mi355_pack_conv_weight_host_dt(w.ctypes.data, cout, k, cin, 1, out.ctypes.data) == 0
    out = out.reshape(1, 1, 4, 2, 64, 8)
    want = torch.from_numpy(np.clip(w, -65504.0, 65504.0)).to(torch.float16).view(torch.int16).numpy().view(np.uint16)
    # values in (65504, 65520) round DOWN to 65504 in IEEE too; >= 65520 would be inf: we saturate instead
    for n in range(cout):
        for c in range(cin):
            nt, lane_lo, kk, hi, j = n // 32, n % 32, c // 16, (c % 16) // 8, c % 8
            assert out[0, 0, nt, kk, hi * 32 + lane_lo, j] == want[n, 0, c], (w[n, 0, c], hex(out[0, 0, nt, kk, hi * 32 + lane_lo, j]), hex(want[n, 0, c]))
    assert lib.mi355_pack_conv_weight_host_dt(w.ctypes.data, cout, k, cin, 7, out.ctypes.data) == -1


def test_fp8_row_packer_matches_the_restated_format(lib):
    """mi355_pack_rowmajor_fp8_host against the oracle's restatement (torch.float8_e4m3fn, power-of-two row scales): bit-exact codes and
    scales, including ties, subnormals, the largest code, an all-zero row; the dequantised values are exactly representable in bf16."""
    from mlx_audio_amd import ops
    from oracle.lm_ref import dequantize_rows_fp8_ref, quantize_rows_fp8_ref

    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 256, generator=g) * 0.02
    w[3] = 0.0
    w[5, 7] = 1.5
    w[6] = torch.linspace(-448.0, 448.0, 256)                     # scale exactly 1: the row walks the whole code range incl. ties
    w[7, :16] = torch.tensor([2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0, -2.0 ** -9, 17.0, 18.0, 19.0, 20.0, 22.0, 26.0, 30.0, 36.0, 44.0, 448.0])
    codes, scale = ops.quantize_rows_fp8(w)
    rc, rs = quantize_rows_fp8_ref(w)
    assert torch.equal(scale, rs)
    assert torch.equal(codes, rc)
    assert float(scale[3]) == 1.0 and int(codes[3].max()) == 0
    assert float(scale[6]) == 1.0 and int(codes[6, -1]) == 0x7E and int(codes[6, 0]) == 0xFE
    dq = ops.dequantize_rows_fp8(codes, scale)
    assert torch.equal(dq, dequantize_rows_fp8_ref(rc, rs))
    assert torch.equal(dq.to(torch.bfloat16).to(torch.float32), dq)   # prefill (bf16 MFMA image) and decode (fp8 GEMV image) see the same weights
    rel = float((dq - w)[:3].norm() / w[:3].norm())
    assert rel < 0.04, rel
    # a fine grid across every binade of the format
    x = torch.cat([torch.linspace(-448, 448, 100001), torch.linspace(-0.05, 0.05, 100001)])
    x = torch.cat([x, torch.zeros((-x.numel() - 1) % 16), torch.tensor([448.0])]).reshape(1, -1)
    c2, s2 = ops.quantize_rows_fp8(x)
    assert float(s2[0]) == 1.0 and torch.equal(c2, x.to(torch.float8_e4m3fn).view(torch.uint8))


def test_fp8_device_decode_identity():
    """The kernels decode an e4m3fn byte by moving it into binary16 position (csrc/common.h cvt_w16<MI355_W_FP8>) and folding 2^8 into the row
    scale; the same integer recipe in numpy reproduces every finite code of the format exactly."""
    b = np.arange(256, dtype=np.uint32)
    x = b << 8
    h = (x & 0x8000) | ((x & 0x7F00) >> 1)
    dec = h.astype(np.uint16).view(np.float16).astype(np.float32) * 256.0
    ref = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    finite = np.isfinite(ref)
    assert finite.sum() == 254 and np.array_equal(dec[finite], ref[finite])
