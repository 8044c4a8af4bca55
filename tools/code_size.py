#!/usr/bin/env python
"""Static code size / registers / scratch of every conv kernel instantiation in the built library objects (CPU only: reads the gfx950 code
objects embedded in mlx_audio_amd/lib/obj/*.o).

    python tools/code_size.py > profiles/r2_static_code_size_conv.txt
"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def main():
    print("# kernel<template args>  code bytes  vgprs  sgprs  scratch bytes/lane  vgpr spills   (conv_ws4_kernel<PREC, PRE, EPI, GEMM, DBG, ABL>:")
    print("#   PRE 0 none 1 LeakyReLU 2 Snake 3 SnakeBeta 4 ELU; EPI 0 none/LeakyReLU 1 GELU 2 SiLU/GELU-tanh/ELU/tanh; conv_gemm_kernel<BM, BN, PREC, VEC>)")
    for o in sorted(glob.glob(os.path.join(ROOT, "mlx_audio_amd", "lib", "obj", "conv_*.o"))):
        tmp = tempfile.mkdtemp()
        try:
            shutil.copy(o, os.path.join(tmp, "x.o"))
            subprocess.run([f"{BIN}/llvm-objdump", "--offloading", "x.o"], cwd=tmp, capture_output=True)
            cos = [f for f in os.listdir(tmp) if f.endswith("gfx950")]
            if not cos:
                print(f"# {os.path.basename(o)}: no gfx950 code object")
                continue
            co = os.path.join(tmp, cos[0])
            sizes = {}
            for line in subprocess.run(["nm", "--print-size", co], capture_output=True, text=True).stdout.splitlines():
                f = line.split()
                if len(f) == 4 and f[2] in "Tt" and "conv" in f[3]:
                    sizes[f[3]] = int(f[1], 16)
            notes = subprocess.run([f"{BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            rec = {}
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "name":
                    cur = rec.setdefault(v, {})
                elif k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count"):
                    cur[k] = v
            print(f"# {os.path.basename(o)}")
            for name, size in sorted(sizes.items(), key=lambda kv: -kv[1]):
                r = rec.get(name, {})
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(mi355_conv_gemm_args.*", "", dem).replace("void ", "").replace("(anonymous namespace)::", "").replace("mi355conv::", "")
                print(f"{dem:60s} {size:7d} {r.get('vgpr_count', '?'):>4s} {r.get('sgpr_count', '?'):>4s} {r.get('private_segment_fixed_size', '?'):>5s} {r.get('vgpr_spill_count', '?'):>4s}")
        finally:
            shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
