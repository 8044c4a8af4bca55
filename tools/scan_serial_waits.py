#!/usr/bin/env python3
"""Finds kernels whose generated code awaits its global loads one at a time.

hipcc turns `cond ? *p : c` (and every load under a lane- or pointer-conditional branch whose other side supplies a value) into a branch around the
load, and -- when the loaded value is copied or used inside the branch -- an `s_waitcnt vmcnt(0)` right behind it: N guarded loads become N SERIAL
memory round trips (round 4: 8 of them in front of the first multiply of gemv1_splitk_kernel, 8 around the norm weights of gemv1_res_kernel).
This tool compiles the given .hip files to gfx950 assembly and reports, per kernel, the loads, the vmcnt(0) waits, and the "serial" ones: a
vmcnt(0) with at most ONE vector-memory load issued since the previous vmcnt wait.

    python tools/scan_serial_waits.py [files...]      # default: every csrc/*.hip; prints kernels with >= 3 serial waits
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path, td):
    out = os.path.join(td, os.path.basename(path) + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-S", "--cuda-device-only", path, "-o", out],
                       stderr=subprocess.PIPE, text=True)
    if r.returncode:
        print(path, "FAILED", r.stderr[-400:])
        return []
    rows, name, loads, since, serial, waits0 = [], None, 0, 0, 0, 0
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, loads, since, serial, waits0 = m.group(1), 0, 0, 0, 0
            continue
        if name is None:
            continue
        s = ln.split(";")[0].strip()
        if s.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            loads += 1
            since += 1
        elif s.startswith("s_waitcnt") and "vmcnt" in s:
            if "vmcnt(0)" in s:
                waits0 += 1
                if since <= 1 and loads > 0:
                    serial += 1
            since = 0
        elif s.startswith("s_endpgm"):
            rows.append((serial, waits0, loads, name))
            name = None
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "mlx_audio_amd", "csrc", "*.hip")))
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            rows = [r for r in scan(f, td) if r[0] >= 3]
            if not rows:
                continue
            print("==", os.path.basename(f))
            for serial, waits0, loads, name in sorted(rows, reverse=True):
                dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip().replace("(anonymous namespace)::", "")
                print("  serial %3d  vmcnt(0) %3d  loads %4d  %s" % (serial, waits0, loads, dem[:130]))


if __name__ == "__main__":
    main()
