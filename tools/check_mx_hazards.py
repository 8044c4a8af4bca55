#!/usr/bin/env python3
"""Static hazard audit of the inline-asm MFMAs of conv precision 5 (csrc/conv_ws4.h: v_mfma_scale_f32_32x32x64_f8f6f4 with its accumulator tied in
place).  hipcc pads no hazard whose producer or consumer sits inside an asm statement, so the three rules the kernel relies on are checked on the
generated code of every conv_ws4_kernel<5, ...> instantiation:

  1. D == C for every scaled MFMA (the tie the asm constraint asks for);
  2. within the 19 wait states behind a scaled MFMA (16 passes), on every control-flow path, nothing touches its D registers except another MFMA
     that takes exactly D as its C (accumulate chain: 0 wait states) -- no VALU / VMEM / LDS / scratch access, no partial overlap;
  3. no VALU instruction writes an operand register (A, B, scale A, scale B) of a scaled MFMA in the two issue slots in front of it, unless an
     s_nop >= 1 stands between them.

    python tools/check_mx_hazards.py            # compiles csrc/conv_ws4_p5.hip and conv_ws4_p6.hip to assembly and audits it; exit code 1 on a finding
Every instruction is counted as ONE wait state (an MFMA or a memory instruction occupies more), s_nop N as N + 1: conservative.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "mlx_audio_amd", "csrc", f) for f in ("conv_ws4_p5.hip", "conv_ws4_p6.hip")]   # e4m3 and FP4 lo passes
FLAGS = ["-fno-slp-vectorize"]   # the production flags (mlx_audio_amd/build.py: COMMON_FLAGS)
NEED = 19


def regs(tok):
    """'v[18:33]' / 'v4' / 'a[0:3]' -> set of (file, index); anything else -> empty."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def parse(path):
    kernels, cur, name = {}, None, None
    for ln in open(path):
        m = re.match(r"^(_ZN9mi355conv15conv_ws4_kernelILi[56]E\w+):", ln)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        cur.append(s)
        if s.startswith("s_endpgm"):
            pass
    return kernels


def audit(name, ins):
    labels = {s[:-1]: i for i, s in enumerate(ins) if s.endswith(":")}
    findings = []

    def ops_of(s):
        parts = s.split(None, 1)
        op = parts[0]
        toks = [t.strip() for t in re.split(r",", parts[1])] if len(parts) > 1 else []
        toks = [t.split()[0] for t in toks if t]
        return op, toks

    for i, s in enumerate(ins):
        if not s.startswith("v_mfma_scale"):
            continue
        op, t = ops_of(s)
        D, A, B, C, SA, SB = regs(t[0]), regs(t[1]), regs(t[2]), regs(t[3]), regs(t[4]), regs(t[5])
        if D != C:
            findings.append(f"{name}: line {i}: D != C: {s}")
        # rule 3: look back two issue slots
        back, j, pad = 0, i - 1, False
        while j >= 0 and back < 2:
            p = ins[j]
            if p.endswith(":"):
                j -= 1
                continue
            pop, pt = ops_of(p)
            if pop == "s_nop" and int(pt[0]) >= 1:
                pad = True
                break
            if pop.startswith("v_") and not pop.startswith("v_mfma") and pt and (regs(pt[0]) & (A | B | SA | SB)):
                findings.append(f"{name}: line {i}: VALU write of an operand {back + 1} slot(s) ahead without a pad: {p}  ->  {s}")
            if not pop.startswith("s_waitcnt"):
                back += 1
            j -= 1
        # rule 2: walk forward on every path
        seen, work = set(), [(i + 1, 0)]
        while work:
            k, st = work.pop()
            while k < len(ins) and st < NEED:
                if (k, st) in seen:
                    break
                seen.add((k, st))
                q = ins[k]
                if q.endswith(":"):
                    k += 1
                    continue
                qop, qt = ops_of(q)
                touched = set()
                for tok in qt:
                    touched |= regs(tok)
                if touched & D:
                    ok = qop.startswith("v_mfma") and len(qt) >= 4 and regs(qt[3]) == D and regs(qt[0]) == D
                    if ok:
                        break   # the chain continues inside the matrix pipe: that MFMA's own D is audited from its own position (asm) or by hipcc
                    findings.append(f"{name}: line {k}: {st} wait state(s) behind line {i} ({s.split()[0]} D={t[0]}): {q}")
                    break
                if qop == "s_nop":
                    st += int(qt[0]) + 1
                else:
                    st += 1
                if qop == "s_endpgm":
                    break
                if qop == "s_branch":
                    k = labels[qt[0]]
                    continue
                if qop.startswith("s_cbranch"):
                    work.append((labels[qt[0]], st))
                k += 1
    return findings


def main():
    kernels = {}
    with tempfile.TemporaryDirectory() as td:
        for i, src in enumerate(SRCS):
            out = os.path.join(td, f"p{i}.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *FLAGS, "-x", "hip", "-S", "--cuda-device-only", src, "-o", out],
                           check=True, stderr=subprocess.DEVNULL)
            kernels.update(parse(out))
    assert any("ILi5E" in k for k in kernels) and any("ILi6E" in k for k in kernels), "no conv_ws4_kernel<5 / 6, ...> in the assembly"
    bad = []
    for name, ins in kernels.items():
        n = sum(1 for s in ins if s.startswith("v_mfma_scale"))
        f = audit(name, ins)
        print(f"{name[:60]}...: {n} scaled MFMAs, {len(f)} finding(s)")
        bad += f
    for f in bad:
        print("  " + f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
