"""The inline-asm MFMAs of conv precision 5 carry their own hazard discipline (hipcc pads nothing around an asm statement): audit the generated
code of every conv_ws4_kernel<5, ...> instantiation (tools/check_mx_hazards.py: accumulator tied in place, nothing but an accumulate-chain MFMA
touches D within 19 wait states on any path, no VALU-written operand within two slots)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_precision5_asm_mfma_hazards():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mx_hazards.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "0 finding(s)" in r.stdout and "scaled MFMAs" in r.stdout
