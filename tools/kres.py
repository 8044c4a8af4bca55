#!/usr/bin/env python3
"""Compact per-kernel resource table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: tools/kres.py file.hip [extra hipcc flags]   (compiles for gfx950 into /tmp, prints name / VGPR / AGPR / SGPR / scratch / spills / LDS)"""
import re, subprocess, sys, os
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", "/tmp/kres_%d.o" % os.getpid(),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
p = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
rows, cur = [], None
for ln in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?): (\S+)\s+\[-Rpass", ln) or re.search(r"remark:\s+(Function Name): (\S+)", ln)
    if not m:
        if "error" in ln: print(ln)
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-5s %-5s %-5s %-8s %-7s %-7s %-6s %s" % ("VGPR", "AGPR", "SGPR", "scratch", "vspill", "sspill", "occ", "kernel"))
for r in rows:
    print("%-5s %-5s %-5s %-8s %-7s %-7s %-6s %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                               r.get("Occupancy [waves/SIMD]"), r["name"].replace("(anonymous namespace)::", "")[:110]))
sys.exit(p.returncode)
