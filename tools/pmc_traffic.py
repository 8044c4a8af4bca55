#!/usr/bin/env python
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: 3 + 2 of the 4 TCC slots).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d A -o f -- <cmd>;  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d B -o w -- <cmd>
    python tools/pmc_traffic.py A/.../f_results.db B/.../w_results.db [kernel-substring] > profiles/<name>.json

Units and gfx950 correction (MI355X_MICROARCH.md section HBM, cdna_hip_programming.md section 7): both counters are in KB; FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read on this rocprofv3 build, so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Values are summed over the counter's hardware instances per dispatch, then averaged over the dispatches of each kernel.
"""
import collections
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    kcol = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in ix else [c for c in cols if "value" in c][0]
    dcol = "dispatch_id"
    per = collections.defaultdict(float)
    for r in cur.execute("select * from counters_collection"):
        if r[ix[ccol]] != counter:
            continue
        per[(r[ix[kcol]], r[ix[dcol]])] += float(r[ix[vcol]])
    out = collections.defaultdict(list)
    for (k, _), v in per.items():
        out[k].append(v)
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    sub = sys.argv[3] if len(sys.argv) > 3 else ""
    res = {}
    for k in sorted(set(f) | set(w)):
        if sub and sub not in k:
            continue
        fk, wk = f.get(k, [0.0]), w.get(k, [0.0])
        fetch_kb, write_kb = sum(fk) / len(fk), sum(wk) / len(wk)
        res[k[:120]] = {"dispatches": max(len(fk), len(wk)), "FETCH_SIZE_KB_avg": fetch_kb, "WRITE_SIZE_KB_avg": write_kb,
                        "hbm_bytes_per_launch_corrected": (2.0 * fetch_kb + write_kb) * 1024.0,
                        "hbm_bytes_total_corrected": (2.0 * sum(fk) + sum(wk)) * 1024.0}
    print(json.dumps({"correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2 of a coalesced stream)", "kernels": res}, indent=1))


if __name__ == "__main__":
    main()
