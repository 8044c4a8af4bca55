#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --pmc ...).

    python tools/rocpd_pmc.py gpurun_out/pmc_sq/pmc_results.db [name-substring]
"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("# columns:", cols, file=sys.stderr)
    rows = cur.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    kcol = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in ix else [c for c in cols if "value" in c][0]
    dcol = "dispatch_id" if "dispatch_id" in ix else None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = {}
    per_disp = collections.defaultdict(float)
    for r in rows:
        k = r[ix[kcol]]
        if sub and sub not in k:
            continue
        key = (k[:90], r[ix[dcol]] if dcol else 0, r[ix[ccol]])
        per_disp[key] += float(r[ix[vcol]])  # sum over dimensions (XCC / SE instances)
        if "start" in ix and "end" in ix:
            dur[(k[:90], r[ix[dcol]] if dcol else 0)] = r[ix["end"]] - r[ix["start"]]
    for (k, d, c), v in per_disp.items():
        agg[k][c].append(v)
    kd = collections.defaultdict(list)
    for (k, d), t in dur.items():
        kd[k].append(t)
    for k in agg:
        n = max(len(v) for v in agg[k].values())
        t = sum(kd[k]) / len(kd[k]) / 1e3 if kd[k] else float("nan")
        print(f"{k}  dispatches={n} avg_us={t:.1f}")
        for c, v in sorted(agg[k].items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
