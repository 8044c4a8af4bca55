#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db on this ROCm build).

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db [passes] [--by-grid] > profiles/<name>.txt

--by-grid additionally groups by the launch's grid size (one line per kernel and shape: the per-shape duration of the decode-step GEMVs).
"""
import sqlite3
import sys


def main():
    by_grid = "--by-grid" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    passes = int(args[1]) if len(args) > 1 else 1
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in cols if "grid" in c.lower() and c.lower().endswith("x")] if by_grid else []
    if by_grid and gcols:
        key = f"{name_col} || ' grid=' || {gcols[0]}"
    else:
        key = name_col
    rows = cur.execute(f"select {key}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by 1 order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# total kernel time {total / 1e6:.3f} ms over {passes} pass(es) = {total / 1e6 / passes:.3f} ms / pass")
    print("total_ms calls avg_us min_us max_us pct kernel")
    for name, n, tot, mn, mx in rows:
        print(f"{tot / 1e6:.3f} {n} {tot / n / 1e3:.1f} {mn / 1e3:.1f} {mx / 1e3:.1f} {100.0 * tot / total:.1f} {name}")


if __name__ == "__main__":
    main()
