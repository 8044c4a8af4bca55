#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db on this ROCm build).

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db [passes] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# total kernel time {total / 1e6:.3f} ms over {passes} pass(es) = {total / 1e6 / passes:.3f} ms / pass")
    print("total_ms calls avg_us min_us max_us pct kernel")
    for name, n, tot, mn, mx in rows:
        print(f"{tot / 1e6:.3f} {n} {tot / n / 1e3:.1f} {mn / 1e3:.1f} {mx / 1e3:.1f} {100.0 * tot / total:.1f} {name}")


if __name__ == "__main__":
    main()
