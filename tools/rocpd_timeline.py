#!/usr/bin/env python
"""One period of a dependent launch chain from a rocprofv3 rocpd database: the dispatches between two consecutive occurrences of a marker kernel, in start
order, each with its duration and the idle gap in front of it -- says whether a decode frame is made of kernel time or of gaps, and what it is made of.

    python tools/rocpd_timeline.py DB marker-substring [occurrence-from-the-end=3] [--span=N] [--list] > profiles/<name>.txt

--from-start=I: the period starts at the I-th marker occurrence counted from the START of the trace (default: counted back from the end).
--span=N: the marker fires N times per period (Qwen3-TTS: 16 sampling launches per frame, CSM: 32): the period is N marker occurrences long.
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    marker = args[1]
    back = int(args[2]) if len(args) > 2 else 3
    span = max([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--span=")] + [1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcol = [c for c in cols if "grid" in c.lower() and c.lower().endswith("x")]
    gsel = gcol[0] if gcol else "0"
    rows = cur.execute(f"select {name_col}, {gsel}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < back + span:
        print(f"# only {len(marks)} launches match {marker!r}")
        return
    # the marker may fire several times per period (e.g. once per code group): a period = the span between marker occurrences `back * per` apart is
    # left to the caller -- here: from the marker `back + 1` from the end to the one `back` from the end
    first = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--from-start=")]
    if first and first[0] + span < len(marks):
        a, b = marks[first[0]], marks[first[0] + span]
    else:
        a, b = marks[-back - span], marks[-back]
    seg = rows[a + 1:b + 1]
    t0 = rows[a][3]
    busy = sum(r[3] - r[2] for r in seg)
    span = seg[-1][3] - t0
    print(f"# period between launches #{a} and #{b} of {len(rows)}: {len(seg)} launches, span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us ({100.0 * busy / span:.1f} %), "
          f"gaps {(span - busy) / 1e3:.1f} us = {(span - busy) / 1e3 / len(seg):.2f} us per launch")
    agg = collections.OrderedDict()
    prev = t0
    for name, grid, s, e in seg:
        k = (short(name), grid)
        d = agg.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        d[2] += max(s - prev, 0) / 1e3
        prev = e
    print("launches total_us avg_us gap_before_avg_us kernel grid")
    for (n, g), (c, t, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:5d} {t:9.1f} {t / c:7.2f} {gp / c:7.2f} {n} grid={g}")
    if "--list" in sys.argv:
        prev = t0
        for name, grid, s, e in seg:
            print(f"  +{(s - t0) / 1e3:9.1f} us  gap {max(s - prev, 0) / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  {short(name)} grid={grid}")
            prev = e


if __name__ == "__main__":
    main()
