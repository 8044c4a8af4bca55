#!/usr/bin/env python
"""Sum of one PMC counter over ALL dispatches of a rocprofv3 run (rocpd database), overall and per kernel name prefix.

    python tools/pmc_total.py <results.db> FETCH_SIZE [divide-by]

For FETCH_SIZE / WRITE_SIZE (KB) the gfx950-corrected HBM bytes are printed too (2 x FETCH_SIZE, 1 x WRITE_SIZE; MI355X_MICROARCH.md).
"""
import collections
import sys

from pmc_traffic import per_kernel


def main():
    per = per_kernel(sys.argv[1], sys.argv[2])
    div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    scale = {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0}.get(sys.argv[2])
    tot = sum(sum(v) for v in per.values())
    print(f"{sys.argv[2]}: total {tot:.1f} over {sum(len(v) for v in per.values())} dispatches" + (f" = {tot * scale / 1e9:.3f} GB corrected = {tot * scale / 1e9 / div:.3f} GB per unit ({div:g} units)" if scale else ""))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for k, v in per.items():
        key = k.split("(")[0][-70:]
        agg[key][0] += sum(v)
        agg[key][1] += len(v)
    for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {s * (scale or 1.0) / (1e9 if scale else 1.0):10.3f} {'GB' if scale else ''} {n:7d} launches  {k}")


if __name__ == "__main__":
    main()
