#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`,
ROCm 7.2 writes NAME_results.db) into the per-kernel table `--stats` would print as CSV:
name, calls, total ms, % of GPU kernel time, avg / min / max us.

    python tools/rocpd_stats.py gpurun_out/x/prof/x_results.db [--steps N] > profiles/x_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    steps = None
    if "--steps" in sys.argv:
        steps = float(sys.argv[sys.argv.index("--steps") + 1])
    db = sqlite3.connect(path)
    rows = list(db.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
        "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# total GPU kernel time {tot:.3f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f"; {steps:g} sampler steps in the trace -> {tot / steps:.3f} ms/step" if steps else ""))
    hdr = f"{'kernel':150s} {'calls':>7s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s}"
    if steps:
        hdr += f" {'ms/step':>9s}"
    print(hdr)
    for name, n, t, avg, mn, mx in rows:
        if t / tot < 0.0005:
            continue
        line = f"{name[:150]:150s} {n:7d} {t:10.3f} {100 * t / tot:6.2f} {avg:10.1f} {mn:9.1f} {mx:10.1f}"
        if steps:
            line += f" {t / steps:9.3f}"
        print(line)


if __name__ == "__main__":
    main()
