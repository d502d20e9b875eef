#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (markdown/CSV-like).
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    rows = sorted(rows, key=lambda r: (0 if "zsr::" in r[0] else 1, -r[2]))  # this library's kernels first, then setup (torch) kernels
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
