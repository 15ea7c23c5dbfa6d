"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/.
    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:70]


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    rows = con.execute("select name, start, end from kernels").fetchall() if {'name', 'start', 'end'} <= set(cols) else []
    if not rows:
        print('columns:', cols)
        return
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('%-70s %8s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-70s %8d %12.1f %10.2f %10.2f %10.2f %6.2f' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    print('%-70s %8d %12.1f' % ('TOTAL', sum(a[0] for a in agg.values()), tot))


if __name__ == '__main__':
    main(sys.argv[1])
