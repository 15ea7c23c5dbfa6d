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


def by_position(path, marker):
    """Second table: the launches of one denoiser step in order (a step starts at each kernel whose name contains `marker`), with the
    mean duration of each position and the mean gap to the previous launch's end -- the kernel boundaries inside the replayed graph."""
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    steps, cur = [], None
    for name, s, e in rows:
        n = short(name)
        if marker in n:
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((n, s, e))
    if not steps:
        return
    from collections import Counter
    length = Counter(len(st) for st in steps).most_common(1)[0][0]
    sig = Counter(tuple(n for n, _, _ in st) for st in steps if len(st) == length).most_common(1)[0][0]
    sel = [st for st in steps if tuple(n for n, _, _ in st) == sig]
    print()
    print('launch order inside one step (%d steps with the most common %d-launch sequence; marker %r):' % (len(sel), length, marker))
    print('%3s %-70s %10s %10s' % ('#', 'kernel', 'avg_us', 'gap_us'))
    tk = tg = 0.0
    for i in range(length):
        d = sum(st[i][2] - st[i][1] for st in sel) / len(sel) / 1e3
        g = sum(st[i][1] - st[i - 1][2] for st in sel) / len(sel) / 1e3 if i else 0.0
        tk += d; tg += g
        print('%3d %-70s %10.2f %10.2f' % (i, sig[i], d, g))
    span = sum(st[-1][2] - st[0][1] for st in sel) / len(sel) / 1e3
    print('    sum of kernels %.1f us + gaps %.1f us = %.1f us from the start of the first launch to the end of the last' % (tk, tg, span))


if __name__ == '__main__':
    main(sys.argv[1])
    if len(sys.argv) > 2:
        by_position(sys.argv[1], sys.argv[2])
