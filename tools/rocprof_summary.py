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


def hook_intervals(path, marker):
    """Third table: every correction call (a launch of corr_prepare_kernel) with the launches around it -- from the end of the last
    plain-step launch before it to the start of the first plain-step launch after its posterior update: how much of that interval the
    GPU spends in kernels and how much idle (graph replay boundaries, eager launches)."""
    con = sqlite3.connect(path)
    rows = [(short(n), s, e) for n, s, e in con.execute("select name, start, end from kernels order by start").fetchall()]
    hooks = [i for i, r in enumerate(rows) if r[0].startswith('corr_prepare_kernel')]
    out, gaps = [], []
    for h in hooks:
        a = h
        while a > 0 and marker not in rows[a][0]:            # back to the start of the hook step's own forward
            a -= 1
        b = h
        while b < len(rows) and not rows[b][0].startswith('posterior'):
            b += 1
        if a <= 0 or b + 1 >= len(rows) or b - a > 40:       # (a correction call outside a sampling loop, e.g. a warm-up: no forward before it)
            continue
        t0, t1 = rows[a - 1][2], rows[b + 1][1]              # end of the previous plain step's last launch .. start of the next step's first
        busy = sum(e - s for _, s, e in rows[a:b + 1])
        out.append(((t1 - t0) / 1e3, busy / 1e3, b + 1 - a))
        gaps.append([(rows[i][0], (rows[i][1] - rows[i - 1][2]) / 1e3) for i in range(a, b + 2)])
    if not out:
        return
    print()
    print('correction steps (%d found): interval from the end of the previous plain step to the start of the next one' % len(out))
    print('    mean interval %.1f us, of which kernels %.1f us (%d launches), idle %.1f us' % (
        sum(o[0] for o in out) / len(out), sum(o[1] for o in out) / len(out), round(sum(o[2] for o in out) / len(out)),
        sum(o[0] - o[1] for o in out) / len(out)))
    n = min(len(g) for g in gaps)
    big = [(i, sum(g[i][1] for g in gaps) / len(gaps)) for i in range(n)]
    print('    idle before a launch (mean us, positions with >= 5 us): ' + ', '.join('%s %.0f' % (gaps[0][i][0][:24], v) for i, v in big if v >= 5))


def big_gaps(path, thresh_us=40.0):
    """Fourth table: idle gaps of the whole trace above a threshold, grouped by (launch before -> launch after): graph-replay boundaries and
    host-bound stretches show up here."""
    con = sqlite3.connect(path)
    rows = [(short(n), s, e) for n, s, e in con.execute("select name, start, end from kernels order by start").fetchall()]
    agg, end = {}, rows[0][2]
    for i in range(1, len(rows)):
        g = (rows[i][1] - end) / 1e3
        if g >= thresh_us:
            a = agg.setdefault((rows[i - 1][0][:34], rows[i][0][:34]), [0, 0.0])
            a[0] += 1; a[1] += g
        end = max(end, rows[i][2])
    print()
    print('idle gaps >= %.0f us (GPU has no kernel running), by the launches around them:' % thresh_us)
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print('    %5d x mean %8.1f us   %s -> %s' % (n, t / n, a, b))


if __name__ == '__main__':
    main(sys.argv[1])
    if len(sys.argv) > 2:
        by_position(sys.argv[1], sys.argv[2])
        hook_intervals(sys.argv[1], sys.argv[2])
        big_gaps(sys.argv[1])
