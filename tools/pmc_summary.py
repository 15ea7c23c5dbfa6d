"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd sqlite db.  python tools/pmc_summary.py <db>"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:70]


def main(path):
    con = sqlite3.connect(path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    print('tables/views:', [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower() or t in ('kernels',)])
    for t in tabs:
        if 'counters_collection' in t or t == 'counters_collection':
            cols = [r[1] for r in con.execute('pragma table_info(%s)' % t)]
            print(t, cols)
    view = 'counters_collection' if 'counters_collection' in tabs else None
    if not view:
        return
    cols = [r[1] for r in con.execute('pragma table_info(%s)' % view)]
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    cn = 'counter_name' if 'counter_name' in cols else 'counter'
    vn = 'value' if 'value' in cols else 'counter_value'
    rows = con.execute('select %s, %s, %s, dispatch_id, grid_size from %s' % (kn, cn, vn, view)).fetchall()
    per = {}
    for k, c, v, d, gsz in rows:
        key = ('%s grid=%s' % (short(k), gsz), c)          # same kernel at different call sites differs by grid
        per.setdefault(key, {}).setdefault(d, 0.0)
        per[key][d] += float(v)                           # sum over XCD/SE instances of one dispatch
    print('%-84s %-14s %10s %16s' % ('kernel', 'counter', 'dispatches', 'mean per dispatch'))
    for (k, c), dd in sorted(per.items()):
        print('%-84s %-14s %10d %16.1f' % (k, c, len(dd), sum(dd.values()) / len(dd)))


if __name__ == '__main__':
    main(sys.argv[1])
