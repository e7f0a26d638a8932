#!/usr/bin/env python
"""Summarises rocprofv3 rocpd (sqlite) outputs into the text tables kept under profiles/.

  python tools/rocpd_summary.py stats  <results.db>            # per-kernel calls / avg duration (--stats view)
  python tools/rocpd_summary.py pmc    <results.db> [substr]   # per-kernel mean of every collected counter
"""
import sqlite3
import sys


def short(name, n=70):
    name = name.replace('(anonymous namespace)::', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print('%-72s %8s %14s %12s %8s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for name, calls, total, avg, pct in rows:
        print('%-72s %8d %14.3f %12.3f %8.2f' % (short(name), calls, total, avg, pct))


def pmc(db, substr=''):
    cur = sqlite3.connect(db).cursor()
    q = ('select kernel_name, counter_name, count(*), avg(value), avg(duration), grid_size, workgroup_size, '
         'lds_block_size, vgpr_count, sgpr_count from counters_collection group by kernel_name, counter_name')
    print('%-60s %-24s %6s %18s %12s %9s %5s %7s %5s %5s' % ('kernel', 'counter', 'n', 'mean_value',
                                                          'avg_ns', 'grid', 'wg', 'lds', 'vgpr', 'sgpr'))
    for r in cur.execute(q):
        if substr and substr not in r[0]:
            continue
        print('%-60s %-24s %6d %18.1f %12.1f %9d %5d %7d %5d %5d' % (short(r[0], 60), r[1], r[2], r[3], r[4],
                                                                    r[5], r[6], r[7], r[8], r[9]))


def seq(db, n='120'):
    """the last n kernel dispatches in start order: start offset, duration, gap to the previous one (us)"""
    con = sqlite3.connect(db)
    cur = con.cursor()
    try:
        rows = list(cur.execute('select name, start, end from kernels order by start'))
    except sqlite3.Error as e:
        print('no `kernels` view (%s); objects: %s' % (e, [r[0] for r in cur.execute("select name from sqlite_master")]))
        return
    rows = rows[-int(n):]
    t0, prev_end = rows[0][1], None
    for name, st, en in rows:
        gap = (st - prev_end) / 1000.0 if prev_end is not None else 0.0
        print('%10.1f %8.2f %7.2f  %s' % ((st - t0) / 1000.0, (en - st) / 1000.0, gap, short(name, 90)))
        prev_end = en


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc, 'seq': seq}[sys.argv[1]](*sys.argv[2:])
