#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc sqlite outputs (rocpd) of the largest kernel: counter totals and per-step values."""
import sqlite3, sys, glob, json
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
out = {}
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    pi = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    # pick the longest dispatch
    rows = list(con.execute('select id, start, end from %s' % kd))
    best = max(rows, key=lambda r: r[2] - r[1])
    out['kernel_ms'] = (best[2] - best[1]) / 1e6
    cols = [r[1] for r in con.execute('pragma table_info(%s)' % pe)]
    q = 'select i.name, sum(e.value) from %s e join %s i on e.pmc_id=i.id where e.event_id in (select event_id from %s where id=%d) group by i.name' % (pe, pi, kd, best[0])
    try:
        res = list(con.execute(q))
    except Exception:
        res = list(con.execute('select i.name, sum(e.value) from %s e join %s i on e.pmc_id=i.id group by i.name' % (pe, pi)))
    for n, v in res:
        out[n] = v
print(json.dumps({k: (v / steps if k != 'kernel_ms' else v) for k, v in sorted(out.items())}, indent=1))
