#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) as a small markdown table.

  python tools/rocpd_summary.py gpurun_out/prof_r01/r01_results.db "command line" > profiles/<name>.md
"""
import sqlite3, sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print('# rocprofv3 --kernel-trace --stats summary\n')
    if cmd:
        print('command: `%s`\n' % cmd)
    print('| kernel | calls | total ms | avg ms | % |')
    print('|---|---:|---:|---:|---:|')
    for name, calls, tot, avg, pct in rows[:12]:
        short = name if len(name) < 90 else name[:87] + '...'
        print('| `%s` | %d | %.3f | %.4f | %.4f |' % (short, calls, tot / 1e3, avg / 1e3, pct))
    others = rows[12:]
    if others:
        print('| (%d other kernels: torch fills/copies, rocFFT) | %d | %.3f | | %.4f |'
              % (len(others), sum(r[1] for r in others), sum(r[2] for r in others) / 1e3, sum(r[4] for r in others)))
    print()
    cur = c.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                    "duration from kernels where name like '%serl_%' order by start")
    print('## serl kernel dispatches\n')
    print('| kernel | grid | block | LDS B | scratch B/lane | VGPR | AGPR | SGPR | ms |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|---:|')
    for r in cur:
        print('| `%s` | %d | %d | %d | %d | %d | %d | %d | %.3f |' % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8] / 1e6))


if __name__ == '__main__':
    main()
