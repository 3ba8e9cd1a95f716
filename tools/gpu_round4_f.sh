#!/bin/bash
# round 4, session f: balancer sweep after the short libm (tools/sweep_libs.txt), light barrier profile, i-cache counters of the mixed sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
REPS=2 bash tools/sweep_run.sh r04f > $O/sweep_print.txt 2>&1
cat $O/sweep_print.txt
SERL_PROFILE=1 SERL_LIB=$R/serl_amd/csrc/libserl_amd_prof3.so timeout 300 python tools/ab.py 150 > $O/ab_prof3.txt 2>> $O/err.txt
cat $O/ab_prof3.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc_icm -o icm -- python $R/bench.py --workload mixed --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_icm.log 2>&1
cd $R
python - $O/pmc_icm <<'PY' > $O/pmc_icache_mixed.json
import sqlite3, sys, glob, json
out = []
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    pi = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')]
    rows = sorted(con.execute('select id, start, end, kernel_id from %s' % kd), key=lambda r: r[1] - r[2])[:4]
    for r in rows:
        d = {'ms': (r[2] - r[1]) / 1e6, 'start_ms': r[1] / 1e6}
        if ks:
            try:
                d['kernel'] = list(con.execute('select kernel_name from %s where id=%d' % (ks[0], r[3])))[0][0][:60]
            except Exception:
                pass
        for n, v in con.execute('select i.name, sum(e.value) from %s e join %s i on e.pmc_id=i.id where e.event_id in (select event_id from %s where id=%d) group by i.name' % (pe, pi, kd, r[0])):
            d[n] = v
        out.append(d)
print(json.dumps(out, indent=1))
PY
cat $O/pmc_icache_mixed.json
rm -rf $O/pmc_icm
