#!/usr/bin/env python3
"""PC sampling of the rollout kernels (rocprofv3 --pc-sampling-beta-enabled): WHICH instructions a team's wavefronts issue and what they wait for.

  on the GPU box:   python tools/pcsamp.py collect <out.json> [--lib <libserl_amd_*.so>] [--target N] -- <command ...>
                    (tries stochastic sampling in cycles, then host-trap sampling in microseconds; re-runs with a scaled interval until the kernel
                     has ~N samples; writes ONE compact aggregate: counts per (kernel, code-object offset, wavefront of the workgroup, issued?,
                     instruction type, reason-not-issued), the sampler's own disassembly + source line per offset, and the head of the raw files)
  anywhere:         python tools/pcsamp.py report <out.json> [--roles 1,3,5,0,2,4,6,7] [--kernel substring] [--md]
                    role x instruction category x {samples issued, samples stalled by reason} -> per env step when --cycles-per-step is given

Rounds 1 - 5 tuned serl_rollout_team_kernel_<variant> on static ISA counts, SQ totals and s_memtime marks; this is the dynamic per-instruction picture
(VERDICT r5 "missing 2").  The work being restated: step() of /root/reference/envs/h2000_v90/_citation.cpython-38-x86_64-linux-gnu.so, six model
evaluations per env step, driven by the loop /root/reference/base/core/agent.py:85-118."""
import collections, glob, json, os, re, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_key(obj, key):
    """first value stored under `key` anywhere in a nested json value"""
    st = [obj]
    while st:
        o = st.pop()
        if isinstance(o, dict):
            if key in o:
                return o[key]
            st.extend(o.values())
        elif isinstance(o, list):
            st.extend(o[:4] if len(o) > 64 else o)      # (record arrays are long: their first entries are enough to find a key)
    return None


def strip_enum(s, prefixes=('ROCPROFILER_PC_SAMPLING_INSTRUCTION_NOT_ISSUED_REASON_', 'ROCPROFILER_PC_SAMPLING_INSTRUCTION_TYPE_', 'ROCPROFILER_PC_SAMPLING_')):
    s = str(s)
    for p in prefixes:
        if s.startswith(p):
            return s[len(p):]
    return s


def aggregate(outdir, kernel_filter='serl_'):
    """-> dict(meta, rows, inst) from the json (preferred) or csv files rocprofv3 wrote under outdir"""
    files = sorted(glob.glob(os.path.join(outdir, '**', '*'), recursive=True))
    meta = {'files': [(os.path.relpath(f, outdir), os.path.getsize(f)) for f in files if os.path.isfile(f)]}
    raw_head = {}
    for f in files:
        if os.path.isfile(f) and f.endswith(('.csv', '.json', '.txt')):
            with open(f, 'r', errors='replace') as fh:
                raw_head[os.path.relpath(f, outdir)] = fh.read(1500)
    meta['raw_head'] = raw_head
    rows = collections.Counter()
    inst = {}
    kernels = collections.Counter()
    jf = [f for f in files if f.endswith('results.json') or (f.endswith('.json') and 'results' in os.path.basename(f))] or [f for f in files if f.endswith('.json')]
    done = False
    for f in jf:
        if os.path.getsize(f) > 6e9:
            meta['json_skipped'] = 'too large: %d bytes' % os.path.getsize(f)
            continue
        try:
            J = json.load(open(f))
        except Exception as ex:
            meta['json_error'] = repr(ex)[:300]
            continue
        recs = None
        for key in ('pc_sample_stochastic', 'pc_sample_host_trap'):
            recs = find_key(J, key)
            if recs:
                meta['record_key'] = key
                break
        if not recs:
            continue
        instr = find_key(J, 'pc_sample_instructions') or []
        comm = find_key(J, 'pc_sample_comments') or []
        # dispatch id -> kernel name
        ksym = {}
        for k in (find_key(J, 'kernel_symbols') or []):
            if isinstance(k, dict):
                ksym[k.get('kernel_id')] = k.get('formatted_kernel_name') or k.get('kernel_name') or k.get('truncated_kernel_name')
        disp = {}
        for k in (find_key(J, 'kernel_dispatch') or []):
            di = k.get('dispatch_info', k) if isinstance(k, dict) else {}
            disp[di.get('dispatch_id')] = ksym.get(di.get('kernel_id'), str(di.get('kernel_id')))
        meta['n_dispatches'] = len(disp)
        meta['first_records'] = recs[:2]
        knames = {}
        for it in recs:
            r = it.get('record', it)
            pc = r.get('pc', {})
            kn = disp.get(r.get('dispatch_id'), '?')
            kernels[kn] += 1
            if kernel_filter and kernel_filter not in kn:
                continue
            ki = knames.setdefault(kn, len(knames))
            off = pc.get('code_object_offset')
            snap = r.get('snapshot', {}) or {}
            issued = r.get('wave_issued')
            wig = r.get('wave_in_group', r.get('wave_in_grp', -1))
            itype = strip_enum(r.get('inst_type', snap.get('inst_type', '')))
            reason = strip_enum(snap.get('stall_reason', snap.get('reason_not_issued', r.get('reason_not_issued', ''))))
            dual = snap.get('dual_issue_valu', 0)
            rows[(ki, pc.get('code_object_id'), off, wig, -1 if issued is None else int(bool(issued)), itype, reason)] += 1
            ii = it.get('inst_index')
            if off not in inst and ii is not None and ii < len(instr):
                inst[off] = [instr[ii], comm[ii] if ii < len(comm) else '']
        meta['kernel_names'] = {v: k for k, v in knames.items()}
        done = True
        break
    if not done:      # csv fallback: columns differ between releases -- keep what is recognisable
        import csv
        for f in files:
            if f.endswith('.csv') and 'pc_sampling' in os.path.basename(f):
                with open(f, newline='') as fh:
                    rd = csv.DictReader(fh)
                    for r in rd:
                        key = (0, r.get('Code_Object_Id', ''), r.get('Code_Object_Offset', r.get('Instruction', '')), r.get('Wave_In_Group', -1),
                               {'1': 1, '0': 0, 'true': 1, 'false': 0}.get(str(r.get('Wave_Issued_Instruction', r.get('Wave_Issued', ''))).lower(), -1),
                               strip_enum(r.get('Instruction_Type', '')), strip_enum(r.get('Stall_Reason', '')))
                        rows[key] += 1
                        inst.setdefault(key[2], [r.get('Instruction', ''), r.get('Instruction_Comment', '')])
                meta['record_key'] = 'csv:' + os.path.basename(f)
                meta['kernel_names'] = {0: 'all dispatches (csv)'}
    meta['samples_by_kernel'] = dict(kernels.most_common(12))
    meta['samples_kept'] = int(sum(rows.values()))
    return dict(meta=meta, rows=[list(k) + [v] for k, v in sorted(rows.items(), key=lambda kv: -kv[1])], inst={str(k): v for k, v in inst.items()})


def collect(argv):
    out = argv[0]
    cmd = argv[argv.index('--') + 1:]
    opts = argv[1:argv.index('--')]
    target = int(opts[opts.index('--target') + 1]) if '--target' in opts else 400000
    env = dict(os.environ)
    if '--lib' in opts:
        env['SERL_LIB'] = os.path.abspath(opts[opts.index('--lib') + 1])
    kfilter = opts[opts.index('--kernel') + 1] if '--kernel' in opts else 'serl_'
    base = '/tmp/pcsamp_%d' % os.getpid()
    tries = [('stochastic', 'cycles', 1 << 20), ('host_trap', 'time', 512)]
    log, best = [], None
    for method, unit, interval in tries:
        for attempt in range(3):
            d = '%s_%s_%d' % (base, method, attempt)
            full = ['rocprofv3', '--kernel-trace', '--pc-sampling-beta-enabled', '--pc-sampling-method', method, '--pc-sampling-unit', unit,
                    '--pc-sampling-interval', str(interval), '--output-format', 'csv', 'json', '-d', d, '-o', 'pcs', '--'] + cmd
            t0 = time.time()
            r = subprocess.run(full, capture_output=True, text=True, env=env, cwd='/tmp', timeout=1500)
            ent = dict(method=method, unit=unit, interval=interval, rc=r.returncode, seconds=round(time.time() - t0, 1), stderr_tail=r.stderr[-1200:], stdout_tail=r.stdout[-600:])
            agg = None
            if r.returncode == 0 or os.path.isdir(d):
                try:
                    agg = aggregate(d, kfilter)
                    ent['samples_kept'] = agg['meta']['samples_kept']
                    ent['samples_by_kernel'] = agg['meta']['samples_by_kernel']
                except Exception as ex:
                    ent['aggregate_error'] = repr(ex)[:500]
            log.append(ent)
            subprocess.run(['rm', '-rf', d])
            n = (agg or {}).get('meta', {}).get('samples_kept', 0)
            if agg and n > 0 and (best is None or abs(n - target) < abs(best['meta']['samples_kept'] - target)):
                best = agg
                best['meta'].update(method=method, unit=unit, interval=interval, command=' '.join(cmd), lib=env.get('SERL_LIB', 'product'))
            if not agg or n == 0:
                break                                   # this method does not work here: the next one
            if 0.4 * target <= n <= 3 * target:
                break
            scale = n / float(target)                    # more samples than wanted -> a longer interval
            new = interval * scale
            if unit == 'cycles':                         # stochastic intervals are powers of two
                new = 1 << max(8, min(30, int(round(__import__('math').log2(max(new, 256))))))
            else:
                new = max(1, int(new))
            if new == interval:
                break
            interval = new
        if best is not None:
            break
    res = best or dict(meta={}, rows=[], inst={})
    res['meta']['attempts'] = log
    json.dump(res, open(out, 'w'))
    print(json.dumps({k: v for k, v in res['meta'].items() if k in ('method', 'unit', 'interval', 'samples_kept', 'samples_by_kernel', 'record_key')})[:1500])
    for e in log:
        print(json.dumps({k: e[k] for k in ('method', 'interval', 'rc', 'seconds', 'samples_kept') if k in e}), e.get('aggregate_error', ''), e['stderr_tail'][-300:].replace('\n', ' | ') if e['rc'] else '')


# ---------------------------------------------------------------------------------------------------------------------------------------
CATS = [('f64 add/mul/fma', r'^v_(add|mul|fma|fmac)_f64'), ('f64 div/sqrt helpers', r'^v_(div_scale|div_fmas|div_fixup|rcp|rsq|sqrt|trig_preop|ldexp|frexp)\w*_f64|^v_(rcp|rsq|sqrt)_f64'),
        ('f64 min/max/cmp', r'^v_(min|max|cmp\w*)_f64|^v_cmp\w*_f64|^v_cmpx?_\w+_f64'), ('f64 convert', r'^v_cvt_\w*f64|^v_cvt_f64|^v_(floor|ceil|trunc|rndne|fract)_f64'),
        ('f32 arithmetic (actor)', r'^v_(pk_)?(add|sub|subrev|mul|fma|fmac|mac|mad|max|min|rcp|rsq|sqrt|exp|log|cvt)\w*_f(32|16)'),
        ('v_cndmask', r'^v_cndmask'), ('v_mov', r'^v_(mov|accvgpr)'), ('readlane / DPP / permute', r'^v_(readlane|readfirstlane|writelane|permlane|mov_b32_dpp)|^ds_(bpermute|permute|swizzle)|_dpp'),
        ('other v_cmp', r'^v_cmpx?_'), ('VALU integer / bit', r'^v_'),
        ('LDS', r'^ds_'), ('s_waitcnt', r'^s_waitcnt'), ('s_nop', r'^s_nop'), ('s_barrier', r'^s_barrier'), ('s_sleep', r'^s_sleep'),
        ('branch', r'^s_(cbranch|branch|setpc|swappc|call)'), ('exec mask', r'^s_\w+_saveexec|^s_(and|or|andn2|orn2|xor|mov|not)_b64.*exec|exec'),
        ('s_mov / s_cselect', r'^s_(mov|cmov|cselect|movk)'), ('scalar memory', r'^s_(load|buffer_load|store|memtime|memrealtime|dcache)'),
        ('vector memory', r'^(global|flat|buffer|scratch)_'), ('SALU other', r'^s_')]


def cat_of(text):
    t = text.strip()
    for name, pat in CATS:
        if re.search(pat, t):
            return name
    return 'other'


def report(argv):
    J = json.load(open(argv[0]))
    roles = [int(x) for x in argv[argv.index('--roles') + 1].split(',')] if '--roles' in argv else None
    ksub = argv[argv.index('--kernel') + 1] if '--kernel' in argv else None
    per_step = float(argv[argv.index('--wave-cycles-per-step') + 1]) if '--wave-cycles-per-step' in argv else None
    names = {int(k): v for k, v in J['meta'].get('kernel_names', {}).items()}
    inst = J['inst']
    tab = collections.defaultdict(lambda: collections.Counter())      # (role, category) -> Counter(issued / reason)
    tot_role = collections.Counter()
    total = 0
    hot = collections.Counter()
    for ki, co, off, wig, issued, itype, reason, n in J['rows']:
        kn = names.get(ki, '?')
        if ksub and ksub not in kn:
            continue
        role = ('w%s' % wig) if roles is None or not (0 <= int(wig) < len(roles)) else ('role %d' % roles[int(wig)] if roles[int(wig)] < 7 else 'actor')
        text = (inst.get(str(off)) or ['?', ''])[0]
        c = cat_of(text)
        key = 'issued' if issued == 1 else (reason or 'not issued')
        tab[(role, c)][key] += n
        tot_role[role] += n
        total += n
        hot[(role, off, text, (inst.get(str(off)) or ['', ''])[1])] += n
    res = dict(meta={k: J['meta'].get(k) for k in ('method', 'unit', 'interval', 'command', 'lib', 'samples_kept', 'samples_by_kernel')}, total=total,
               by_role={r: n for r, n in sorted(tot_role.items())},
               table=[dict(role=r, category=c, samples=int(sum(v.values())), **{k: int(x) for k, x in v.items()}) for (r, c), v in sorted(tab.items(), key=lambda kv: (kv[0][0], -sum(kv[1].values())))],
               hottest=[dict(role=r, offset=o, inst=t, src=s, samples=n) for (r, o, t, s), n in hot.most_common(60)])
    if '--md' in argv:
        reasons = sorted({k for v in tab.values() for k in v}, key=lambda k: (k != 'issued', k))
        print('| role | category | samples | share of role |' + ''.join(' %s |' % k.lower() for k in reasons))
        print('|---|---|---:|---:|' + '---:|' * len(reasons))
        for (r, c), v in sorted(tab.items(), key=lambda kv: (kv[0][0], -sum(kv[1].values()))):
            s = sum(v.values())
            if s < 0.004 * tot_role[r]:
                continue
            print('| %s | %s | %d | %.1f %% |' % (r, c, s, 100.0 * s / tot_role[r]) + ''.join(' %d |' % v.get(k, 0) for k in reasons))
    else:
        print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if len(sys.argv) >= 3 and sys.argv[1] == 'collect':
        collect(sys.argv[2:])
    elif len(sys.argv) >= 3 and sys.argv[1] == 'report':
        report(sys.argv[2:])
    else:
        sys.exit(__doc__)
