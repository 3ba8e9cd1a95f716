#!/usr/bin/env python3
"""Per-ROLE, per-CATEGORY instruction picture of serl_rollout_team_kernel_<variant> -- the dynamic table the rounds before tuned without.

  python tools/isa/role_profile.py [--variant nominal] [--pmc gpurun_out/<tag>/roleprof.json] [--md] [--json out.json]

PC sampling and thread trace are not available on this pool (rocprofv3: "PC sampling configuration is not supported on any of the agents"; no trace
decoder library in the image), so the picture is assembled from two measurements that ARE available:

 1. STATIC x EXECUTION WEIGHT (here, no GPU).  The unit is compiled exactly as the product compiles it plus line tables; every instruction of the kernel is
    symbolised with its INLINE CHAIN (llvm-symbolizer --inlines), which names the role function it was inlined into (citw_<v>_team_eval_w<r>: a
    wavefront runs one role for the whole episode) and the line of the generated file the call came from.  That line decides how often the
    instruction runs per model evaluation: 1 for the straight-line glue, 1/6 for the block that depends on the command alone (`if (stage == 0)`),
    0 for the operands behind a gate the trimmed flight condition leaves closed, for the interval-repair passes, and for the large-argument /
    ocml fall-backs of the short libm.  Instructions outside the role functions belong to the step skeleton every team wavefront executes (ODE5
    combination, barriers, the step's top) or to the actor wavefront.  Sum over the hot path = modelled instructions per env step.
 2. HARDWARE COUNTERS BY ABLATION (on the GPU: tools/gpu_session.sh <tag> roleprof).  SQ_INSTS_VALU / SALU / LDS / BRANCH / SMEM and the typed
    VALU counters (ADD / MUL / FMA / TRANS _F64, _F32, CVT, INT32 / INT64) of one launch of builds in which ONE role skips its part of the evaluation
    (CITW_ABLATE_MASK, state frozen) against the same build with nothing skipped: the difference is what that role issues per env step.  The
    modelled counts of (1) are checked against them; the categories the hardware does not type (moves, selects, compares, lane reads, waits, nops)
    come from (1).

The work being restated: step() of /root/reference/envs/h2000_v90/_citation.cpython-38-x86_64-linux-gnu.so."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from serl_amd import build as B
from pcsamp import cat_of, CATS

LLVM = '/opt/rocm/lib/llvm/bin'


def gen_line_weights(path, variant):
    """line number of gen/citation_<v>_team.inc -> executions per model evaluation (1, 1/6 or 0), by brace-tracking the emitted text"""
    w = {}
    stack = []          # [(weight, depth at which the region opened)]
    depth = 0
    active = [True]     # preprocessor: one-episode-per-team build (CITW_GROUP_LANES == 64, CITW_SPEC_LOOKUP on, SEARCH/L2 share = 1)
    for ln, text in enumerate(open(path), 1):
        s = text.strip()
        if s.startswith('#if') or s.startswith('#elif') or s.startswith('#else') or s.startswith('#endif'):
            continue      # (the compiler resolves them; lines of the dead branches never appear in the line table)
        cur = stack[-1][0] if stack else 1.0
        opens = None
        if re.match(r'if \(stage == 0\) \{', s):
            opens = cur / 6.0
        elif re.match(r'if \(!?b\d+\) \{\s*/\* only the selects', s) or 'only the selects on this condition read these' in s and s.startswith('if ('):
            opens = 0.0
        elif s.startswith('if (citw_spec_tail<') or s.startswith('if (HAVE_SC && !SC.valid)'):
            opens = 0.0
        w[ln] = cur if opens is None else cur      # the `if` line itself runs with the enclosing weight
        d_open, d_close = s.count('{'), s.count('}')
        if opens is not None and d_open > d_close:
            stack.append((opens, depth))
        elif opens is not None and d_open == d_close:
            # one-line region: `if (HAVE_SC && !SC.valid) { ... }`: its body shares the line -- count the line cold
            w[ln] = opens
        depth += d_open - d_close
        while stack and depth <= stack[-1][1]:
            stack.pop()
    return w


COLD_FRAMES = re.compile(r'^(sincos|sin|cos|tan|pow|__ocml|__ockl|citw_search_count|citw_search_pass|citw_search<|citw_lookup2d<|citw_lookup1d<|citw_lookup2d_pass|citw_lookup1d_pass)')
LIBM_GUARDED = re.compile(r'^(exp|log10|log)$')


def main():
    args = sys.argv[1:]
    variant = args[args.index('--variant') + 1] if '--variant' in args else 'nominal'
    unit = 'rollout_team_%s.hip' % variant
    kname = 'serl_rollout_team_kernel_%s' % variant
    flags = [a for a in args if a.startswith('-D')]
    gen = os.path.join(B.CSRC, 'gen', 'citation_%s_team.inc' % variant)
    wline = gen_line_weights(gen, variant)
    with tempfile.TemporaryDirectory() as td:
        co, elf = os.path.join(td, 'u.co'), os.path.join(td, 'u.elf')
        r = subprocess.run([B.HIPCC] + B.FLAGS + flags + ['--cuda-device-only', '-gline-tables-only', '-c', os.path.join(B.CSRC, unit), '-o', co], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + co, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf], check=True)
        dis = subprocess.run([LLVM + '/llvm-objdump', '-d', elf], capture_output=True, text=True).stdout
        lines = dis.split('\n')
        starts = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l)]
        body = []
        for j, i in enumerate(starts):
            if kname in lines[i]:
                body = lines[i + 1:(starts[j + 1] if j + 1 < len(starts) else len(lines))]
                break
        insts = []
        for ln in body:
            m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', ln)
            if m:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
        sym = subprocess.run([LLVM + '/llvm-symbolizer', '--obj=' + elf, '--inlines', '--functions=short'], input='\n'.join('0x%x' % a for a, _, _ in insts),
                             capture_output=True, text=True).stdout
    chains = []
    for blk in sym.strip().split('\n\n'):
        ls = blk.split('\n')
        chains.append([(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)])
    assert len(chains) == len(insts), (len(chains), len(insts))
    table = collections.defaultdict(lambda: collections.Counter())        # role -> category -> weighted instructions per EVALUATION
    static = collections.defaultdict(lambda: collections.Counter())       # role -> category -> static instructions
    hot_static = collections.Counter()
    for (addr, op, opnds), ch in zip(insts, chains):
        role, wt = None, None
        for k, (fn, loc) in enumerate(ch):
            m = re.match(r'citw_%s_team_eval_w(\d)$' % variant, fn)
            if m:
                role = 'role %s' % m.group(1)
                lm = re.search(r'_team\.inc:(\d+)', loc)
                wt = wline.get(int(lm.group(1)), 1.0) if lm else 1.0
                # cold helper frames below the role frame: the plain search / look-up passes only run behind an interval repair (their call sites are
                # weighted 0 already); ocml bodies behind the short libm's range guards never run on flight angles
                inner = [f for f, _ in ch[:k]]
                if any(re.match(r'^(sincos|sin|cos|tan|pow|atan)$', f) for f in inner):
                    wt = 0.0
                break
        if role is None:
            fns = [f for f, _ in ch]
            if any(f.startswith('serl_team_actor_wave_') for f in fns):
                role = 'actor'
            elif any(f.startswith('citw_team_stage_') or f.startswith('serl_stage_actor_lds') for f in fns):
                role = 'staging (once per episode)'
            elif any(f.startswith('citw_team_integrate_') or f.startswith('serl_team_episode_') for f in fns):
                role = 'team skeleton (x7)'
            else:
                role = 'kernel entry'
        c = cat_of(op + ' ' + opnds)
        static[role][c] += 1
        if wt is not None:
            table[role][c] += wt
            hot_static[role] += wt > 0
    res = dict(variant=variant, kernel=kname, static_instructions=len(insts),
               static_by_role={r: dict(v, total=sum(v.values())) for r, v in static.items()},
               modelled_per_evaluation={r: dict({k: round(x, 1) for k, x in v.items()}, total=round(sum(v.values()), 1)) for r, v in table.items()},
               hot_static={r: int(n) for r, n in hot_static.items()})
    team = sum(sum(v.values()) for v in table.values())
    res['modelled_team_roles_per_env_step'] = round(6 * team, 1)
    if '--json' in args:
        json.dump(res, open(args[args.index('--json') + 1], 'w'), indent=1)
    if '--md' in args:
        cats = [c for c, _ in CATS] + ['other']
        roles = sorted(table)
        print('Modelled instructions per MODEL EVALUATION (static instruction x executions of its source region), role functions only:\n')
        print('| category | ' + ' | '.join(roles) + ' | all roles |')
        print('|---|' + '---:|' * (len(roles) + 1))
        for c in cats:
            row = [table[r].get(c, 0.0) for r in roles]
            if sum(row) < 0.5:
                continue
            print('| %s | ' % c + ' | '.join('%.0f' % x for x in row) + ' | %.0f |' % sum(row))
        print('| **total** | ' + ' | '.join('%.0f' % sum(table[r].values()) for r in roles) + ' | %.0f |' % team)
        print('\nStatic instructions by part of the kernel: ' + ', '.join('%s %d' % (r, sum(v.values())) for r, v in sorted(static.items())))
    else:
        print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
