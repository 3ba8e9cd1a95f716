#!/usr/bin/env python3
"""Static instruction mix of ONE kernel of a unit as the product build compiles it, attributed to source files:

  python tools/isa/kernel_isa.py [unit.hip] [kernel-name substring] [extra hipcc flags ...] [--json]
  (defaults: rollout_team_nominal.hip serl_rollout_team_kernel)

Compiles the unit for gfx950 with the product's flags + line tables (device only), disassembles it and reports for the kernel:
registers / spills / LDS from the code-object notes; instructions; the moves that only materialise a constant -- split into f64 literal
halves (`s_mov_b32 s, 0x3fd65718`), LDS / integer constants (`v_mov_b32 v, 0x234a0`) and zeros -- by the source file of their line;
DS instructions with and without an immediate offset; s_waitcnt / s_nop.  Round 5 used it to count what the register sets of f64 literals
(serl_kregs.h) and the LDS layout remove: profiles/r05_isa_team_kernel.md."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from serl_amd import build as B

LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    args = [a for a in sys.argv[1:] if a != '--json']
    unit = next((a for a in args if a.endswith('.hip')), 'rollout_team_nominal.hip')
    rest = [a for a in args if not a.endswith('.hip')]
    kname = next((a for a in rest if not a.startswith('-')), 'serl_rollout_team_kernel')
    flags = [a for a in rest if a.startswith('-')]
    with tempfile.TemporaryDirectory() as td:
        co, elf = os.path.join(td, 'u.co'), os.path.join(td, 'u.elf')
        r = subprocess.run([B.HIPCC] + B.FLAGS + flags + ['--cuda-device-only', '-gline-tables-only', '-c', os.path.join(B.CSRC, unit), '-o', co],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + co, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--output=' + elf], check=True)
        notes = subprocess.run([LLVM + '/llvm-readelf', '--notes', elf], capture_output=True, text=True).stdout
        dis = subprocess.run([LLVM + '/llvm-objdump', '-d', '-l', elf], capture_output=True, text=True).stdout
    meta = {}
    blocks = re.split(r'(?=\.name:\s)', notes)
    for blk in re.findall(r'- \.agpr_count.*?(?=\n\s+- \.agpr_count|\Z)', notes, re.S):
        nm = re.search(r'\.name:\s+(\S+)', blk)
        if nm and kname in nm.group(1):
            for key in ('vgpr_count', 'vgpr_spill_count', 'sgpr_count', 'sgpr_spill_count', 'group_segment_fixed_size', 'private_segment_fixed_size'):
                m = re.search(r'\.%s:\s+(\d+)' % key, blk)
                if m:
                    meta[key] = int(m.group(1))
            meta['name'] = nm.group(1)
            break
    lines = dis.split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l)]
    body = []
    for j, i in enumerate(starts):
        if kname in lines[i]:
            body = lines[i:(starts[j + 1] if j + 1 < len(starts) else len(lines))]
            break
    ops, by_file, lit = collections.Counter(), collections.Counter(), collections.Counter()
    ds_off = ds_all = n = 0
    cur = '?'
    for ln in body:
        m = re.match(r'; (/\S+):(\d+)', ln)
        if m:
            cur = os.path.basename(m.group(1))
            continue
        m = re.match(r'\s+(\S+)\s+(.*?)\s*//', ln)
        if not m:
            continue
        n += 1
        op, a = m.group(1), m.group(2)
        ops[op] += 1
        if op.startswith('ds_'):
            ds_all += 1
            ds_off += 'offset' in a
        if op in ('s_mov_b32', 'v_mov_b32_e32'):
            mm = re.search(r', (0x[0-9a-f]+)$', a)
            if mm:
                kind = 'int_or_lds_address' if int(mm.group(1), 16) < 0x100000 else 'f64_literal_half'
                lit[kind] += 1
                by_file[(cur, kind)] += 1
            elif re.search(r', 0$', a):
                lit['zero'] += 1
    out = dict(unit=unit, kernel=meta.get('name', kname), flags=flags, meta=meta, instructions=n, constant_moves=dict(lit),
               constant_moves_by_file={'%s: %s' % k: v for k, v in sorted(by_file.items(), key=lambda kv: -kv[1])},
               ds_instructions=ds_all, ds_with_immediate_offset=ds_off, s_waitcnt=ops['s_waitcnt'], s_nop=ops['s_nop'],
               f64_arithmetic=sum(v for k, v in ops.items() if k.startswith(('v_add_f64', 'v_mul_f64', 'v_fma_f64', 'v_fmac_f64', 'v_div_', 'v_rcp_f64', 'v_sqrt_f64', 'v_ldexp_f64', 'v_rndne_f64'))),
               top_ops=dict(ops.most_common(16)))
    if '--json' in sys.argv:
        print(json.dumps(out))
        return
    print('%s  [%s %s]' % (out['kernel'], unit, ' '.join(flags)))
    print('  registers', meta)
    print('  instructions %d   f64 arithmetic %d   constant moves %s   DS %d (%d with an immediate offset)   s_waitcnt %d   s_nop %d'
          % (n, out['f64_arithmetic'], dict(lit), ds_all, ds_off, ops['s_waitcnt'], ops['s_nop']))
    for k, v in list(out['constant_moves_by_file'].items())[:12]:
        print('    %-50s %d' % (k, v))


if __name__ == '__main__':
    main()
