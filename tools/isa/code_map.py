#!/usr/bin/env python3
"""Where a kernel's code lies and which 64-byte lines of it the steady-state loop touches -- the instruction-cache footprint of the lane-group team kernels
(VERDICT r5 "next 6": four episodes per team = 82 KB of code for a 64 KB instruction cache).

  python tools/isa/code_map.py [--unit rollout_team4_nominal.hip] [--kernel serl_rollout_teamg4_kernel_nominal] [-D...] [--runs]

The unit is compiled as the product compiles it plus line tables; every instruction is symbolised with its inline chain (llvm-symbolizer --inlines).  An
instruction is COLD when a frame of its chain is one of: the general-purpose ocml bodies behind the short libm's range guards (sincos / pow / exp / log10 /
atan large-argument paths), the plain index-search and look-up passes that only run behind an interval repair, staging (once per launch), episode reset.
Reported: bytes of code, cold bytes, the number of distinct 64 B lines that hold at least one HOT instruction (= what the loop keeps asking the cache for),
lines that hold both (cold code interleaved with hot), and with --runs the cold runs in address order."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from serl_amd import build as B

LLVM = '/opt/rocm/lib/llvm/bin'
COLD = re.compile(r'^(sincos|sin|cos|tan|pow|exp|log10|log|atan|__ocml|__ockl|citw_search_count|citw_search_pass|citw_search<|citw_lookup2d<|citw_lookup1d<|'
                  r'citw_lookup2d_pass|citw_lookup1d_pass|cit_lookup_index_slow|citw_team_stage_|serl_stage_actor_lds|cit_reset|citw_reset|serl_team_reset)')


def main():
    args = sys.argv[1:]
    unit = args[args.index('--unit') + 1] if '--unit' in args else 'rollout_team4_nominal.hip'
    flags = [a for a in args if a.startswith('-D') or a.startswith('-m')]
    with tempfile.TemporaryDirectory() as td:
        co, elf = os.path.join(td, 'u.co'), os.path.join(td, 'u.elf')
        r = subprocess.run([B.HIPCC] + B.FLAGS + flags + ['--cuda-device-only', '-gline-tables-only', '-c', os.path.join(B.CSRC, unit), '-o', co], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + co, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf], check=True)
        dis = subprocess.run([LLVM + '/llvm-objdump', '-d', elf], capture_output=True, text=True).stdout
        lines = dis.split('\n')
        starts = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <', l)]
        kernels = [re.search(r'<(.*)>', lines[i]).group(1) for i in starts]
        want = args[args.index('--kernel') + 1] if '--kernel' in args else None
        out = []
        for j, i in enumerate(starts):
            if want is not None and want not in lines[i]:
                continue
            if want is None and 'kernel' not in lines[i]:
                continue
            body = lines[i + 1:(starts[j + 1] if j + 1 < len(starts) else len(lines))]
            insts = []
            for ln in body:
                m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):((?: [0-9A-F]{8})+)', ln)
                if m:
                    insts.append((int(m.group(3), 16), 4 * len(m.group(4).split())))
            if not insts:
                continue
            sym = subprocess.run([LLVM + '/llvm-symbolizer', '--obj=' + elf, '--inlines', '--functions=short'], input='\n'.join('0x%x' % a for a, _ in insts),
                                 capture_output=True, text=True).stdout
            chains = []
            for blk in sym.strip().split('\n\n'):
                ls = blk.split('\n')
                chains.append([ls[k] for k in range(0, len(ls) - 1, 2)])
            assert len(chains) == len(insts)
            base = insts[0][0]
            hot_lines, cold_lines = set(), set()
            hot_by = collections.Counter()
            cold_bytes = collections.Counter()
            runs = []
            for (addr, size), ch in zip(insts, chains):
                why = next((f for f in ch if COLD.match(f)), None)
                line = (addr - base) // 64
                if why:
                    cold_lines.add(line)
                    cold_bytes[COLD.match(why).group(1)] += size
                    if runs and runs[-1][1] == addr:
                        runs[-1][1] = addr + size
                    else:
                        runs.append([addr, addr + size, why])
                else:
                    hot_lines.add(line)
                    # the outermost frame below the kernel that names a part: a role function of the generated evaluation, the actor wavefront, the team skeleton
                    part = next((f for f in reversed(ch[:-1]) if re.match(r'(citw_\w+_eval_w\d|serl_team\w*_actor\w*|serl_actor_forward\w*|citw_team\w*_integrate\w*|serl_team\w*_episode\w*|serl_half\w*)', f)), ch[-1])
                    hot_by[re.sub(r'<.*', '', part)] += size
            total = insts[-1][0] + insts[-1][1] - base
            res = dict(kernel=kernels[j], code_bytes=total, cold_bytes=sum(cold_bytes.values()), cold_by_frame=dict(cold_bytes),
                       hot_lines_64B=len(hot_lines), hot_footprint_bytes=64 * len(hot_lines), lines_hot_and_cold=len(hot_lines & cold_lines),
                       hot_bytes_by_part=dict(hot_by.most_common()), cold_runs=len(runs), cold_bytes_in_the_last_eighth=sum(b - a for a, b, _ in runs if a - base >= total * 7 // 8))
            if '--runs' in args:
                res['runs'] = [dict(at=a - base, bytes=b - a, frame=w) for a, b, w in runs if b - a >= 64]
            out.append(res)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
