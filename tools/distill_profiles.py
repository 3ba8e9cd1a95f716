#!/usr/bin/env python3
"""Copy the summaries of one `tools/profile_round.sh <tag>` run from gpurun_out/<tag>/ into profiles/<series>_*:
   python tools/distill_profiles.py r03p r03_a        (also refreshes profiles/pmc_current.json and floors_current.json, which bench.py reads)"""
import json, shutil, sys
tag, series = sys.argv[1], sys.argv[2]
O, P = 'gpurun_out/' + tag, 'profiles'
for n in ('serl50', 'total512', 'serl10', 'serl10_samecu', 'serl10_pop128', 'pop64', 'pop128', 'pop341', 'mixed', 'mixed_nofused', 'mixed_total2048', 'rccl1'):
    line = [l for l in open('%s/bench_%s.json' % (O, n)) if l.startswith('{')][-1]
    open('%s/%s_bench_%s.json' % (P, series, n), 'w').write(line)
for n in ('serl50', 'serl10', 'serl10_pop128', 'pop512'):
    shutil.copy('%s/kernel_stats_%s.md' % (O, n), '%s/%s_kernel_stats_%s.md' % (P, series, n))
shutil.copy(O + '/valu_latency.json', '%s/%s_valu_latency.json' % (P, series))
shutil.copy(O + '/valu_latency.json', P + '/valu_latency_current.json')      # bench.py: roofline_fp64.peak_measured
shutil.copy(O + '/critical_path.json', '%s/%s_critical_path.json' % (P, series))
open('%s/%s_cycle_profile.txt' % (P, series), 'w').write(''.join(open(O + '/' + f).read() for f in ('ab.txt', 'ab_prof1.txt', 'ab_prof2.txt') if __import__('os').path.exists(O + '/' + f)))
shutil.copy(O + '/refill.json', '%s/%s_refill.json' % (P, series))
sha = open(O + '/csrc_sha256.txt').read().strip()      # serl_amd/build.py source_hash() of the tree that was measured
floors = json.load(open(O + '/critical_path.json')); floors['csrc_sha256'] = sha
json.dump(floors, open(P + '/floors_current.json', 'w'), indent=1)
sq, pf, pw = (json.load(open('%s/pmc_%s.json' % (O, k))) for k in ('sq', 'fetch', 'write'))
cp = json.load(open(O + '/critical_path.json'))
b = json.loads(open('%s/%s_bench_serl50.json' % (P, series)).read())
steps = 150 * 8001
traffic = 2 * pf['FETCH_SIZE'] * 1024 + pw['WRITE_SIZE'] * 1024
wc = sq['SQ_WAVE_CYCLES']
pmc = dict(
    command='rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline  (three separate passes: '
            'FETCH_SIZE | WRITE_SIZE | SQ_*; tools/profile_round.sh)',
    workload='serl50', pop=50, workload_desc='pop=50 x 3 evals x 8001 steps, 1 GPU (bench.py default)',
    kernel='serl_rollout_team_kernel_nominal: seven wavefronts integrate the model, an eighth runs the actor of the next step beside them (two per SIMD)',
    FETCH_SIZE_KB=pf['FETCH_SIZE'], WRITE_SIZE_KB=pw['WRITE_SIZE'], kernel_ms=[pf['kernel_ms'], pw['kernel_ms'], sq['kernel_ms']],
    corrections='MI355X_MICROARCH.md HBM section: rocprofv3 on gfx950 tallies FETCH_SIZE at half the bytes of the 128-B requests -> doubled; WRITE_SIZE as reported',
    traffic_bytes_per_launch=traffic, algorithmic_bytes_per_launch=steps * 48, ratio=traffic / (steps * 48),
    sq={k: v for k, v in sq.items() if k != 'kernel_ms'}, csrc_sha256=sha)
pmc['issue'] = dict(
    bound='valu-issue / dependent latency (63 of 64 lanes of every glue instruction carry the same scalar; two wavefronts per SIMD)',
    active_frac=sq['SQ_ACTIVE_INST_ANY'] / wc, wait_frac=sq['SQ_WAIT_ANY'] / wc, issue_stall_frac=sq['SQ_WAIT_INST_ANY'] / wc,
    note='fractions of SQ_WAVE_CYCLES summed over the eight wavefronts of a workgroup (two per SIMD: a SIMD issues for one of them at a time; '
         'the actor wavefront is parked ~80 % of its time by design)',
    simd_issue_frac=2 * sq['SQ_ACTIVE_INST_ANY'] / wc,
    valu_per_env_step=sq['SQ_INSTS_VALU'] / steps, salu_per_env_step=sq['SQ_INSTS_SALU'] / steps, lds_per_env_step=sq['SQ_INSTS_LDS'] / steps,
    issue_floor_us_per_env_step=cp['issue_floor_trimmed_us_per_env_step'], issue_floor_full_dag_us_per_env_step=cp['issue_floor_us_per_env_step'],
    dependency_floor_us_per_env_step=cp['dependency_floor_us_per_env_step'],
    measured_us_per_env_step=b['t_step_us'], issue_floor_frac=cp['issue_floor_trimmed_us_per_env_step'] / b['t_step_us'],
    issue_floor_frac_full_dag=cp['issue_floor_us_per_env_step'] / b['t_step_us'],
    frac_note='issue floor (minimal instruction count of what the trimmed flight condition executes x 4 cycles over 4 SIMDs, tools/dag/critical_path.py; `_full_dag`: every node) / measured time '
              'per env step; the dependency floor weights a look-up with its own dependent chain (three LDS round trips, two divisions)',
    source='profiles/%s_pmc.json, profiles/%s_critical_path.json, profiles/%s_valu_latency.json' % (series, series, series))
json.dump(pmc, open('%s/%s_pmc.json' % (P, series), 'w'), indent=1)
json.dump(pmc, open(P + '/pmc_current.json', 'w'), indent=1)
print(json.dumps(pmc['issue'], indent=1))
print('traffic ratio', traffic / (steps * 48))
