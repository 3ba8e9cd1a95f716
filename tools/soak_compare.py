import numpy as np, json, sys, collections
res = {}
for b in sys.argv[1:]:
    d = np.load('gpurun_out/r06soak/%s.npz' % b)
    by = collections.defaultdict(dict)
    for k in d.files:
        case, seed, key = k.rsplit('_', 2) if k.count('_') >= 2 else (k, '', '')
        parts = k.split('_')
        # <case>_<seed>_<key> where key may contain underscores: find the integer part
        for i, p in enumerate(parts):
            if p.isdigit():
                case, seed, key = '_'.join(parts[:i]), int(p), '_'.join(parts[i + 1:]); break
        by[(case, key)][seed] = d[k]
    bad = 0; n = 0
    for (case, key), m in by.items():
        seeds = sorted(m)
        for s in seeds[1:]:
            n += 1
            if not np.array_equal(m[seeds[0]], m[s], equal_nan=True): bad += 1
    res[b] = dict(cases=sorted(set(c for c, _ in by)), seeds=len(set(s for m in by.values() for s in m)), comparisons=n, mismatches=bad)
print(json.dumps(res))
