#!/usr/bin/env python3
"""Every pairing of the seven team roles on the four SIMDs of a CU (hardware wavefronts w and w + 4 share one; wavefront 7 is the actor):
105 role maps for libserl_amd_devroles.so (-DSERL_DEV_ROLE_MAP=1: the map comes in through SERL_JITTER_SITES, a nibble per hardware wavefront).
    python tools/sweep_roles.py            -> one hex map per line (the compiled map of rollout_team_<v>.hip first)"""
import itertools


def pairings(items):
    if not items:
        yield []
        return
    a = items[0]
    for i in range(1, len(items)):
        rest = items[1:i] + items[i + 1:]
        for p in pairings(rest):
            yield [(a, items[i])] + p


def to_map(roles_of_wave):
    m = 0
    for w, r in enumerate(roles_of_wave):
        m |= r << (4 * w)
    return m


if __name__ == '__main__':
    shipped = [1, 3, 5, 0, 2, 4, 6, 7]      # SERL_TEAM_ROLES of rollout_team_<v>.hip (the LDS-resident actor; SERL_TEAMS_ROLES, beside a streaming actor, is 0 6 2 3 4 5 1 7)
    seen = set()
    out = ['%08x' % to_map(shipped)]
    seen.add(frozenset([frozenset((shipped[w], shipped[w + 4])) for w in range(4)]))
    for beside_actor in range(7):
        others = [r for r in range(7) if r != beside_actor]
        for p in pairings(others):
            key = frozenset([frozenset(x) for x in p] + [frozenset((beside_actor, 7))])
            if key in seen:
                continue
            seen.add(key)
            waves = [0] * 8
            for s, (a, b) in enumerate(p):
                waves[s], waves[s + 4] = a, b
            waves[3], waves[7] = beside_actor, 7
            out.append('%08x' % to_map(waves))
    print('\n'.join(out))
