"""The oracle's dynamics restatement (oracle/citation_ref.c) pinned against the reference binary.

  * golden: tests/golden/dyn_open_loop.npz holds what the REFERENCE'S OWN shared object returned for a seeded
    3 000-step command sequence, for all 9 distinct builds (5 code variants) -- bit-exact.
  * live (build container only, marker needs_reference): every one of the 14 build directories, step by
    step against the loaded reference library -- outputs, the 19 continuous states and all block signals
    rtB bit-exact, including re-initialisation.
"""
import numpy as np
import pytest

BUILDS9 = ['h2000_v90', 'h2000_v150', 'h10000_v90', 'cg', 'cg_for', 'cg_timed', 'ice', 'gust', 'test']
BUILDS14 = BUILDS9 + ['be', 'jr', 'sa', 'se', 'noise']


@pytest.mark.parametrize('build', BUILDS9)
def test_open_loop_golden_bit_exact(golden, build):
    from oracle.dynamics import CitationDynamics
    g = golden('dyn_open_loop')
    sim = CitationDynamics(build)
    xs = np.stack([sim.step(c) for c in g[build + '_cmd']])
    np.testing.assert_array_equal(xs[::10], g[build + '_x'])
    np.testing.assert_array_equal(xs[-1], g[build + '_xlast'])


def test_first_step_returns_initial_conditions_and_transport_delay():
    """out of call k does not depend on cmd of call k (outputs are latched before integrating)."""
    from oracle.dynamics import CitationDynamics
    a, b = CitationDynamics('h2000_v90'), CitationDynamics('h2000_v90')
    x0 = a.step(np.zeros(10))
    np.testing.assert_allclose(x0[[3, 9]], [90.0, 2000.0])
    big = np.zeros(10); big[0] = 0.1
    np.testing.assert_array_equal(b.step(big), x0)
    assert not np.array_equal(a.step(np.zeros(10)), b.step(np.zeros(10)))


def test_reinitialise_restores_identical_image():
    from oracle.dynamics import CitationDynamics
    sim = CitationDynamics('ice')
    cmd = np.zeros(10); cmd[1] = 0.02
    first = np.stack([sim.step(cmd) for _ in range(50)])
    sim.initialize()
    again = np.stack([sim.step(cmd) for _ in range(50)])
    np.testing.assert_array_equal(first, again)


def test_instances_are_independent():
    """The reference allows one simulator per library image (file-scope state); the restatement is re-entrant."""
    from oracle.dynamics import CitationDynamics
    a, b = CitationDynamics('h2000_v90'), CitationDynamics('h2000_v90')
    ca = np.zeros(10); ca[0] = 0.03
    cb = np.zeros(10); cb[1] = -0.03
    xa = [a.step(ca) for _ in range(20)]
    ref = CitationDynamics('h2000_v90')
    for k in range(20):
        b.step(cb)
        np.testing.assert_array_equal(ref.step(ca), xa[k])


@pytest.mark.needs_reference
@pytest.mark.parametrize('build', BUILDS14)
def test_live_reference_library_step_by_step(build):
    from oracle.dynamics import CitationDynamics
    from oracle.refso import RefCitation
    ref, sim = RefCitation(build), CitationDynamics(build)
    rng = np.random.default_rng(hash(build) % 1000)
    nB = len(sim.B)
    for k in range(400):
        cmd = np.zeros(10)
        cmd[:3] = np.deg2rad(rng.uniform(-6, 6, 3)) * (k % 37 != 0)
        if k > 200:
            cmd[8:10] = rng.uniform(0, 0.2)
        xr, xs = ref.step(cmd), sim.step(cmd)
        np.testing.assert_array_equal(xs, xr, err_msg='%s step %d' % (build, k))
        np.testing.assert_array_equal(np.asarray(sim.X), np.asarray(ref.X), err_msg='%s X step %d' % (build, k))
        if k % 50 == 0:
            np.testing.assert_array_equal(np.asarray(sim.B)[:nB].view(np.uint64), np.asarray(ref.B)[:nB].view(np.uint64),
                                          err_msg='%s rtB step %d' % (build, k))
    ref.initialize(); sim.initialize()
    np.testing.assert_array_equal(sim.step(np.zeros(10)), ref.step(np.zeros(10)))
