"""`Segment.track`'s fast paths against the elements tracked one by one, on drawn lattices (benchmarks/fuzz_paths.py: one-call stretches
with cavities / monitors / apertures / screens, merged runs, non-linear chains; scalar and (B,) settings, vectorised beams and
energies, ParameterBeams; settings edited in place, assigned as new tensors and diagnostics switched between tracks of one Segment;
gradients of losses on particles, beam properties and screen images). The one-by-one path is what the reference-generated goldens
pin (tests/test_gpu_parity.py, test_oracle_diagnostics.py); 190 000 seeds of the script were run in round 6: every disagreement
was a float32 rounding flip at a bin / screen / aperture edge or a statistic of one or two surviving particles."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))


@pytest.mark.parametrize("first", [0, 1000, 2000, 3000])
def test_drawn_lattices_fast_paths_equal_the_walk(first):
    import warnings

    import fuzz_paths

    bad = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in range(first, first + 120):
            specs, dt, B, n, fails = fuzz_paths.one_case(seed)
            if fails:
                bad.append((seed, str(dt), B, n, fails, [k for k, _ in specs]))
    assert not bad, bad


def test_drawn_space_charge_lattices_chain_equals_the_walk():
    """benchmarks/fuzz_sc_chain.py: [linear run, SpaceChargeKick]+ lattices — beams that keep their tile order and beams that go through
    a focus between kicks — through `Segment.track` (the tile-ordered chain) and kick by kick (11 600 seeds agreed in round 6)."""
    import warnings

    import fuzz_sc_chain

    bad, chained = [], 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in range(60):
            info, dt, n, fails = fuzz_sc_chain.one_case(seed)
            chained += info[3] > 0
            if fails:
                bad.append((seed, str(dt), n, info, fails))
    assert not bad, bad
    assert chained >= 20          # (the sweep is about the chain: most of its cases must take it)
