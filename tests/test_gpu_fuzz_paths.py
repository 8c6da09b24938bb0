"""`Segment.track`'s fast paths against the elements tracked one by one, on drawn lattices (benchmarks/fuzz_paths.py: one-call stretches
with cavities / monitors / apertures / screens, merged runs, non-linear chains; scalar and (B,) settings, vectorised beams and
energies, ParameterBeams; settings edited in place, assigned as new tensors and diagnostics switched between tracks of one Segment;
gradients of losses on particles, beam properties and screen images). The one-by-one path is what the reference-generated goldens
pin (tests/test_gpu_parity.py, test_oracle_diagnostics.py); 30 000 seeds of the script were run in round 6 without a disagreement."""
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
