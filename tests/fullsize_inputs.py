"""Seeded synthetic inputs of the full-size BASELINE.json configs (C3 / C4 / C5), produced with numpy only so that the
golden generator (which imports the reference here) and the GPU tests (no reference on the GPU box) build bit-identical
arrays. Values are fp32-representable: the fp64 and fp32 variants of a test track the same numbers.

C4: water-bag ellipsoid, 1 mm radii in x / y, 0.1 mm in tau, cold (sigma_px = sigma_py = sigma_p =
1e-5), 1 nC, 250 MeV; 10 x [Drift 0.1, SpaceChargeKick(0.2, 128^3), Drift 0.1, Quadrupole(0.1, +-4.2), Drift 0.1].
"""
import numpy as np

C4_ENERGY = 2.5e8
C4_TOTAL_CHARGE = -1e-9          # electrons: particle charges are negative (particle_beam.py:418-424 uses species sign)
C4_GRID = (128, 128, 128)
C4_CELLS = 10
C4_SAMPLE_STRIDE = 251           # sampled particles = every 251st (3985 of 1e6)


def c4_particles(n: int = 1_000_000, seed: int = 20260929) -> np.ndarray:
    """(n, 7) float64 with fp32-representable entries."""
    rng = np.random.default_rng(seed)
    r = rng.random(n) ** (1.0 / 3.0)
    cos_t = 2.0 * rng.random(n) - 1.0
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    phi = 2.0 * np.pi * rng.random(n)
    x = np.ones((n, 7))
    x[:, 0] = 1e-3 * r * sin_t * np.cos(phi)
    x[:, 2] = 1e-3 * r * sin_t * np.sin(phi)
    x[:, 4] = 1e-4 * r * cos_t
    x[:, 1] = 1e-5 * rng.standard_normal(n)
    x[:, 3] = 1e-5 * rng.standard_normal(n)
    x[:, 5] = 1e-5 * rng.standard_normal(n)
    return x.astype(np.float32).astype(np.float64)


def c4_charges(n: int = 1_000_000) -> np.ndarray:
    return np.full(n, np.float32(C4_TOTAL_CHARGE / n), dtype=np.float32).astype(np.float64)


def c4_quad_k1(cell: int) -> float:
    return 4.2 if cell % 2 == 0 else -4.2


def gaussian_particles(n: int, seed: int) -> np.ndarray:
    """`ParticleBeam.from_parameters` defaults (particle_beam.py:193-216): sigma_x = sigma_y = 175 um, sigma_px = sigma_py =
    4e-6, sigma_tau = 8e-6, sigma_p = 2e-3, uncorrelated; (n, 7) float64 with fp32-representable entries."""
    rng = np.random.default_rng(seed)
    x = np.ones((n, 7))
    x[:, :6] = rng.standard_normal((n, 6)) * np.array([175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3])
    return x.astype(np.float32).astype(np.float64)


# C3: ARES EA subcell AREASOLA1 -> AREABSCR1 of docs/examples/ARESlatticeStage3v1_9.json
C3_B = 4096
C3_N = 100_000
C3_SAMPLE_ROWS = (0, 1, 511, 1024, 2047, 2048, 3333, 4095)


def c3_k1_scan() -> np.ndarray:
    return np.linspace(-30.0, 30.0, C3_B).astype(np.float32)


def c3_lattice_spec():
    """[(kind, kwargs)] in element order; k1 of AREAMQZM1 is the scanned setting (None)."""
    return [
        ("Marker", {}), ("Drift", {"length": 0.17504}), ("Quadrupole", {"length": 0.122, "k1": None}),
        ("Drift", {"length": 0.428}), ("Quadrupole", {"length": 0.122, "k1": -14.3}), ("Drift", {"length": 0.204}),
        ("VerticalCorrector", {"length": 0.02, "angle": 9e-5}), ("Drift", {"length": 0.204}),
        ("Quadrupole", {"length": 0.122, "k1": 3.142}), ("Drift", {"length": 0.179}),
        ("HorizontalCorrector", {"length": 0.02, "angle": -1e-4}), ("Drift", {"length": 0.45}),
    ]
