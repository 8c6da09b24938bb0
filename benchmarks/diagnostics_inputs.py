"""The lattices and beams of bench.py's `DIAGNOSTICS_LATTICES` entries as DATA, drawn on the host from fixed seeds — so that the
engine (bench.py, float32 on the GPU) and the reference (tests/golden/generate_golden_bench_diagnostics.py, float64 on the CPU of
the build container) track the same numbers and bench.py can assert what it times against the reference's results
(tests/golden/bench_diagnostics.json), like the headline asserts its sigma_x.

Everything here is plain torch on the CPU: no cheetah / cheetah_amd import. A lattice is a list of [kind, {setting: value}] with
floats, lists of floats (vector settings) and strings; `build(module, spec, fk)` turns one entry into an element of either package."""
import torch

N_BEAM = 100_000
N_SMALL = 10_000
ROWS = 64


def particles(n: int = N_BEAM, seed: int = 4321) -> torch.Tensor:
    """(n, 7) float32 rows of a Gaussian beam (sigma_x = sigma_y = 175 um, sigma_px = sigma_py = 4e-6, sigma_tau = 8e-6,
    sigma_p = 2e-3: the defaults of particle_beam.py:193-216) from the CPU generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 7, generator=g, dtype=torch.float64)
    x *= torch.tensor([175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3, 0.0], dtype=torch.float64)
    x[:, 6] = 1.0
    return x.to(torch.float32)


def corrector_angles(cells: int = 25, rows: int = ROWS, seed: int = 77) -> list:
    g = torch.Generator().manual_seed(seed)
    return [[float(v) for v in 1e-5 * torch.randn(rows, generator=g, dtype=torch.float64).to(torch.float32)] for _ in range(cells)]


def quad(i: int, k: float = 4.2) -> list:
    return ["Quadrupole", {"length": 0.2, "k1": k if i % 2 == 0 else -k}]


def bpm_lattice() -> list:
    out = []
    for i in range(25):
        out += [quad(i), ["Drift", {"length": 0.8}], ["BPM", {"is_active": True}], ["Drift", {"length": 0.2}]]
    return out


def aperture_lattice() -> list:
    out = []
    for i in range(25):
        out += [quad(i), ["Drift", {"length": 0.8}], ["Aperture", {"x_max": 5e-3, "y_max": 5e-3, "is_active": True}], ["Drift", {"length": 0.2}]]
    return out


def cavity_linac(phase=-10.0, off=()) -> list:
    """16 cells [Drift, Quadrupole, Cavity]; `phase`: a float or a list of ROWS floats (the phase of EVERY cavity scanned)."""
    out = []
    for i in range(16):
        out += [["Drift", {"length": 0.3}], ["Quadrupole", {"length": 0.2, "k1": 3.0 if i % 2 else -3.0}],
                ["Cavity", {"length": 1.0377, "voltage": 0.0 if i in off else 18e6, "phase": phase, "frequency": 1.3e9}]]
    return out


def orbit_response() -> list:
    angles = corrector_angles()
    out = []
    for i in range(25):
        out += [quad(i), ["HorizontalCorrector", {"length": 0.05, "angle": angles[i]}], ["Drift", {"length": 0.8}],
                ["BPM", {"is_active": True}]]
    return out


def phases() -> list:
    return [float(v) for v in torch.linspace(-30.0, 30.0, ROWS, dtype=torch.float32)]


def energies() -> list:
    return [float(v) for v in torch.linspace(8e7, 1.2e8, ROWS, dtype=torch.float32)]


def build(module, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, **fk)


def segment(module, specs, fk):
    return module.Segment([build(module, s, fk) for s in specs])


def parameter_beam_moments():
    """mu (7,), cov (7, 7) float64 of the ParameterBeam bench.py tracks (the beam defaults above)."""
    mu = torch.zeros(7, dtype=torch.float64)
    mu[6] = 1.0
    cov = torch.diag(torch.tensor([175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3, 0.0], dtype=torch.float64) ** 2)
    return mu, cov
