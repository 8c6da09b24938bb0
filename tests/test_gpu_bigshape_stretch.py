"""The BIG-SHAPE kernels of the lattice stretch against the reference and the oracle. From 1e6 particle rows on the particle pass
takes two particles per lane (`lattice_apply_kernel<T, 2>`: packed float32 FMAs, wave sums at the monitors through DPP row
broadcasts); the reference-generated fixtures for stretches hold 1 200 - 1 500 particles, which select the one-particle-per-lane
kernel. Here the fixtures' beams and settings are PLACED INSIDE big problems (the pattern of tests/golden/bench_lattices.npz):

 * one plain beam of 1.2e6 / 4.3e6 particles through drawn lattices with cavities, monitors and apertures
   (tests/golden/diagnostics_stretch.npz): its first 1 500 rows are the fixture's beam -> those rows of the result against the
   REFERENCE's run, ALL rows, the survival probabilities and every monitor's reading against the oracle's element-by-element walk
   (tests/test_oracle_diagnostics.py `_walk`: chxo_build_rmatrix, x @ R.mT, Cavity.track, bpm.py:77-87, aperture.py:104-128);
 * scans of 64 lattice settings over a shared beam of 70 001 particles with cavities, monitors and apertures, and of 4 096
   settings over 1e5 particles (11.5 GB of result) with monitors and apertures (tests/golden/scan_stretch.npz): rows 0-3 of the
   scan are the fixture's four settings, the beam's first 1 200 particles the fixture's -> that corner against the REFERENCE's
   (4, 1200, 7) result, a handful of whole rows and their readings against the oracle's walk of that row. Scans whose rows start on
   16-byte boundaries take `lattice_scan_wave_kernel` (workgroups walk chunks of rows, particle pairs packed in registers, in float32
   the monitors behind a linear prefix evaluated by moment transport): the 4096 x 1e5 shapes and five shapes of 64 - 1024 rows.
Measured on the MI355X (relative to a coordinate's scale; worst case of the parametrisations): float32 4.5e-7 vs the reference, 1.0e-6
vs the oracle over the rows that survive (5.1e-6 with the rows an aperture took, which fly on to 0.2 m), readings 6.1e-8; float64
5.8e-15 / 1.0e-14 / 3.3e-15 — the bounds are 4 x those or the bounds of the small-shape tests, whichever is larger."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_oracle_diagnostics import _walk

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(ca, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(ca, kind)(**args, **fk)


def _spy_lattice_track(segment, calls):
    host = segment._lib.host()

    class Spy:
        def __getattr__(self, name):
            fn = getattr(host, name)
            return fn if name != "lattice_track" else (lambda *a: (calls.append(len(a)), fn(*a))[1])

    return Spy()


def _big_beam(rng, x0, w0, n):
    """n rows whose first len(x0) are the fixture's: the rest are fixture rows redrawn with 30 % noise (the same phase-space region)."""
    pick = rng.integers(0, len(x0), size=n - len(x0))
    extra = x0[pick] * (1.0 + 0.3 * rng.standard_normal((n - len(x0), 7)))
    extra[:, 6] = 1.0
    w_extra = np.where(rng.random(n - len(x0)) < 0.05, 0.0, rng.random(n - len(x0)))
    return np.concatenate([x0, extra]), np.concatenate([w0, w_extra])


@pytest.mark.parametrize("dt,lattice,n", [(torch.float32, 5, 1_200_003), (torch.float32, 3, 4_300_000), (torch.float64, 5, 1_200_003),
                                          (torch.float32, 7, 1_000_000)])
def test_big_plain_beam_vs_reference_and_oracle(dt, lattice, n, oracle):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    g = np.load(os.path.join(GOLDEN, "diagnostics_stretch.npz"))
    i, f64 = lattice, dt == torch.float64
    fk = {"dtype": dt, "device": "cuda"}
    specs = json.loads(str(g[f"lat{i}_spec"]))
    rng = np.random.default_rng(1000 + i)
    x, w = _big_beam(rng, g[f"lat{i}_in"], g[f"lat{i}_w"], n)
    E0 = float(g[f"lat{i}_energy"])
    seg = ca.Segment([_build(ca, s, fk) for s in specs])
    bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
    beam = ca.ParticleBeam(torch.tensor(x, **fk), torch.tensor(E0, **fk), survival_probabilities=torch.tensor(w, **fk), **fk)
    calls = []
    old = segment._HOST
    segment._HOST = _spy_lattice_track(segment, calls)
    try:
        with torch.no_grad():
            out = seg.track(beam)
    finally:
        segment._HOST = old
    assert len(calls) == 1, calls                              # one stretch call; n >= 1e6: two particles per lane
    got = out.particles.double().cpu().numpy()
    w_got = out.survival_probabilities.double().cpu().numpy()
    # (a) the fixture's rows against the REFERENCE's run
    ref = g[f"lat{i}_out"]
    k = len(ref)
    scale = np.abs(ref).max(axis=0)
    err_ref = (np.abs(got[:k] - ref) / scale).max()
    assert err_ref < (1e-13 if f64 else 3e-6), err_ref
    if f64:
        assert np.array_equal(w_got[:k], g[f"lat{i}_w_out"])
    else:
        assert (np.abs(w_got[:k] - g[f"lat{i}_w_out"]) > 1e-6).sum() <= 4
    assert float(out.energy) == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13 if f64 else 1e-6)
    assert float(out.s) == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-13 if f64 else 1e-6)
    # (b) ALL rows, the survival probabilities and the readings against the oracle's walk (float64 on the beam as the device holds it)
    x_dev = beam.particles.double().cpu().numpy()
    w_dev = beam.survival_probabilities.double().cpu().numpy()
    ox, ow, oE, os_, oread = _walk(oracle, specs, x_dev, w_dev, float(beam.energy))
    alive = ow > 0
    oscale = np.abs(ox[alive]).max(axis=0)
    # (a particle an aperture has taken keeps flying — to amplitudes far beyond the beam's: its error against its own size)
    err_or = (np.abs(got - ox) / np.maximum(oscale, np.abs(ox))).max()
    assert err_or < (1e-12 if f64 else 2e-5), err_or              # (the worst of 1.2e6 - 4.3e6 float32 rows: 5.1e-6 measured)
    differ = int((np.abs(w_got - ow) > 1e-6).sum())
    assert differ <= (0 if f64 else max(8, n // 100_000)), differ          # (a float32 coordinate on the other side of an aperture edge)
    assert float(out.energy) == pytest.approx(oE, rel=1e-13 if f64 else 1e-6) and float(out.s) == pytest.approx(os_, rel=1e-12 if f64 else 1e-6)
    r_got = torch.stack([b.reading for b in bpms]).double().cpu().numpy()
    live = np.isfinite(oread).all(axis=1)
    assert np.array_equal(np.isfinite(r_got).all(axis=1), live)
    size = np.abs(ox[alive][:, [0, 2]]).max()
    err_r = np.abs(r_got[live] - oread[live]).max() / size
    assert err_r < (1e-13 if f64 else 3e-7), err_r
    print(f"big plain beam lattice {i} {dt} n={n}: vs reference {err_ref:.2e}, vs oracle {err_or:.2e}, readings {err_r:.2e}, survival differs {differ}")


def _scan_specs(rng, specs, rows):
    """The fixture's (4,) settings grown to (rows,): rows 0-3 keep the fixture's values, the others are drawn around them."""
    out = []
    for kind, kw in specs:
        new = {}
        for q, v in kw.items():
            if isinstance(v, list) and len(v) == 4 and q in ("k1", "angle"):
                base = np.asarray(v)
                more = base[rng.integers(0, 4, size=rows - 4)] * (1.0 + 0.2 * rng.standard_normal(rows - 4))
                new[q] = [float(t) for t in np.concatenate([base, more])]
            else:
                new[q] = v
        out.append([kind, new])
    return out


def _row_specs(specs, b, rows):
    return [[k, {q: (v[b] if isinstance(v, list) and len(v) == rows and q in ("k1", "angle") else v) for q, v in kw.items()}] for k, kw in specs]


@pytest.mark.parametrize("dt,lattice,rows,n,check_rows", [
    (torch.float32, 3, 64, 70_001, (0, 3, 17, 63)), (torch.float32, 5, 64, 70_001, (1, 40)), (torch.float64, 5, 64, 70_001, (2, 63)),
    (torch.float32, 1, 64, 70_001, (0, 33)), (torch.float32, 4, 4096, 100_000, (0, 3, 2047, 4095)),
    # seven monitors, thousands of rows: the lanes keep their own sums over the tiles of a row (`lattice_apply_kernel<T, 2, 0, 1>`)
    (torch.float32, 0, 4096, 100_000, (0, 1, 2047, 4095)), (torch.float64, 0, 1024, 50_003, (2, 1023)),
    # rows that start on 16-byte boundaries take `lattice_scan_wave_kernel` (a workgroup keeps its particles and walks a chunk of
    # rows, wave-private output staging): cavities + monitors + apertures, a last tile with 1 / 2 / 3 of a wave's 4 x 64 particles,
    # float64 (two particles per lane), a chunk of rows that does not divide the scan
    (torch.float32, 3, 64, 70_000, (0, 3, 17, 63)), (torch.float32, 5, 64, 70_084, (1, 40, 63)), (torch.float64, 5, 64, 70_002, (2, 63)),
    (torch.float32, 1, 67, 70_212, (0, 33, 66)), (torch.float64, 0, 1024, 50_004, (2, 1023))])
def test_big_scans_vs_reference_and_oracle(dt, lattice, rows, n, check_rows, oracle, monkeypatch):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    # (the row-chunk kernel is taken from 1.6e7 particle rows on; the 64-row shapes here ask for it whenever the layout allows)
    monkeypatch.setenv("CHX_TUNE_SCAN_WAVE", "2")
    g = np.load(os.path.join(GOLDEN, "scan_stretch.npz"))
    i, f64 = lattice, dt == torch.float64
    fk = {"dtype": dt, "device": "cuda"}
    rng = np.random.default_rng(2000 + i)
    specs = _scan_specs(rng, json.loads(str(g[f"lat{i}_spec"])), rows)
    x, w = _big_beam(rng, g[f"lat{i}_in"], g[f"lat{i}_w"], n)
    E0 = float(g[f"lat{i}_energy"])
    seg = ca.Segment([_build(ca, s, fk) for s in specs])
    bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
    beam = ca.ParticleBeam(torch.tensor(x, **fk), torch.tensor(E0, **fk), survival_probabilities=torch.tensor(w, **fk), **fk)
    calls = []
    old = segment._HOST
    segment._HOST = _spy_lattice_track(segment, calls)
    try:
        with torch.no_grad():
            out = seg.track(beam)
    finally:
        segment._HOST = old
    assert len(calls) == 1, calls                              # the whole scan is one stretch call (rows x n >= 1e6 particle rows)
    assert tuple(out.particles.shape) == (rows, n, 7)
    # (a) the fixture's corner — its four settings, its 1 200 particles — against the REFERENCE's run
    ref = g[f"lat{i}_out"]
    k = ref.shape[1]
    corner = out.particles[:4, :k].double().cpu().numpy()
    err_ref = (np.abs(corner - ref) / np.abs(ref).max(axis=(0, 1))).max()
    assert err_ref < (1e-13 if f64 else 3e-6), err_ref
    w_ref = g[f"lat{i}_w_out"]
    w_all = out.survival_probabilities
    if w_ref.ndim == 2:
        w_corner = w_all[:4, :k].double().cpu().numpy()
        assert (np.abs(w_corner - w_ref) > 1e-6).sum() <= (0 if f64 else 8)
    assert float(out.energy) == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13 if f64 else 1e-6)
    # (b) whole rows (and their monitors) against the oracle's walk of that row's settings
    x_dev = beam.particles.double().cpu().numpy()
    w_dev = beam.survival_probabilities.double().cpu().numpy()
    err_or = err_r = 0.0
    for b in check_rows:
        ox, ow, oE, os_, oread = _walk(oracle, _row_specs(specs, b, rows), x_dev, w_dev, float(beam.energy))
        got = out.particles[b].double().cpu().numpy()
        alive = ow > 0
        err = (np.abs(got - ox) / np.maximum(np.abs(ox[alive]).max(axis=0), np.abs(ox))).max()
        err_or = max(err_or, err)
        assert err < (1e-12 if f64 else 3e-6), (b, err)
        w_got = (w_all[b] if w_all.dim() == 2 else w_all).double().cpu().numpy()
        assert (np.abs(w_got - ow) > 1e-6).sum() <= (0 if f64 else 8), b
        size = np.abs(ox[alive][:, [0, 2]]).max()
        for m, bpm in enumerate(bpms):
            r = bpm.reading.double().cpu().numpy()
            r = r[b] if r.ndim == 2 else r
            if np.isfinite(oread[m]).all():
                e = np.abs(r - oread[m]).max() / (size + np.abs(oread[m]).max())
                err_r = max(err_r, e)
                assert e < (1e-13 if f64 else 3e-7), (b, m, e)
            else:
                assert not np.isfinite(r).all()
    print(f"big scan lattice {i} {dt} {rows} x {n}: corner vs reference {err_ref:.2e}, rows vs oracle {err_or:.2e}, readings {err_r:.2e}")


def test_transported_monitor_readings_equal_the_particle_sums(monkeypatch):
    """In a big float32 scan a monitor with nothing but maps and monitors in front of it is evaluated by taking the shared beam's weighted
    mean through the row's maps in fp64 (`lattice_scan_bpm_transport_kernel`) instead of summing the tracked particles
    (bpm.py:77-87 either way). Both forms on the same scan: monitors in front of the first aperture are transported, the ones behind it
    summed; the readings agree to the rounding of a float32 reading (measured 1.7e-8 - 3.3e-8 of the beam size; bound 1.2e-7), the
    particles and survival probabilities bit for bit."""
    import cheetah_amd as ca

    fk = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(3)
    B, N = 128, 80_000
    k1 = torch.linspace(-5.0, 5.0, B, **fk)
    els = []
    for i in range(6):
        els += [ca.Quadrupole(t(0.2), k1=k1 if i % 2 == 0 else t(-3.0), misalignment=t([1e-4 * i, -5e-5]), **fk), ca.Drift(t(0.4), **fk),
                ca.BPM(is_active=True, misalignment=t([2e-5, -1e-5 * i]), **fk)]
        if i == 3:
            els.append(ca.Aperture(x_max=t(2e-3), y_max=t(2e-3), shape="elliptical", is_active=True, **fk))
    seg = ca.Segment(els)
    bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
    beam = ca.ParticleBeam.from_parameters(num_particles=N, sigma_x=t(4e-4), sigma_y=t(3e-4), mu_x=t(1e-4), mu_y=t(-2e-4),
                                           energy=t(1e8), **fk)
    w = torch.where(torch.rand(N, device="cuda") < 0.1, torch.zeros(N, **fk), torch.rand(N, **fk))
    beam = ca.ParticleBeam(beam.particles, beam.energy, survival_probabilities=w, **fk)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CHX_TUNE_SCAN_TRANSPORT", mode)
        with torch.no_grad():
            out = seg.track(beam)
        res[mode] = (out.particles.clone(), out.survival_probabilities.clone(), torch.stack([b.reading.clone() for b in bpms]))
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])
    size = float(beam.particles[:, [0, 2]].abs().max())
    err = (res["0"][2].double() - res["1"][2].double()).abs().amax(dim=(1, 2)) / size
    print("transported vs summed readings, per monitor (of the beam size):", [f"{float(e):.1e}" for e in err])
    assert float(err[:4].max()) < 1.2e-7 and float(err[:4].max()) > 0.0       # transported: equal up to the particles' rounding, not identical
    assert float(err[4:].max()) == 0.0                                      # behind the aperture: the particle sums in both runs


def test_scan_with_one_row_of_weights_per_setting(monkeypatch):
    """A shared beam (N, 7) under (B,) settings whose survival probabilities are a (B, N) tensor — every row of the scan its own weights
    (`Bx = 1`, `Bw = B`): the row-chunk kernel reads a row's weights at the head of that row and sums the monitors over the particles
    (no moment transport: there is no single weighted mean of the beam). Against the one-(tile, row)-per-workgroup kernel on the same
    call: particles and survival probabilities bit for bit, readings to the summation order."""
    import cheetah_amd as ca

    fk = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(1)
    B, N = 16, 40_000
    els = [ca.Quadrupole(t(0.2), k1=torch.linspace(-4, 4, B, **fk), **fk), ca.Drift(t(0.5), **fk), ca.BPM(is_active=True, **fk),
           ca.Aperture(x_max=t(1e-3), y_max=t(1e-3), is_active=True, **fk), ca.Drift(t(0.5), **fk), ca.BPM(is_active=True, **fk)]
    seg = ca.Segment(els)
    base = ca.ParticleBeam.from_parameters(num_particles=N, sigma_x=t(4e-4), sigma_y=t(4e-4), energy=t(1e8), **fk)
    beam = ca.ParticleBeam(base.particles, base.energy, survival_probabilities=torch.rand(B, N, **fk), **fk)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CHX_TUNE_SCAN_WAVE", mode)
        with torch.no_grad():
            out = seg.track(beam)
        res[mode] = (out.particles.clone(), out.survival_probabilities.clone(), torch.stack([e.reading for e in seg.elements if isinstance(e, ca.BPM)]))
    assert tuple(res["2"][0].shape) == (B, N, 7) and tuple(res["2"][1].shape) == (B, N)
    assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
    assert float(res["2"][1].min()) == 0.0                                  # the aperture took particles
    size = float(base.particles[:, [0, 2]].abs().max())
    assert float((res["0"][2].double() - res["2"][2].double()).abs().max()) <= 1.2e-7 * size


@pytest.mark.parametrize("what", ["energies", "phases"])
def test_big_cavity_scans_through_both_scan_kernels(what, monkeypatch):
    """A 6-cell linac [Drift, Quadrupole, active Cavity] with a monitor behind every second cell, scanned over 64 beam energies
    (`small_runs` carries the energy-rows bit, not the no-cavity vouch) / over 64 phases of every cavity, one shared beam of 130 000
    particles (8.3e6 particle rows: the row-chunk kernel WITH the cavity epilogue) against the one-(tile, row)-per-workgroup kernel on the
    same call: particles, outgoing energies and readings bit for bit (every monitor sits behind a cavity: particle sums in both)."""
    import cheetah_amd as ca

    fk = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(5)
    B, N = 64, 130_000
    els = []
    for i in range(6):
        ph = torch.linspace(-30.0, 30.0, B, **fk) if what == "phases" else t(-10.0 + i)
        els += [ca.Drift(t(0.3), **fk), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **fk),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=ph, frequency=t(1.3e9), **fk)]
        if i % 2 == 1:
            els.append(ca.BPM(is_active=True, **fk))
    seg = ca.Segment(els)
    energy = torch.linspace(8e7, 1.6e8, B, **fk) if what == "energies" else t(1e8)
    base = ca.ParticleBeam.from_parameters(num_particles=N, sigma_x=t(3e-4), sigma_y=t(3e-4), sigma_tau=t(1e-4), energy=t(1e8), **fk)
    beam = ca.ParticleBeam(base.particles, energy, **fk)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("CHX_TUNE_SCAN_WAVE", mode)
        with torch.no_grad():
            out = seg.track(beam)
        res[mode] = (out.particles.clone(), out.energy.clone(), torch.stack([e.reading for e in seg.elements if isinstance(e, ca.BPM)]))
    assert tuple(res["2"][0].shape) == (B, N, 7)
    assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
    assert float((res["2"][0][0] - res["2"][0][B - 1]).abs().max()) > 0           # the rows differ
    # the cavities act on delta: the rows' outgoing energy spread follows the scan
    assert float(res["2"][1].max() - res["2"][1].min()) > 1e6
    size = float(base.particles[:, [0, 2]].abs().max())
    assert float((res["0"][2].double() - res["2"][2].double()).abs().max()) <= 1.2e-7 * size
