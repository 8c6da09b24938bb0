"""Parity of what bench.py TIMES, at the shapes it times it (1e6 particles, 100 elements, fp32):

* the headline step — `Segment.track_elementwise(fused=False)` + the global moments — BIT-EXACT against the oracle's fma chain
  (`oracle.track_elementwise`, element.py:180-193 restated) on all 1e6 rows, moments against `oracle.moments` of those rows;
* DKD_FODO100 (both arithmetic widths) and SECOND_ORDER_FODO100 through `Segment.track` (the one-call chains
  `chx_dkd_chain` / `chx_second_order_chain`, i.e. `dkd_kernel`, `second_order_pk_kernel` at the benchmark's launch shape)
  against the oracle's non-linear restatement (oracle/chx_oracle_nonlinear.inc) and against the REFERENCE's own run
  (tests/golden/bench_lattices.npz, generate_golden_bench_lattices.py) on a 4096-particle sample placed in rows [0, 4096) of
  the 1e6-particle beam.

Every bound below is stated next to the error measured on the MI355X of the test box (this file prints them with -s)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 1_000_000
N_CELLS = 25
SAMPLE = 4096


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available()
    import cheetah_amd

    cheetah_amd._lib.lib()
    return cheetah_amd


def _fodo(ca, method=None):
    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    opt = {} if method is None else {"tracking_method": method}
    els = []
    for _ in range(N_CELLS):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2), **opt, **kw), ca.Drift(t(0.8), **opt, **kw),
                ca.Quadrupole(t(0.2), k1=t(-4.2), **opt, **kw), ca.Drift(t(0.8), **opt, **kw)]
    return els


def _bench_beam(ca, g):
    """bench.py's beam (seed 1234, ParticleBeam.from_parameters defaults, 1e6 particles, fp32) with the golden sample in its
    first 4096 rows."""
    torch.manual_seed(1234)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float32, device="cuda")
    x = beam.particles.clone()
    x[:SAMPLE] = torch.from_numpy(g["sample"]).cuda()
    assert float(beam.energy) == float(g["energy"])
    return ca.ParticleBeam(x, beam.energy, particle_charges=beam.particle_charges, species=beam.species)


def _rel(a, b):
    """max |a - b| per coordinate in units of the coordinate's scale max |b|"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return (np.abs(a - b).max(axis=0) / np.abs(b).max(axis=0))[:6]


def test_headline_step_is_bit_exact_against_the_oracle_at_full_size(ca, oracle, golden):
    from cheetah_amd import sharding

    g = golden("bench_lattices.npz")
    seg, beam = ca.Segment(_fodo(ca)), _bench_beam(ca, g)
    out = seg.track_elementwise(beam, fused=False)
    mom = sharding.global_moments(out)
    (kind, run), = seg._plan()
    stack = seg._run_stack(run, beam.energy, beam.species)          # the 100 maps the device applied, (100, 1, 7, 7) fp32
    maps = stack.reshape(-1, 7, 7).cpu().numpy()
    # the device's maps are the oracle's maps of the same float32 settings: fp64 build, one rounding (bound: 2 ulp of the largest
    # entry of a map; measured 0 - 1 ulp)
    E = float(beam.energy)
    f = np.float32
    cell = [oracle.build_rmatrix("quadrupole", [f(0.2), f(4.2), 0, 0, 0], E), oracle.build_rmatrix("drift", [f(0.8)], E),
            oracle.build_rmatrix("quadrupole", [f(0.2), f(-4.2), 0, 0, 0], E), oracle.build_rmatrix("drift", [f(0.8)], E)]
    for i in range(4):
        want = cell[i].reshape(7, 7)
        ulp = np.spacing(np.float32(np.abs(want).max()))
        assert np.abs(maps[i].astype(np.float64) - want).max() <= 2 * ulp, i
        assert np.array_equal(maps[i], maps[i + 4 * 7])              # every cell's maps are the same bits
    x = beam.particles.cpu().numpy()
    ref = oracle.track_elementwise(x, maps)
    got = out.particles.cpu().numpy()
    assert np.array_equal(got, ref), "1e6 x 100 element-by-element fp32 fma chain: device != oracle"
    # the fused in-register variant and the reference-semantics merged track of the same step
    assert torch.equal(seg.track_elementwise(beam, fused=True).particles, out.particles)
    # global moments of the outgoing beam (one-pass shifted fp64 on the device, two-pass fp64 in the oracle): measured 1.0e-16
    # (means, in sigmas) / 5.3e-14 (covariances, relative to sigma_i sigma_j); bounds 4x that
    om = oracle.moments(ref[None])
    dm = mom.cpu().numpy()
    sig = np.sqrt(np.diag(om["cov"][0]))
    e_mu = np.abs(dm[2:8] - om["mu"][0]) / sig
    k, e_cov = 8, 0.0
    for i in range(6):
        for j in range(i, 6):
            e_cov = max(e_cov, abs(dm[k] - om["cov"][0, i, j]) / (sig[i] * sig[j]))
            k += 1
    print(f"\nheadline moments vs oracle: mean {e_mu[:5].max():.2e} sigma, cov {e_cov:.2e}")
    assert dm[0] == om["W"][0] == N
    assert e_mu[:5].max() < 4e-16 and e_cov < 2.2e-13
    # the reference's own fp32 / fp64 runs of the sample rows (its matmul sums the seven products of a row in another order than
    # the fma chain): measured 6.1e-6 / 4.8e-6 of a coordinate's scale — fp32 round-off of 100 steps; the reference's fp32 run is
    # 3.1e-6 away from its own fp64 run. Bounds 4x the measured values.
    e32, e64 = _rel(got[:SAMPLE], g["linear_f32"]), _rel(got[:SAMPLE], g["linear_f64"])
    print(f"headline sample vs reference fp32 {e32.max():.2e}, fp64 {e64.max():.2e}")
    assert e32[:5].max() < 2.4e-5 and e64[:5].max() < 2.0e-5
    assert np.array_equal(got[:, 5], x[:, 5]) and np.all(got[:, 6] == 1.0)


@pytest.mark.parametrize("precision", ["double", "mixed", "storage"])
def test_dkd_fodo100_at_full_size(ca, oracle, golden, precision):
    from cheetah_amd import _ops

    g = golden("bench_lattices.npz")
    els = _fodo(ca, "drift_kick_drift")
    for e in els:
        e.dkd_precision = precision
    seg, beam = ca.Segment(els), _bench_beam(ca, g)
    calls, orig = [], _ops.dkd_chain
    _ops.dkd_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        out = seg.track(beam)
    finally:
        _ops.dkd_chain = orig
    assert calls == [100]                                         # the path the bench times: ONE chx_dkd_chain call
    got = out.particles[:SAMPLE].cpu().numpy()
    # the oracle's Bmad-X restatement (bmadx.py:track_a_drift / track_a_quadrupole; fp64 arithmetic on the fp32 rows, rounded to
    # fp32 after every element — exactly what "double" does), element by element on the sample
    x = g["sample"].copy()
    energy = float(g["energy"])
    m, nq = oracle.ELECTRON_MASS_EV, -1.0
    f = np.float32
    for i in range(100):
        kind = "quadrupole" if i % 2 == 0 else "drift"
        par = [f(0.2), f(4.2 if i % 4 == 0 else -4.2), 0, 0, 0] if kind == "quadrupole" else [f(0.8)]
        o, e_out = oracle.dkd_track(kind, x[None], par, energy, m, nq, 1, 3)
        x = o[0]
    assert float(out.energy) == pytest.approx(energy, rel=0, abs=0)
    e_ora, e64, e32 = _rel(got, x), _rel(got, g["drift_kick_drift_f64"]), _rel(got, g["drift_kick_drift_f32"])
    ref32_vs_64 = _rel(g["drift_kick_drift_f32"], g["drift_kick_drift_f64"])
    print(f"\ndkd[{precision}] vs oracle {e_ora}, vs reference fp64 {e64}, vs reference fp32 {e32}; reference fp32 vs its fp64 {ref32_vs_64}")
    if precision == "double":
        # measured vs the oracle: x, px, y, py, delta bit-identical, tau 8.8e-8 (one fp32 rounding); vs the reference's float64 run
        # of the same rows: 1.2e-6 x, 8.1e-7 px, 6.8e-7 y, 5.3e-7 py, 5.0e-7 tau (100 roundings to the beam's float32 — the
        # reference's own float32 run is 7.6e-6 / 8.0e-3 tau away from that), delta 6.7e-14. Bounds 4x measured.
        assert e_ora[:4].max() == 0.0 and e_ora[5] == 0.0 and e_ora[4] < 3.6e-7
        assert e64[:5].max() < 5e-6 and e64[5] < 3e-13
    elif precision == "mixed":
        # the default of float32 beams: tau and delta as in "double" (5.0e-7, 6.7e-14 against the reference's float64 run), the
        # transverse coordinates 6.7e-6 x, 2.6e-6 px, 4.7e-6 y, 2.3e-6 py after 100 elements (the reference's own float32 run:
        # 7.6e-6 x and 8.0e-3 tau, 3.6e-5 delta). Bounds 4x measured.
        assert e64[:4].max() < 2.7e-5 and e64[4] < 2e-6 and e64[5] < 3e-13
    else:
        # float32 arithmetic like the reference's own tensor code (opt-in): measured against the reference's float64 run 3.5e-5 x,
        # 1.1e-5 px, 1.4e-5 y, 9.7e-6 py, 1.1e-2 tau, 4.2e-4 delta of the coordinate's scale after 100 elements — the reference's own
        # float32 run: 7.6e-6 transverse, 8.0e-3 tau, 3.6e-5 delta. The (tau, delta) <-> (z, pz) conversions cancel in float32 on
        # both sides; "double" (the default) is the accurate variant. Bounds 4x measured.
        assert e64[:4].max() < 1.4e-4 and e64[4] < 4.4e-2 and e64[5] < 1.7e-3
    # the chain call == the elements' own track() one after the other, bit for bit, on all 1e6 rows
    b = beam
    for e in els:
        b = e.track(b)
    assert torch.equal(out.particles, b.particles) and torch.equal(out.s, b.s)


def test_second_order_fodo100_at_full_size(ca, oracle, golden):
    from cheetah_amd import _ops

    g = golden("bench_lattices.npz")
    els = _fodo(ca, "second_order")
    seg, beam = ca.Segment(els), _bench_beam(ca, g)
    calls, orig = [], _ops.second_order_chain
    _ops.second_order_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            out = seg.track(beam)
    finally:
        _ops.second_order_chain = orig
    assert calls == [100]                                         # ONE chx_second_order_chain call (second_order_pk_kernel)
    got = out.particles[:SAMPLE].cpu().numpy()
    # oracle: T of every element (track_methods.py:80-296) in fp64 from the fp32 settings, rounded to the beam dtype like the
    # reference's buffers, x_i' = sum_jk T_ijk x_j x_k in the working dtype (element.py:207-217)
    x = g["sample"].copy()
    energy = float(g["energy"])
    f = np.float32
    Ts = {}
    for i in range(4):
        kind = "quadrupole" if i % 2 == 0 else "drift"
        par = [f(0.2), f(4.2 if i == 0 else -4.2), 0, 0, 0] if kind == "quadrupole" else [f(0.8)]
        Ts[i] = oracle.build_ttensor(kind, par, energy).astype(np.float32).astype(np.float64)
    for i in range(100):
        x = oracle.apply_second_order(x[None], Ts[i % 4])[0]
    e_ora, e64, e32 = _rel(got, x), _rel(got, g["second_order_f64"]), _rel(got, g["second_order_f32"])
    print(f"\nsecond order vs oracle {e_ora}, vs reference fp64 {e64}, vs reference fp32 {e32}")
    # measured: 3.9e-6 vs the oracle (both are fp32 chains of 100 x 28 products, summed in different orders), 4.9e-6 vs the
    # reference's float64 run, 6.2e-6 vs its float32 run (which is 2.7e-6 away from its own float64 run). Bounds 4x measured.
    assert e_ora[:5].max() < 1.6e-5 and e64[:5].max() < 2.0e-5 and e32[:5].max() < 2.5e-5
    assert np.array_equal(got[:, 5], g["sample"][:, 5])
    with torch.no_grad():
        b = beam
        for e in els:
            b = e.track(b)
    assert torch.equal(out.particles, b.particles) and torch.equal(out.s, b.s)
