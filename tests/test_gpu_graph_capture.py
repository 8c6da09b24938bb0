"""Forward + backward of config C5 captured into a device graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: libchx's
launches are plain launches on the current stream, its workspaces come from torch's allocator and nothing on the path synchronises,
so a whole optimisation step can be captured; the replay follows in-place updates of the trainable setting (the kernels read it
through its pointer) and gives the eager numbers (/root/reference/tests/test_differentiable.py:10-32 is the eager step). Run in a
process of its own: the capture must come before any backward pass on the default stream touches the Parameter."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_step_replays_from_a_graph_with_the_eager_numbers():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "c5_graph.py")], capture_output=True, text=True, timeout=600,
                          cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"c5_graph"')]
    assert line, proc.stdout[-1000:]
    g = json.loads(line[-1])["c5_graph"]
    # after k1 was set to 2.5 in place: the replayed step and the eager step agree
    assert g["loss"] == pytest.approx(g["loss_eager"], rel=1e-6) and g["grad"] == pytest.approx(g["grad_eager"], rel=1e-5)
    assert g["loss"] == pytest.approx(8.05307e-05, rel=1e-3)                 # sigma_x at the screen for k1 = 2.5
    assert g["graph_replay_us"] < g["eager_us"]


@pytest.mark.parametrize("which", ["c1", "control", "control_parameter_beam", "linac"])
def test_small_beam_steps_replay_from_a_graph(which):
    """Track + screen reading of the README segment, and the control step whose five settings are written in place: captured,
    replayed, equal to the eager step; the replay follows new settings."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "graph_modes.py"), which], capture_output=True, text=True,
                          timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    g = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"graph_mode"')][-1])["graph_mode"]
    assert g["replay_equals_eager"] is True
    if which.startswith("control"):
        assert g["replay_follows_in_place_settings"] is True
    if os.environ.get("CHX_CHECK_PLANS") == "1":
        return        # (every address re-derived per track: tests/test_gpu_check_plans.py checks the results, not the clock)
    if which == "linac":
        # the eager step of this lattice is ONE chx_lattice_track call (two launches; Segment._lattice_stretch): 40 us against the
        # 600 us of the element-by-element walk — a replayed graph (44 us) has nothing left to win
        assert g["eager_us"] < 120 and g["graph_replay_us"] < 120
    else:
        # round 5: the eager step is ONE stretch call as well (active Screens are items of it: two launches, the C++ host step) —
        # 27 / 53 / 57 us where round 4 measured 70 / 127 / 152, about what the replay of its graph costs (31 / 35 / 47 us)
        assert g["eager_us"] < 90 and g["graph_replay_us"] < 90


def test_replay_follows_settings_changed_outside_the_step_on_cached_paths():
    """Paths whose maps are cached on the host against version counters (a vectorised setting, elements tracked on their own, a
    drift-kick-drift element's parameter array): while `capture` records, those caches are off, so the kernels that derive the
    maps from the settings are part of the graph and a replay sees a setting that was edited in place BETWEEN replays."""
    import torch

    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(0)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=t(2e-4), sigma_px=t(2e-5), energy=t(1e8), **kw)
    k1 = torch.linspace(-3.0, 3.0, 4, **kw)
    quad = ca.Quadrupole(t(0.2), k1=k1, **kw)                               # vectorised: no persistent device plan
    lone = ca.Quadrupole(t(0.3), k1=t(1.5), **kw)                           # tracked on its own: Element._cached_map
    dkd = ca.Quadrupole(t(0.2), k1=t(2.0), tracking_method="drift_kick_drift", **kw)
    seg = ca.Segment([ca.Drift(t(0.5), **kw), quad, ca.Drift(t(0.5), **kw)])

    def step():
        return seg.track(beam).sigma_x, lone.track(beam).sigma_x, dkd.track(beam).sigma_x

    with torch.no_grad():
        for _ in range(3):
            step()                                                          # the host caches are warm ...
        captured = ca.graph.capture(step)                                   # ... and must not be used by the recording
        first = [v.clone() for v in captured()]
        quad.k1.mul_(-1.0)
        lone.k1.fill_(-4.0)
        dkd.k1.fill_(-6.0)
        replayed = [v.clone() for v in captured()]
        eager = step()
    for a, b, c in zip(replayed, eager, first):
        assert torch.allclose(a, b, rtol=1e-6), (a, b)
        assert not torch.allclose(a, c, rtol=1e-3), "the replay did not follow the edited setting"


def test_space_charge_chain_is_capturable():
    """The chain of tile-ordered kicks with its side stream (fork / join events become graph edges), the device-side re-ordering
    decision and the host guard's pinned copy: recorded and replayed with the eager result. (The replay is no faster than the
    eager track on this platform — bench.py does not use it — but a captured optimisation step may contain such a lattice.)"""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "graph_modes.py"), "c4"], capture_output=True, text=True,
                          timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    g = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{"graph_mode"')][-1])["graph_mode"]
    assert g["replay_equals_eager"] is True


def test_chain_capture_with_a_single_warm_up_track():
    """The chain's host guard samples the first tracks of a plan (pinned copy + event): a recording that starts before the
    sampling is over must not put that event into the graph."""
    import torch

    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(1)
    beam = ca.ParticleBeam.from_parameters(num_particles=120_000, sigma_x=t(3e-4), sigma_y=t(2e-4), sigma_tau=t(1e-4), energy=t(5e7),
                                           total_charge=t(1e-9), **kw)
    seg = ca.Segment([el for i in range(3) for el in (ca.SpaceChargeKick(t(0.1), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.2), **kw),
                                                      ca.Quadrupole(t(0.1), k1=t(2.0 if i % 2 else -2.0), **kw))])
    with torch.no_grad():
        step = ca.graph.capture(lambda: seg.track(beam).particles, warmup=1)
        replayed = step().clone()
        eager = seg.track(beam).particles
    scale = eager.abs().max(dim=0).values
    assert torch.all((replayed - eager).abs().max(dim=0).values <= 1e-5 * scale + 1e-12)
    assert torch.isfinite(replayed).all()


def test_a_lattice_of_many_element_types_is_capturable():
    """Drift, quadrupole, corrector, marker, active cavity, aperture, BPM, a second-order quadrupole, a drift-kick-drift dipole and
    a screen in one Segment: recorded once; the replay follows in-place changes of a quadrupole strength and of the cavity phase
    and gives the eager step's image, beam moments, BPM reading and energy."""
    import torch

    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)
    beam = ca.ParticleBeam.from_parameters(num_particles=50_000, energy=t(1e8), **kw)
    screen = ca.Screen(resolution=(200, 100), pixel_size=t([1e-5, 1e-5]), is_active=True, name="screen", **kw)
    quad = ca.Quadrupole(t(0.2), k1=t(3.0), name="quad", **kw)
    cav = ca.Cavity(t(1.0), voltage=t(1e7), phase=t(5.0), frequency=t(1.3e9), **kw)
    seg = ca.Segment([ca.Drift(t(0.5), **kw), quad, ca.HorizontalCorrector(t(0.1), angle=t(1e-4), **kw), ca.Marker(**kw), cav, ca.Drift(t(0.5), **kw),
                      ca.Aperture(x_max=t(5e-3), y_max=t(5e-3), is_active=True, **kw), ca.BPM(is_active=True, name="bpm", **kw),
                      ca.Quadrupole(t(0.2), k1=t(1.0), tracking_method="second_order", **kw),
                      ca.Dipole(t(0.5), angle=t(0.02), tracking_method="drift_kick_drift", **kw), screen])
    def step():
        out = seg.track(beam)
        return screen.reading, out.sigma_x, out.mu_y, seg.bpm.reading, out.energy
    with torch.no_grad():
        cap = ca.graph.capture(step)
        a = [v.clone() for v in cap()]
        quad.k1.fill_(-2.0); cav.phase.fill_(20.0)
        b = [v.clone() for v in cap()]
        c = step()
    names = ("image", "sigma_x", "mu_y", "bpm", "energy")
    for name, x, y, z in zip(names, b, c, a):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7 * float(y.abs().max())), name
        if name in ("sigma_x", "energy"):
            assert float((x - z).abs().max()) > 0, name          # the replay saw the new settings


@pytest.mark.parametrize("method", ["second_order", "drift_kick_drift"])
def test_a_stretch_of_non_linear_elements_and_linear_runs_is_capturable(method):
    """Linear drifts between second-order / drift-kick-drift magnets go through the device in one pass (Segment._second_order_run /
    _dkd_run; the maps of the linear runs are kept while nothing changed). While `capture` records, nothing is kept: the launches
    that derive every map from the settings are part of the graph, and a replay follows a magnet's strength, a linear element's
    setting and the beam energy edited in place between replays — same numbers as the eager step."""
    import torch

    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(5)
    nl = {"tracking_method": method}
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, energy=t(5e7), **kw)
    lin_quad = ca.Quadrupole(t(0.1), k1=t(0.5), **kw)
    magnet = ca.Quadrupole(t(0.2), k1=t(3.0), **nl, **kw)
    seg = ca.Segment([ca.Drift(t(0.5), **kw), lin_quad, magnet, ca.Drift(t(0.4), **kw), ca.Marker(**kw),
                      ca.Quadrupole(t(0.2), k1=t(-2.5), **nl, **kw), ca.Drift(t(0.3), **kw)])

    def step():
        out = seg.track(beam)
        return out.particles, out.energy, out.s

    with torch.no_grad():
        eager0 = [v.clone() for v in step()]
        cap = ca.graph.capture(step)
        a = [v.clone() for v in cap()]
        for x, y in zip(a, eager0):
            assert torch.equal(x, y)
        magnet.k1.fill_(-1.0); lin_quad.k1.fill_(2.0); beam.energy.mul_(1.2)
        b = [v.clone() for v in cap()]
        c = step()
    for x, y in zip(b, c):
        assert torch.equal(x, y)
    assert not torch.equal(a[0], b[0])


@pytest.mark.parametrize("beam_kind", ["particles", "parameters"])
def test_a_lattice_with_monitors_apertures_and_cavities_replays_from_a_graph(beam_kind):
    """The diagnostics stretch (chx_lattice_track_diag / chx_parameter_lattice_track) inside a captured step: the replay equals the
    eager step, follows correctors written in place, and the monitors' reading tensors are the capture's static tensors."""
    import torch

    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els, bpms, cors = [], [], []
    for i in range(6):
        bpm = ca.BPM(is_active=True, **kw)
        cor = ca.HorizontalCorrector(t(0.05), angle=t(1e-5 * i), **kw)
        bpms.append(bpm)
        cors.append(cor)
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), cor, ca.Drift(t(0.5), **kw), bpm]
        if i == 2 and beam_kind == "particles":
            els += [ca.Aperture(x_max=t(8e-4), y_max=t(8e-4), **kw)]
        if i % 3 == 1:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    seg = ca.Segment(els)
    if beam_kind == "particles":
        beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(6e7), sigma_x=t(3e-4), **kw)
        state = lambda b: b.particles  # noqa: E731
    else:
        beam = ca.ParameterBeam.from_parameters(energy=t(6e7), sigma_x=t(3e-4), **kw)
        state = lambda b: b.mu  # noqa: E731

    def step():
        out = seg.track(beam)
        return torch.cat([state(out).reshape(-1)[:70], torch.stack([b.reading for b in bpms]).reshape(-1)])

    with torch.no_grad():
        eager = step().clone()
        captured = ca.graph.capture(step)
        assert torch.equal(captured().clone(), eager)
        for c in cors:
            c.angle.mul_(-3.0).add_(2e-5)
        replayed = captured().clone()
        now = step().clone()
        assert torch.equal(replayed, now) and not torch.equal(replayed, eager)


@pytest.mark.parametrize("beam_kind", ["particles", "parameters"])
def test_a_grid_scan_by_broadcasting_replays_from_a_graph(beam_kind):
    """A grid scan whose settings have shapes (3, 1) and (1, 4): the stretch reads them through expanded copies. Inside a recording
    the copies are made as nodes of the graph, so a replay follows strengths written in place BETWEEN two replays, like every other
    setting."""
    import torch

    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    ka = torch.linspace(-3.0, 3.0, 3, **kw).reshape(3, 1).contiguous()
    kb = torch.linspace(-2.0, 2.0, 4, **kw).reshape(1, 4).contiguous()
    bpms = [ca.BPM(is_active=True, **kw) for _ in range(3)]
    seg = ca.Segment([ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=ka, **kw), ca.Drift(t(0.5), **kw), bpms[0],
                      ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw), bpms[1],
                      ca.Quadrupole(t(0.2), k1=kb, **kw), ca.Drift(t(0.4), **kw), bpms[2]])
    if beam_kind == "particles":
        beam = ca.ParticleBeam.from_parameters(num_particles=5_000, energy=t(6e7), mu_x=t(1e-4), **kw)
        state = lambda b: b.particles  # noqa: E731
    else:
        beam = ca.ParameterBeam.from_parameters(energy=t(6e7), mu_x=t(1e-4), **kw)
        state = lambda b: b.mu  # noqa: E731

    def step():
        out = seg.track(beam)
        return torch.cat([state(out).reshape(-1)[:140], torch.cat([b.reading.reshape(-1) for b in bpms])])

    with torch.no_grad():
        eager = step().clone()
        captured = ca.graph.capture(step)
        assert torch.equal(captured().clone(), eager)
        ka.mul_(-1.7)
        kb.add_(0.3)
        replayed = captured().clone()
        now = step().clone()
        assert torch.equal(replayed, now) and not torch.equal(replayed, eager)
