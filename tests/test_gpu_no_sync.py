"""No hidden host synchronisation on the tracking path: after two warm-up calls (which fill the host-side caches: is_active
flags, species scalars, screen geometry), the steady-state calls below must not make torch synchronise with the device — no
`.item()`, no host read, no pageable host-to-device copy (`torch.tensor(x, device="cuda")`). Checked with
torch.cuda.set_sync_debug_mode("warn"), which is calibrated first on two calls that do synchronise. include/chx.h promises
the same for the library itself (launch only); this covers the Python layer above it."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def sync_warnings(fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            fn()
        finally:
            torch.cuda.set_sync_debug_mode("default")
    return [w for w in rec if "synchronizing" in str(w.message).lower() and "prototype" not in str(w.message).lower()]


def test_steady_state_tracking_calls_do_not_synchronise():
    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    beam = ca.ParticleBeam.from_parameters(num_particles=50_000, **kw)
    # the switch sees what it should see
    assert len(sync_warnings(lambda: torch.tensor(1.0, device="cuda"), warm=0)) == 1
    assert len(sync_warnings(lambda: float(beam.sigma_x), warm=0)) == 1

    screen = ca.Screen(resolution=(200, 100), pixel_size=t([1e-5, 1e-5]), is_active=True, name="screen", **kw)
    quad = ca.Quadrupole(t(0.2), k1=t(3.0), name="quad", **kw)
    seg = ca.Segment([ca.Drift(t(0.5), **kw), quad, ca.HorizontalCorrector(t(0.1), angle=t(1e-4), **kw), ca.Marker(**kw),
                      ca.Cavity(t(1.0), voltage=t(1e7), phase=t(5.0), frequency=t(1.3e9), **kw), ca.Drift(t(0.5), **kw),
                      ca.Aperture(x_max=t(5e-3), y_max=t(5e-3), is_active=True, **kw), ca.BPM(is_active=True, **kw), screen])
    values = [t(2.0), t(-2.0)]
    flip = [0]

    def control_step():
        flip[0] ^= 1
        quad.k1 = values[flip[0]]
        out = seg.track(beam)
        return screen.reading, out.sigma_x, out.mu_y

    flows = {
        "Segment.track": lambda: seg.track(beam),
        "control step": control_step,
        "beam.clone": lambda: beam.clone(),
    }
    kick_seg = ca.Segment([ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.1), **kw)])
    flows["space-charge kick"] = lambda: kick_seg.track(beam)
    pb = ca.ParameterBeam.from_parameters(**kw)
    flows["ParameterBeam step"] = lambda: (seg.track(pb), screen.reading)
    for el in (ca.Marker(**kw), ca.BPM(is_active=True, **kw), ca.Aperture(x_max=t(1e-3), is_active=True, **kw),
               ca.Screen(is_active=True, **kw), ca.Dipole(t(0.5), angle=t(0.1), tracking_method="drift_kick_drift", **kw),
               ca.Quadrupole(t(0.2), k1=t(3.0), tracking_method="second_order", **kw)):
        flows[f"{type(el).__name__}.track"] = lambda el=el: el.track(beam)
    k1 = torch.nn.Parameter(t(3.0))
    seg5 = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(1.0), **kw)])

    def forward_backward():
        k1.grad = None
        seg5.track(beam).sigma_x.backward()

    flows["forward + backward"] = forward_backward
    vec = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=torch.linspace(-3, 3, 8, **kw), **kw), ca.Drift(t(1.0), **kw)])
    flows["vectorised track"] = lambda: vec.track(beam)
    flows["track_moments"] = lambda: vec.track_moments(beam)
    # the chain of tile-ordered space-charge kicks incl. its guard (pinned copy of the chain's header + event poll, first tracks
    # of a plan) and a run of drift-kick-drift elements in one call
    big = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
    chain = ca.Segment([el for i in range(3) for el in (ca.SpaceChargeKick(t(0.1), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.2), **kw),
                                                        ca.Quadrupole(t(0.1), k1=t(2.0 if i % 2 else -2.0), **kw))])
    flows["space-charge chain"] = lambda: chain.track(big)
    dkd = ca.Segment([ca.Drift(t(0.2), tracking_method="drift_kick_drift", **kw),
                      ca.Quadrupole(t(0.1), k1=t(2.0), tracking_method="drift_kick_drift", **kw),
                      ca.Drift(t(0.2), tracking_method="drift_kick_drift", **kw)])
    flows["drift-kick-drift run"] = lambda: dkd.track(beam)
    for name, fn in flows.items():
        hits = sync_warnings(fn)
        assert not hits, (name, [(w.filename, w.lineno) for w in hits])
