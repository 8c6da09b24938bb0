"""LatticeJSON import / export (cheetah/latticejson.py) and the full ARES lattice.

CPU: file written by the reference loads, re-saves to the same elements / lattices, nested segments and
element-valued features (Superimposed) survive a round trip. GPU: the loaded 195-element lattice tracked through
the HIP path reproduces the reference's result (active cavities, solenoid, correctors, CIC screen reading)."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ARES = os.path.join(HERE, "golden", "ares_lattice.json")


def test_reference_file_round_trips(tmp_path):
    import cheetah_amd as ca

    seg = ca.Segment.from_lattice_json(ARES)
    assert len(seg.elements) == 195 and seg.name == "ares"
    assert isinstance(seg.AREAMQZM1, ca.Quadrupole) and isinstance(seg.ARLIRSBL1, ca.Cavity)
    assert seg.AREABSCR1.resolution == (2448, 2040) and seg.AREABSCR1.method == "histogram"
    out = tmp_path / "again.json"
    seg.to_lattice_json(str(out), title="t", info="i")
    a, b = json.load(open(ARES)), json.load(open(out))
    assert b["title"] == "t" and b["info"] == "i" and b["root"] == a["root"]
    assert a["lattices"] == b["lattices"]
    assert a["elements"] == b["elements"]          # every class name, feature and value identical
    again = ca.Segment.from_lattice_json(str(out), dtype=torch.float64)
    assert again.AREAMQZM2.length.dtype == torch.float64
    assert [type(e) for e in again.elements] == [type(e) for e in seg.elements]


def test_nested_segments_and_element_valued_features_round_trip(tmp_path):
    import cheetah_amd as ca

    t = torch.tensor
    inner = ca.Segment([ca.Drift(t(0.5), name="d_in"), ca.Quadrupole(t(0.2), k1=t(1.5), tilt=t(0.1), name="q_in", num_steps=3,
                                                                       tracking_method="drift_kick_drift")], name="inner")
    sup = ca.Superimposed(ca.Quadrupole(t(0.4), k1=t(-2.0), name="q_base"), ca.Marker(name="centre"), name="sup")
    seg = ca.Segment([ca.Drift(t(1.0), name="d0", metadata={"pv": "A:B"}), inner, sup,
                      ca.Screen(resolution=(64, 32), pixel_size=t([1e-4, 2e-4]), name="scr", is_active=True),
                      ca.Aperture(x_max=t(1e-3), shape="elliptical", name="ap"),
                      ca.TransverseDeflectingCavity(t(1.0), voltage=t(1e6), frequency=t(3e9), name="tdc")], name="cell")
    path = tmp_path / "cell.json"
    seg.to_lattice_json(str(path))
    d = json.load(open(path))
    assert d["root"] == "cell" and set(d["lattices"]) == {"cell", "inner"}
    assert d["lattices"]["cell"] == ["d0", "inner", "sup", "scr", "ap", "tdc"]
    assert d["elements"]["sup"][1]["base_element"] == "q_base" and "q_base" in d["elements"]
    assert d["elements"]["d0"][1]["metadata"] == {"pv": "A:B"}
    assert d["elements"]["ap"][1]["y_max"] == float("inf")
    back = ca.Segment.from_lattice_json(str(path))
    assert isinstance(back.inner, ca.Segment) and back.inner.q_in.num_steps == 3
    assert back.inner.q_in.tracking_method == "drift_kick_drift" and float(back.inner.q_in.tilt) == pytest.approx(0.1)
    assert isinstance(back.sup, ca.Superimposed) and float(back.sup.base_element.k1) == -2.0
    assert back.scr.is_active and back.scr.resolution == (64, 32) and back.ap.shape == "elliptical"
    assert back.d0.metadata == {"pv": "A:B"}
    path2 = tmp_path / "cell2.json"
    back.to_lattice_json(str(path2))
    assert json.load(open(path2)) == d


@pytest.mark.gpu
def test_ares_lattice_tracks_like_the_reference(golden):
    import cheetah_amd as ca

    g = golden("ares_track.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    seg = ca.Segment.from_lattice_json(ARES, **kw)
    for name, attr, value in zip(g["names"], g["attrs"], g["values"]):
        setattr(getattr(seg, str(name)), str(attr), torch.tensor(float(value), **kw))
    seg.ARLIRSBL2.phase = torch.tensor(-10.0, **kw)
    seg.AREABSCR1.is_active = True
    seg.AREABSCR1.method = "cloud-in-cell"
    beam = ca.ParticleBeam(torch.tensor(g["incoming"], **kw), torch.tensor(g["energy_in"], **kw),
                           particle_charges=torch.tensor(g["charges"], **kw), species=ca.Species("electron", **kw))
    out = seg.track(beam)
    assert float(out.energy) == pytest.approx(float(g["energy_out"]), rel=1e-14)
    assert float(out.s) == pytest.approx(float(g["s_out"]), rel=1e-13)
    assert float(seg.length) == pytest.approx(float(g["length"]), rel=1e-13)
    got, exp = out.particles.cpu().numpy(), g["outgoing"]
    for j in range(6):
        assert np.max(np.abs(got[:, j] - exp[:, j])) <= 1e-10 * np.max(np.abs(exp[:, j])), j
    img = seg.AREABSCR1.reading.cpu().numpy()
    assert tuple(img.shape) == tuple(g["screen_shape"])
    ref = np.zeros_like(img)
    ref[tuple(g["screen_idx"].T)] = g["screen_val"]
    assert np.allclose(img, ref, rtol=1e-7, atol=1e-9 * ref.max())
    assert np.count_nonzero(img) == len(g["screen_val"])
