"""The plan machinery under its own debug switch. `CHX_CHECK_PLANS=1` makes every track re-derive every address a persistent plan
holds (`_FastRun.verify`, `_LatticePlan.verify`) and raise on a mismatch; the suites that exercise the plans — merged runs, lattice
stretches (monitors, apertures, cavities, screens), graph capture, the drift-kick-drift fast path — must pass UNCHANGED under it:
the plans hold what the elements hold, after every kind of edit those suites make (in-place writes, assignments, toggled cavities,
copies, pickles). Plus: a storage swapped behind the host's back is found. (segment.py:545-574 / utils/cache.py:6-68 of the
reference are the contract the plans implement.)"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_suites_pass_with_every_address_rederived_per_track():
    env = dict(os.environ, CHX_CHECK_PLANS="1")
    files = ["tests/test_gpu_fast_run.py", "tests/test_gpu_lattice_stretch.py", "tests/test_gpu_graph_capture.py",
             "tests/test_gpu_dkd_fast_path.py", "tests/test_gpu_screen_stretch.py"]
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *files], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " passed" in res.stdout and "failed" not in res.stdout


def test_swapped_storage_is_found():
    code = '''
import torch, cheetah_amd as ca
kw = {"dtype": torch.float32, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
quad = ca.Quadrupole(t(0.2), k1=t(4.2), **kw)
seg = ca.Segment([ca.Drift(t(0.5), **kw), quad, ca.Drift(t(0.5), **kw), ca.Screen(resolution=(32, 32), is_active=True, **kw)])
beam = ca.ParticleBeam.from_parameters(num_particles=1000, **kw)
with torch.no_grad():
    seg.track(beam)
    quad.k1.data = t(-3.0)          # no attribute assignment, no in-place write: the plans still hold the old address
    try:
        seg.track(beam)
    except RuntimeError as exc:
        print("FOUND", str(exc)[:60])
'''
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, CHX_CHECK_PLANS="1"), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "FOUND" in res.stdout, res.stdout + res.stderr[-500:]
