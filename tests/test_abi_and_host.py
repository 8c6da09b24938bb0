"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/chx.h declares;
host-side logic (segment partitioning, caches, broadcasting helpers, fail-loud behaviour)."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "chx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    import cheetah_amd._lib as L

    lib = L.lib()  # raises if libchx.so is missing or a declared symbol cannot be bound
    syms = declared_symbols()
    assert len(syms) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    exported = set(re.findall(r" T (chx_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert set(syms) == set(L.SIGNATURES), (sorted(set(syms) ^ set(L.SIGNATURES)))
    assert lib.chx_abi_version() == 9
    assert lib.chx_kind_num_params(2) == 5 and lib.chx_kind_num_params(3) == 9 and lib.chx_kind_num_params(99) == -1
    assert lib.chx_status_string(-3) == b"misaligned buffer"
    # pure host-side queries (no device needed)
    assert lib.chx_moments_workspace_bytes(1, 1_000_000) > 0
    assert lib.chx_apply_bwd_workspace_bytes(4, 1000) > 0


def test_library_targets_gfx950_only():
    import cheetah_amd._lib as L

    blob = open(L.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_has_no_cpu_fallback():
    """Tracking a CPU beam must fail loudly instead of silently running elsewhere."""
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(torch.tensor(1.0)), ca.Quadrupole(torch.tensor(0.2), k1=torch.tensor(4.2))])
    beam = ca.ParticleBeam.from_parameters(num_particles=100)
    with pytest.raises(RuntimeError, match="GPU only"):
        seg.track(beam)
    with pytest.raises(RuntimeError, match="GPU only"):
        _ = beam.sigma_x


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cheetah_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), (dirpath, f)
                assert "chx_oracle" not in src and "chxo_" not in src and "libchx_oracle" not in src, (dirpath, f)


def test_segment_partitioning_and_names():
    import cheetah_amd as ca

    t = torch.tensor
    seg = ca.Segment([
        ca.Drift(t(1.0)), ca.Quadrupole(t(0.2), k1=t(1.0), name="q"), ca.Cavity(t(1.0), voltage=t(1e6), name="cav"),
        ca.Drift(t(0.5), name="d"), ca.Screen(is_active=True, name="scr"), ca.Drift(t(0.5), name="d"), ca.Marker(name="m"),
    ])
    assert not seg.is_skippable
    assert seg.q.k1.item() == 1.0
    assert isinstance(seg.d, list) and len(seg.d) == 2
    assert seg.cav.is_active and not seg.cav.is_skippable
    assert seg.scr.is_active and not seg.scr.is_skippable
    seg.scr.is_active = False
    assert seg.scr.is_skippable
    seg.cav.voltage = t(0.0)
    assert seg.cav.is_skippable and seg.is_skippable
    assert float(seg.length) == pytest.approx(3.2)
    assert ca.Segment([ca.Drift(t(1.0))]).first_order_transfer_map is not None


def test_revision_and_feature_keys_track_changes():
    import cheetah_amd as ca
    from cheetah_amd._cache import TensorKey

    q = ca.Quadrupole(torch.tensor(0.2), k1=torch.tensor(4.2))
    r0 = q.__dict__["_revision"]
    static0, tensors0 = q._feature_key()
    key0 = TensorKey(tensors0)
    assert key0.matches(q._feature_key()[1])
    q.k1 = torch.tensor(1.0)
    assert q.__dict__["_revision"] > r0 and not key0.matches(q._feature_key()[1])
    key1 = TensorKey(q._feature_key()[1])
    q.k1.add_(1.0)
    assert not key1.matches(q._feature_key()[1])            # in-place edit: _version moved
    q.k1 = torch.nn.Parameter(torch.tensor(2.0))
    assert any(t.requires_grad for t in q._feature_key()[1])
    assert [n for n, _ in q.named_parameters()] == ["k1"]
    key2 = TensorKey(q._feature_key()[1])
    q.k1.requires_grad_(False)
    assert not key2.matches(q._feature_key()[1])            # requires_grad is part of the key
    assert q._feature_key()[0] == static0                   # the non-tensor features did not change


def test_caches_are_not_fooled_by_recycled_tensor_ids():
    """caches keyed on id(tensor) served stale entries when CPython reused the id of a freed tensor
    (Cavity.is_active stale in 49 of 200 trials). The keys now hold the tensors and compare with `is`."""
    import cheetah_amd as ca

    stale = 0
    for _ in range(300):
        c = ca.Cavity(torch.tensor(1.0), voltage=torch.tensor(0.0), phase=torch.tensor(0.0), frequency=torch.tensor(1.3e9))
        assert not c.is_active
        c.voltage = torch.tensor(1e6)   # replaces (and frees) the tensor the flag was cached for
        c.voltage = torch.tensor(2e6)   # may receive the freed tensor's id
        stale += not c.is_active
    assert stale == 0
    # the map cache key: same hazard on defining tensors
    from cheetah_amd._cache import TensorKey

    q = ca.Quadrupole(torch.tensor(0.2), k1=torch.tensor(4.2))
    for _ in range(300):
        key = TensorKey(q._feature_key()[1])
        q.k1 = torch.tensor(1.0)
        q.k1 = torch.tensor(2.0)
        assert not key.matches(q._feature_key()[1])
    # a Segment run's token records requires_grad of buffers too
    from cheetah_amd.accelerator.segment import _Run

    run = _Run([q])
    e = torch.tensor(1e8)
    sp = ca.Species("electron")
    tok = run.current_token(e, sp)
    q.k1.requires_grad_(True)
    assert run.current_token(e, sp) != tok


def test_rbend_edge_setters():
    """rbend.py:107-117: assigning rbend_e1 / rbend_e2 updates the dipole pole-face angles."""
    import cheetah_amd as ca

    r = ca.RBend(torch.tensor(1.0), angle=torch.tensor(0.2), rbend_e1=torch.tensor(0.01), rbend_e2=torch.tensor(0.02))
    assert float(r.dipole_e1) == pytest.approx(0.11) and float(r.dipole_e2) == pytest.approx(0.12)
    rev = r.__dict__["_revision"]
    r.rbend_e1 = torch.tensor(0.05)
    r.rbend_e2 = torch.tensor(-0.03)
    assert float(r.dipole_e1) == pytest.approx(0.15) and float(r.dipole_e2) == pytest.approx(0.07)
    assert float(r.rbend_e1) == pytest.approx(0.05) and float(r.rbend_e2) == pytest.approx(-0.03)
    assert r.__dict__["_revision"] > rev
    seg = ca.Segment([r])
    seg.set_attrs_on_every_element(rbend_e1=torch.tensor(0.0))
    assert float(r.dipole_e1) == pytest.approx(0.1)


def test_flat_bcast_helper():
    from cheetah_amd import _ops

    x = torch.zeros(5, 7)
    f, B = _ops.flat_bcast(x, (3, 2), 2)
    assert f.shape == (1, 5, 7) and B == 1
    x = torch.zeros(3, 2, 5, 7)
    f, B = _ops.flat_bcast(x, (3, 2), 2)
    assert f.shape == (6, 5, 7) and B == 6
    x = torch.zeros(3, 1, 5, 7)
    f, B = _ops.flat_bcast(x, (3, 2), 2)
    assert f.shape == (6, 5, 7) and B == 6
    assert _ops.dtype_code(torch.float32) == 0 and _ops.dtype_code(torch.float64) == 1
    with pytest.raises(TypeError):
        _ops.dtype_code(torch.float16)


def test_beam_factories_shapes_and_defaults():
    import cheetah_amd as ca

    torch.manual_seed(0)
    b = ca.ParticleBeam.from_parameters(num_particles=20000)
    assert b.particles.shape == (20000, 7) and torch.all(b.particles[:, 6] == 1)
    s = b.particles[:, :6].std(dim=0)
    assert torch.allclose(s, torch.tensor([175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3]), rtol=1e-3)
    assert float(b.energy) == 1e8 and b.particle_charges.shape == (20000,)
    b = ca.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14), beta_y=torch.tensor(42.0), num_particles=5000)
    sx = (3.14 * 7.1971891e-13) ** 0.5
    assert float(b.particles[:, 0].std()) == pytest.approx(sx, rel=1e-3)
    b = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=5000, radius_x=torch.tensor(1e-3),
                                             radius_y=torch.tensor(2e-3), radius_tau=torch.tensor(1e-4))
    r2 = (b.particles[:, 0] / 1e-3) ** 2 + (b.particles[:, 2] / 2e-3) ** 2 + (b.particles[:, 4] / 1e-4) ** 2
    assert float(r2.max()) <= 1.0 + 1e-5
    sp = ca.Species("proton")
    assert sp.mass_eV_float == pytest.approx(938272089.43) and sp.num_elementary_charges_float == 1.0


def test_persistent_run_plan_host_logic():
    """`_FastRun` (the host half of chx_run_track): eligibility, the process-wide epoch, incremental refresh that patches only the
    pointers of re-assigned settings and keeps the device state. No kernel is launched here."""
    import cheetah_amd as ca
    from cheetah_amd import _ops
    from cheetah_amd.accelerator.element import Element
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.segment import _FastRun

    t = torch.tensor
    q1 = ca.Quadrupole(t(0.2), k1=t(4.2), name="q1")
    seg = ca.Segment([ca.Marker(), ca.Drift(t(0.5)), q1, ca.HorizontalCorrector(t(0.02), angle=t(1e-4), name="ch"), ca.BPM()])
    run = seg._plan()[0][1]
    fr = _FastRun(run, torch.float32, torch.device("cpu"))
    assert fr.ok and fr.E == 3                       # Marker and BPM are identity maps: left out
    assert list(fr.kinds) == [_ops.KIND["drift"], _ops.KIND["quadrupole"], _ops.KIND["hcor"]]
    state, ptrs = fr.state, fr.ptrs
    assert fr.ptrs[1 * _ops.MAX_PARAMS + 1] == q1.k1.data_ptr()
    # an in-place edit moves nothing on the host
    epoch = Element._epoch
    q1.k1.add_(1.0)
    assert Element._epoch == epoch and fr.epoch == epoch
    # a re-assignment moves the epoch; refresh patches that one pointer and keeps arrays and state
    new_k1 = t(-3.0)
    q1.k1 = new_k1
    assert Element._epoch > epoch
    fr.refresh()
    assert fr.ok and fr.state is state and fr.ptrs is ptrs and fr.ptrs[1 * _ops.MAX_PARAMS + 1] == new_k1.data_ptr()
    assert any(x is new_k1 for x in fr.tensors)
    # vectorised or trainable settings, other dtypes: not eligible, and eligible again afterwards
    q1.k1 = t([1.0, 2.0])
    fr.refresh()
    assert not fr.ok
    q1.k1 = t(1.0)
    fr.refresh()
    assert fr.ok
    # a trainable strength is read like any other setting: whether a track may USE the plan is asked per call
    # (`torch.is_grad_enabled() and _any_requires_grad(*plan.tensors)`), so a model evaluated under no_grad keeps its plans
    q2 = ca.Quadrupole(t(0.2), k1=torch.nn.Parameter(t(1.0)))
    fr2 = _FastRun(ca.Segment([q2])._plan()[0][1], torch.float32, torch.device("cpu"))
    assert fr2.ok and any(x is q2.k1 for x in fr2.tensors) and segment._any_requires_grad(*fr2.tensors)
    assert not _FastRun(run, torch.float64, torch.device("cpu")).ok
    # a cavity's skippability depends on a tensor VALUE: switched off it is an element of its run's plan (four settings, the
    # standing-wave builder); when the voltage comes back `_plan` re-partitions — the cavity is an item of its own, the partition
    # seen before returns with its plans when it is switched off again
    cav = ca.Cavity(t(1.0), voltage=t(0.0), phase=t(0.0), frequency=t(1.3e9))
    lin = ca.Segment([ca.Drift(t(1.0)), cav])
    plan_off = lin._plan()
    fr_cav = _FastRun(plan_off[0][1], torch.float32, torch.device("cpu"))
    assert fr_cav.ok and fr_cav.E == 2 and fr_cav.kinds[1] == _ops.KIND["cavity_sw"] and fr_cav.ptrs[_ops.MAX_PARAMS + 1] == cav.voltage.data_ptr()
    cav.voltage.fill_(1e6)
    plan_on = lin._plan()
    assert [k for k, _ in plan_on] == ["run", "element"] and plan_on[1][1] is cav and not cav._plannable()
    cav.voltage.zero_()
    assert lin._plan() is plan_off and cav._plannable()
    # editing the element list moves the epoch too
    e0 = Element._epoch
    seg.elements.append(ca.Drift(t(0.1)))
    assert Element._epoch > e0 and len(seg._plan()[0][1].elements) == 6
    # .to() replaces buffers behind __setattr__'s back: the elements are touched all the same
    e1, rev = Element._epoch, q1.__dict__["_revision"]
    seg.double()
    assert Element._epoch > e1 and q1.__dict__["_revision"] > rev and q1.k1.dtype == torch.float64


def test_register_fft_butterflies_on_the_host(tmp_path):
    """csrc/chx_fft_reg.h (the packed radix-4 butterflies of the line-FFT kernels) is host-callable: every transform
    size and direction against a direct DFT in long double, compiled for the host only (no GPU, no HIP runtime call)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "fft_reg_check"
    subprocess.run([hipcc, "--cuda-host-only", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "cheetah_amd", "csrc"),
                    "-o", str(exe), os.path.join(ROOT, "tests", "host", "fft_reg_check.hip")], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "f32: ok" in out.stdout and "f64: ok" in out.stdout


def test_beam_state_can_become_a_parameter():
    """The state tensors of both beam classes are read through class-level properties; assigning an nn.Parameter (a beam
    that is optimised) must move the name into `_parameters` like nn.Module does — and, like any nn.Module, refuse a plain
    tensor for that name afterwards."""
    import cheetah_amd as ca

    b = ca.ParticleBeam(torch.zeros(10, 7), torch.tensor(1e8))
    b.particles = torch.nn.Parameter(torch.ones(10, 7))
    assert isinstance(b.particles, torch.nn.Parameter) and "particles" in b._parameters and "particles" not in b._buffers
    assert [n for n, _ in b.named_parameters()] == ["particles"]
    with pytest.raises(TypeError, match="cannot assign"):
        b.particles = torch.full((10, 7), 2.0)
    b.energy = torch.tensor(2e8)                       # a buffer stays a buffer
    assert "energy" in b._buffers and float(b.energy) == 2e8
    p = ca.ParameterBeam(torch.zeros(7), torch.eye(7), torch.tensor(1e8))
    p.mu = torch.nn.Parameter(torch.ones(7))
    assert isinstance(p.mu, torch.nn.Parameter) and "mu" in p._parameters and "mu" not in p._buffers
    p.cov = 2 * torch.eye(7)
    assert "cov" in p._buffers and float(p.cov[0, 0]) == 2.0
    assert not hasattr(ca.ParticleBeam.__new__(ca.ParticleBeam), "nonexistent_attribute")


def test_cross_device_launch_is_refused(monkeypatch):
    """Kernels go to the current device's stream: tensors of another GPU of the same process must not be launched on it."""
    from cheetah_amd import _ops

    monkeypatch.setattr(_ops, "_raw_stream", lambda index: 0)
    monkeypatch.setattr(_ops, "_current_device", lambda: 1)
    _ops.check_current_device(torch.device("cuda:1"))
    _ops.check_current_device(None)
    with pytest.raises(RuntimeError, match=r"live on cuda:0 but the current device is cuda:1"):
        _ops.check_current_device(torch.device("cuda:0"))


def test_constructor_error_behaviour_mirrors_the_reference():
    """What the reference raises (or does not raise) for invalid constructor arguments, observed by running it:
    Screen asserts on the method and the resolution (screen.py:84-91); a Cavity accepts any `cavity_type` and fails with
    ValueError only when the map of an ACTIVE cavity is needed (cavity.py:337); an Aperture accepts any shape; Species and
    Superimposed assert; Segment look-ups raise ValueError."""
    import cheetah_amd as ca

    with pytest.raises(AssertionError, match="Invalid method"):
        ca.Screen(method="bogus")
    with pytest.raises(AssertionError, match="Invalid resolution"):
        ca.Screen(resolution=(1, 2, 3))
    off = ca.Cavity(length=torch.tensor(1.0), cavity_type="bogus")
    assert off.cavity_type == "bogus" and off._kind_name() == "cavity_sw"       # switched off: drift-like map
    on = ca.Cavity(length=torch.tensor(1.0), voltage=torch.tensor(1e6), frequency=torch.tensor(1.3e9), cavity_type="bogus")
    with pytest.raises(ValueError, match="Invalid cavity type: bogus"):
        on._kind_name()
    assert ca.Aperture(shape="bogus").shape == "bogus"
    with pytest.raises(AssertionError, match="zero length"):
        ca.Superimposed(ca.Drift(length=torch.tensor(1.0)), ca.Drift(length=torch.tensor(0.1)))
    seg = ca.Segment([ca.Drift(length=torch.tensor(0.1 * i), name=f"d{i}") for i in range(8)], name="long")
    for call in (lambda: seg.subcell(start="nope"), lambda: seg.subcell(end="nope"), lambda: seg.element_index("nope"),
                 lambda: seg.partition_at("nope")):
        with pytest.raises(ValueError):
            call()
    assert len(seg.subcell(start="d3", end="d2").elements) == 0
    # printed like the reference (segment.py:1061-1082): a ModuleList, beyond five elements the first and last two
    text = repr(seg)
    assert text.startswith("Segment(elements=ModuleList(\n  (0): Drift(name='d0'") and "\n   ⋮\n" in text and text.endswith("name='long')")
    assert repr(ca.Segment([ca.Marker(name="m")], name="s")) == "Segment(elements=ModuleList(\n  (0): Marker(name='m')\n), name='s')"


def test_segment_with_a_device_plan_can_be_copied_and_pickled(tmp_path):
    """A Segment whose runs carry a persistent device plan (`_FastRun`: ctypes pointer arrays) deep-copies, pickles and
    `torch.save`s like the reference's plain nn.Module (segment.py:45-71): derived caches are left out of the state, the copy
    plans again on first use and addresses its OWN tensors."""
    import copy
    import io
    import pickle

    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import _FastRun

    t = torch.tensor
    seg = ca.Segment([ca.Drift(t(0.5)), ca.Quadrupole(t(0.2), k1=t(4.2), name="q1"), ca.Drift(t(0.3)),
                      ca.SpaceChargeKick(t(0.1)), ca.Drift(t(0.2)), ca.Screen(name="scr", is_active=True)])
    for kind, item in seg._plan():
        if kind == "run":
            item.fast = _FastRun(item, torch.float32, torch.device("cpu"))   # what a GPU track leaves behind
            assert item.fast.ok
    assert seg.__dict__["_plan_cache"] is not None
    dup = copy.deepcopy(seg)
    assert dup.__dict__["_plan_cache"] is None and seg.__dict__["_plan_cache"] is not None
    assert dup.q1 is not seg.q1 and dup.q1.k1.data_ptr() != seg.q1.k1.data_ptr() and dup.elements[1] is dup.q1
    plan = dup._plan()
    fr = _FastRun(plan[0][1], torch.float32, torch.device("cpu"))
    assert fr.ok and fr.ptrs[1 * 9 + 1] == dup.q1.k1.data_ptr()                # the copy's plan points at the copy's tensors
    blob = pickle.dumps(seg)
    back = pickle.loads(blob)
    assert [type(e).__name__ for e in back.elements] == [type(e).__name__ for e in seg.elements]
    assert torch.equal(back.q1.k1, seg.q1.k1) and back.__dict__["_plan_cache"] is None
    buf = io.BytesIO()
    torch.save(seg, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert torch.equal(again.q1.k1, seg.q1.k1) and len(again._plan()) == len(seg._plan())


def test_storage_swap_under_a_planned_tensor_is_seen_when_its_element_is_reread():
    from cheetah_amd.accelerator.segment import _FastRun
    import cheetah_amd as ca

    t = torch.tensor
    q1 = ca.Quadrupole(t(0.2), k1=t(4.2), name="q1")
    seg = ca.Segment([ca.Drift(t(0.5)), q1])
    fr = _FastRun(seg._plan()[0][1], torch.float32, torch.device("cpu"))
    k1 = q1.k1
    old = k1.data_ptr()
    k1.data = t(-1.0)                       # same tensor object, new storage, no assignment
    with pytest.raises(RuntimeError, match="storage of setting"):
        fr.verify()                         # what CHX_CHECK_PLANS=1 runs per track
    q1.length = t(0.25)                     # any assignment on the element re-reads it: the swapped address is picked up
    fr.refresh()
    assert fr.ok and fr.ptrs[1 * 9 + 1] == k1.data_ptr() != old
    fr.verify()


def test_requires_grad_scan_extension():
    """cheetah_amd._chxtorch (csrc/chx_torch_host.cpp): the flag scan the host layer runs over a run's setting tensors."""
    import torch

    from cheetah_amd import _chxtorch
    from cheetah_amd.accelerator import segment

    ts = tuple(torch.zeros(2) for _ in range(50))
    assert _chxtorch.any_requires_grad(ts) is False and _chxtorch.any_requires_grad(()) is False
    assert _chxtorch.any_requires_grad(ts + (torch.nn.Parameter(torch.zeros(1)),)) is True
    ts[7].requires_grad_(True)                     # in place: no counter moves — the reason the scan runs on every track
    assert _chxtorch.any_requires_grad(ts) is True
    assert _chxtorch.any_requires_grad((None, 1.0, "k1")) is False
    with pytest.raises(TypeError):
        _chxtorch.any_requires_grad(list(ts))
    assert segment._any_requires_grad(*ts) is True and segment._any_requires_grad() is False
    assert segment._any_requires_grad.__module__ == "cheetah_amd.accelerator.segment"      # the extension, not torch._C's parser
