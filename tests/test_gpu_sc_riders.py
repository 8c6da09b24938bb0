"""The kicks of a chain from the second one on form their grid geometry INSIDE the deposit / corner-table launches and run the deposit's
bookkeeping inside the convolution's first pass (csrc/chx_sc_geom_dev.h, chx_sc_tiles.h `sc_tile_schedule_block`; the kick's index in
the chain travels in bits 8.. of `chx_sc_kick_sorted`'s flags). Without an index the same kick launches the one-workgroup geometry
and bookkeeping kernels. Both must give the same beam (space_charge_kick.py:531-550 for the geometry): the sums of the riders are
built with fp64 atomics, so "the same" is the rounding of a float32 coordinate, not bit for bit; the chain's header (what the host's
guard reads) must agree exactly."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _chain(ca, beam, kicks, bins, indexed, side, R=None):
    from cheetah_amd import _ops

    dt = beam.particles.dtype
    N = beam.particles.shape[0]
    kw = {"dtype": dt, "device": "cuda"}
    state = _ops.sc_tile_state(N, bins, dt, beam.particles.device)
    assert state is not None
    x = beam.particles.contiguous()
    q = beam.particle_charges.to(dt).contiguous()
    w = beam.survival_probabilities.to(dt).contiguous()
    outs = []
    for i in range(kicks):
        length = torch.tensor([0.2 + 0.05 * i], **kw)                  # (every kick its own length and extent)
        extent = torch.tensor([[3.0 + 0.1 * i, 3.0, 3.0 - 0.05 * i]], **kw)
        x = _ops.sc_kick_sorted(x, q, w, beam.energy.to(dt).reshape(1), length, extent, beam.species.mass_eV_float, N, bins, state,
                                i == 0, i == kicks - 1, side_stream=side, post_map_ptr=None if R is None else R.data_ptr(),
                                index=i if indexed else 0)
        outs.append(x)
    torch.cuda.synchronize()
    return outs, state[:32].view(torch.int32).clone()


@pytest.mark.parametrize("bins,n,with_side,with_map", [((64, 64, 64), 200_000, True, True), ((32, 32, 32), 120_000, False, False),
                                                       ((128, 64, 32), 150_001, True, False), ((128, 128, 128), 300_000, True, True)])
def test_riders_equal_the_geometry_and_bookkeeping_launches(bins, n, with_side, with_map):
    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(11)
    beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(3e-4), sigma_y=t(2e-4), sigma_tau=t(1e-4), sigma_px=t(2e-5),
                                           sigma_py=t(3e-5), sigma_p=t(1e-3), energy=t(5e7), total_charge=t(1e-9), **kw)
    side = torch.cuda.Stream() if with_side else None
    R = None
    if with_map:
        R = torch.eye(7, **kw)
        R[0, 1] = 0.3
        R[2, 3] = 0.3                                                   # a drift between the kicks, applied inside the gather pass
    got, hdr_got = _chain(ca, beam, 4, bins, True, side, R)
    ref, hdr_ref = _chain(ca, beam, 4, bins, False, side, R)
    ref2, _ = _chain(ca, beam, 4, bins, False, side, R)                  # the chain against itself: the float atomics of the deposit
    assert torch.equal(hdr_got, hdr_ref), (hdr_got, hdr_ref)
    assert int(hdr_got[6]) == 4                                          # four deposits were booked
    scale = ref[-1].abs().max(dim=0).values.clamp_min(1e-30)
    eps = float(torch.finfo(dt).eps)

    def ulps(a, b, k):
        if k < 3:       # rows in tile order, and the order inside a tile is the arrival order of the sort's atomics: compare as sets
            a, b = a.sort(dim=0).values, b.sort(dim=0).values
        return float(((a - b).abs().max(dim=0).values / (eps * scale))[:6].max())

    for k in range(4):
        assert torch.isfinite(got[k]).all()
        err, noise = ulps(got[k], ref[k], k), ulps(ref2[k], ref[k], k)
        print(f"kick {k}: riders vs launches {err:.2f} ulp of the coordinate's scale, launches vs launches {noise:.2f}")
        assert err <= 2.0 * noise + 2.0, (k, err, noise)
    # and the riders' kicks do something: the chain differs from no kick at all
    assert float((got[-1][:, 1] - beam.particles[:, 1]).abs().max()) > 0


def test_indexed_kick_refuses_foreign_moments():
    """An indexed kick takes its geometry from the chain's own sums; handing it the beam moments of a sharded beam is a usage error."""
    import ctypes

    from cheetah_amd import _lib, _ops

    lib = _lib.lib()
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    N, bins = 70_000, (32, 32, 32)
    x = torch.randn(N, 7, **kw) * 1e-3
    x[:, 6] = 1.0
    state = _ops.sc_tile_state(N, bins, dt, x.device)
    ws_bytes = lib.chx_sc_kick_sorted_workspace_bytes(N, _ops._bins3(bins), _ops.dtype_code(dt))
    ws = _ops.workspace(ws_bytes, x.device)
    mom = torch.zeros(1, 29, dtype=torch.float64, device="cuda")
    e, L, ext = torch.tensor([5e7], **kw), torch.tensor([0.2], **kw), torch.tensor([[3.0, 3.0, 3.0]], **kw)
    rho = ctypes.c_void_p()
    rc = lib.chx_sc_kick_sorted_begin(x.data_ptr(), None, None, e.data_ptr(), L.data_ptr(), ext.data_ptr(), 510998.95, N, _ops._bins3(bins),
                                      _ops.dtype_code(dt), ws.data_ptr(), ws_bytes, state.data_ptr(), state.numel(), (1 << 8), mom.data_ptr(), 0,
                                      ctypes.byref(rho), _ops.stream_ptr(), None)
    assert rc < 0
