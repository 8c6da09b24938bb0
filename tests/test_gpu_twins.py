"""The device entry points of include/chx.h against their host twins of include/chx_cpu.h (oracle/libchx_cpu.so) on the same inputs:
what a binding author who developed against the twins on a machine without a GPU may expect when the device library takes over.
Both are called through ctypes with the ONE argument list the two headers share — device pointers for one, host pointers for the
other."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

i64, vp, dbl = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double


@pytest.fixture(scope="module")
def libs():
    from cheetah_amd import _lib
    from oracle import chx_oracle

    return _lib.lib(), ctypes.CDLL(chx_oracle.build_cpu_abi())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hp(a):
    return a.ctypes.data_as(vp)


@pytest.mark.parametrize("dtype,code", [(np.float64, 1), (np.float32, 0)])
def test_device_entry_points_equal_their_host_twins(libs, dtype, code):
    from cheetah_amd import _ops

    gpu, cpu = libs
    rng = np.random.default_rng(31)
    tdt = torch.float64 if code else torch.float32
    eps = np.finfo(dtype).eps
    mass = 510998.95069
    stream = _ops.stream_ptr()
    # ---- chx_build_rmatrix: a batch of quadrupoles, of dipoles
    for kind, P, draw in ((2, 5, lambda n: np.stack([rng.uniform(0.05, 0.5, n), rng.uniform(-15, 15, n), rng.uniform(-0.3, 0.3, n) * (rng.random(n) < 0.5),
                                                     rng.normal(size=n) * 1e-4, rng.normal(size=n) * 1e-4], axis=1)),
                          (3, 9, lambda n: np.stack([rng.uniform(0.2, 1.0, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-2, 2, n), rng.uniform(-0.1, 0.1, n),
                                                     rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n) * (rng.random(n) < 0.5), rng.uniform(0, 0.5, n),
                                                     rng.uniform(0, 0.5, n), rng.uniform(0, 0.05, n)], axis=1))):
        B = 64
        p = draw(B).astype(dtype)
        e = rng.uniform(5e6, 3e8, B).astype(dtype)
        want = np.empty((B, 7, 7), dtype=dtype)
        assert cpu.chx_build_rmatrix_cpu(kind, hp(p), hp(e), dbl(mass), dbl(-1.0), i64(B), i64(B), i64(B), code, hp(want), None) == 0
        got = torch.empty((B, 7, 7), dtype=tdt, device="cuda")
        pd, ed = dev(p), dev(e)
        assert gpu.chx_build_rmatrix(kind, pd.data_ptr(), ed.data_ptr(), mass, -1.0, B, B, B, code, got.data_ptr(), stream) == 0
        # (the device's and the host's sines / cosines differ in the last bits of float64; float32 maps are rounded from them)
        assert np.allclose(got.cpu().numpy(), want, rtol=1e-11 if code else 2 * eps, atol=1e-14 if code else 1e-9), kind
    # ---- chx_compose_maps + chx_apply_affine7 (the fma chain: the same bits)
    E, B, N = 6, 3, 1000
    maps = (np.eye(7) + 0.05 * rng.standard_normal((E, B, 7, 7))).astype(dtype)
    maps[..., 6, :] = [0, 0, 0, 0, 0, 0, 1]
    md = dev(maps)
    hptrs = (vp * E)(*[maps[e].ctypes.data for e in range(E)])
    dptrs = (vp * E)(*[md[e].data_ptr() for e in range(E)])
    bc = (ctypes.c_uint8 * E)(*([0] * E))
    want = np.empty((B, 7, 7), dtype=dtype)
    got = torch.empty((B, 7, 7), dtype=tdt, device="cuda")
    assert cpu.chx_compose_maps_cpu(hptrs, bc, i64(E), i64(B), code, hp(want), None) == 0
    assert gpu.chx_compose_maps(dptrs, bc, E, B, code, got.data_ptr(), stream) == 0
    assert np.allclose(got.cpu().numpy(), want, rtol=4 * eps, atol=4 * eps)
    x = (rng.standard_normal((B, N, 7)) * 1e-3).astype(dtype)
    x[..., 6] = 1
    xd, Rd = dev(x), dev(want)
    y_want, y_got = np.empty_like(x), torch.empty_like(xd)
    assert cpu.chx_apply_affine7_cpu(hp(x), hp(want), hp(y_want), i64(B), i64(B), i64(B), i64(N), code, None) == 0
    assert gpu.chx_apply_affine7(xd.data_ptr(), Rd.data_ptr(), y_got.data_ptr(), B, B, B, N, code, stream) == 0
    assert np.array_equal(y_got.cpu().numpy(), y_want)
    # ---- chx_moments, chx_merge_moments, chx_moment_entry, chx_moment_entry_mapped_bwd, chx_moments_mapped_bwd
    w = (0.2 + rng.random((B, N))).astype(dtype)
    wd = dev(w)
    mom_want = np.empty((B, 29))
    assert cpu.chx_moments_cpu(hp(x), hp(w), i64(B), i64(B), i64(B), i64(N), code, hp(mom_want), None, ctypes.c_size_t(0), None) == 0
    ws_bytes = gpu.chx_moments_workspace_bytes(B, N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    mom_got = torch.empty((B, 29), dtype=torch.float64, device="cuda")
    assert gpu.chx_moments(xd.data_ptr(), wd.data_ptr(), B, B, B, N, code, mom_got.data_ptr(), ws.data_ptr(), ws_bytes, stream) == 0
    sig = np.sqrt(np.abs(mom_want[:, [8, 14, 19, 23, 26, 28]]))
    assert np.allclose(mom_got.cpu().numpy()[:, :8], mom_want[:, :8], rtol=1e-12, atol=1e-12 * sig.max())
    assert np.allclose(mom_got.cpu().numpy()[:, 8:], mom_want[:, 8:], rtol=1e-9, atol=1e-12 * sig.max() ** 2)
    merged_want, merged_got = np.empty((1, 29)), torch.empty((1, 29), dtype=torch.float64, device="cuda")
    assert cpu.chx_merge_moments_cpu(hp(mom_want), ctypes.c_int32(B), i64(1), hp(merged_want), None) == 0
    assert gpu.chx_merge_moments(dev(mom_want).data_ptr(), B, 1, merged_got.data_ptr(), stream) == 0
    assert np.allclose(merged_got.cpu().numpy(), merged_want, rtol=1e-13, atol=0)
    y_mom = np.empty((B, 29))
    assert cpu.chx_moments_cpu(hp(y_want), hp(w), i64(B), i64(B), i64(B), i64(N), code, hp(y_mom), None, ctypes.c_size_t(0), None) == 0
    g = rng.standard_normal(B).astype(dtype)
    for index, take_sqrt in ((8, 1), (4, 0), (20, 0), (26, 1)):
        e_want, e_got = np.empty(B, dtype=dtype), torch.empty(B, dtype=tdt, device="cuda")
        assert cpu.chx_moment_entry_cpu(hp(y_mom), i64(B), index, take_sqrt, code, hp(e_want), None) == 0
        assert gpu.chx_moment_entry(dev(y_mom).data_ptr(), B, index, take_sqrt, code, e_got.data_ptr(), stream) == 0
        assert np.array_equal(e_got.cpu().numpy(), e_want)
        d_want, d_got = np.empty((B, 49)), torch.empty((B, 49), dtype=torch.float64, device="cuda")
        assert cpu.chx_moment_entry_mapped_bwd_cpu(hp(g), hp(y_mom), index, take_sqrt, hp(want), hp(mom_want), i64(B), i64(B), i64(B), code,
                                                   hp(d_want), 1, None) == 0
        assert gpu.chx_moment_entry_mapped_bwd(dev(g).data_ptr(), dev(y_mom).data_ptr(), index, take_sqrt, Rd.data_ptr(), dev(mom_want).data_ptr(),
                                               B, B, B, code, d_got.data_ptr(), 1, stream) == 0
        assert np.allclose(d_got.cpu().numpy(), d_want, rtol=1e-12, atol=1e-14 * np.abs(d_want).max())
    d_out = rng.standard_normal((B, 29))
    d_want, d_got = np.empty((B, 49)), torch.empty((B, 49), dtype=torch.float64, device="cuda")
    assert cpu.chx_moments_mapped_bwd_cpu(hp(d_out), hp(want), hp(mom_want), i64(B), i64(B), i64(B), code, hp(d_want), None) == 0
    assert gpu.chx_moments_mapped_bwd(dev(d_out).data_ptr(), Rd.data_ptr(), dev(mom_want).data_ptr(), B, B, B, code, d_got.data_ptr(), stream) == 0
    assert np.allclose(d_got.cpu().numpy(), d_want, rtol=1e-12, atol=1e-14 * np.abs(d_want).max())
    # ---- chx_run_build_compose: scalars read where they live
    kinds = np.array([1, 2, 5, 1, 2, 4], dtype=np.int32)
    params = [[0.3], [0.12, 4.2, 0.05, 1e-4, -2e-4], [0.05, -1e-4], [0.7], [0.12, -3.9, 0.0, 0.0, 0.0], [0.04, 2e-4]]
    hs = [[np.array([v], dtype=dtype) for v in p] for p in params]
    ds = [[dev(a) for a in row] for row in hs]
    hptrs, dptrs = (vp * (6 * 9))(), (vp * (6 * 9))()
    for e in range(6):
        for k in range(len(params[e])):
            hptrs[e * 9 + k] = hs[e][k].ctypes.data
            dptrs[e * 9 + k] = ds[e][k].data_ptr()
    energy = np.array([1.3e8], dtype=dtype)
    m_want, c_want = np.empty((6, 7, 7), dtype=dtype), np.empty((7, 7), dtype=dtype)
    m_got, c_got = torch.empty((6, 7, 7), dtype=tdt, device="cuda"), torch.empty((7, 7), dtype=tdt, device="cuda")
    assert cpu.chx_run_build_compose_cpu(hp(kinds), hptrs, i64(6), hp(energy), dbl(mass), dbl(-1.0), code, hp(m_want), hp(c_want), None) == 0
    assert gpu.chx_run_build_compose(hp(kinds), dptrs, 6, dev(energy).data_ptr(), mass, -1.0, code, m_got.data_ptr(), c_got.data_ptr(), stream) == 0
    assert np.allclose(m_got.cpu().numpy(), m_want, rtol=1e-12 if code else 2 * eps, atol=1e-15 if code else 1e-9)
    assert np.allclose(c_got.cpu().numpy(), c_want, rtol=1e-11 if code else 8 * eps, atol=1e-14 if code else 1e-8)
