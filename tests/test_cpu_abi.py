"""include/chx_cpu.h: the host twins of the core C-ABI entry points (oracle/libchx_cpu.so, compiled from the oracle's C) — every
declared symbol is exported with the argument list of its chx.h namesake, and gives the oracle's numbers (which the golden
fixtures pin to the reference: tests/test_oracle_golden.py). Runs without a GPU: that is the point of the twins."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i64, vp, dbl = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double


def _lib():
    from oracle import chx_oracle

    return ctypes.CDLL(chx_oracle.build_cpu_abi())


def _decls(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return {m.group(2): re.sub(r"\s+", " ", m.group(3)).strip()
            for m in re.finditer(r"\b(int|size_t|int64_t)\s+(chx_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)}


def _ptr(a):
    return a.ctypes.data_as(vp)


def test_every_twin_is_exported_with_the_signature_of_its_namesake():
    lib = _lib()
    twins, abi = _decls("chx_cpu.h"), _decls("chx.h")
    assert set(twins) == {"chx_abi_version_cpu", "chx_build_rmatrix_cpu", "chx_compose_maps_cpu", "chx_apply_affine7_cpu",
                          "chx_moments_cpu", "chx_cic_deposit_cpu", "chx_track_elementwise_cpu", "chx_cavity_coeffs_cpu",
                          "chx_cavity_track_cpu", "chx_hist2d_cpu", "chx_sc_kick_workspace_bytes_cpu", "chx_sc_kick_cpu",
                          "chx_track_fused_cpu", "chx_apply_bwd_workspace_bytes_cpu", "chx_apply_affine7_bwd_cpu", "chx_moments_bwd_cpu",
                          "chx_moments_bwd_w_cpu", "chx_cic_deposit_bwd_cpu", "chx_sc_gather_kick_cpu", "chx_merge_moments_cpu",
                          "chx_moments_mapped_bwd_cpu", "chx_moment_entry_cpu", "chx_moment_entry_mapped_bwd_cpu",
                          "chx_build_rmatrix_scalars_cpu", "chx_run_build_compose_cpu"}
    norm = lambda sig: re.sub(r"\s*/\*.*?\*/", "", sig).replace(" ,", ",")  # noqa: E731
    for name, sig in twins.items():
        assert hasattr(lib, name), name
        base = name[:-4]
        assert base in abi, base
        assert norm(sig) == norm(abi[base]), (name, sig, abi[base])
    exported = os.popen(f"nm -D --defined-only {os.path.join(ROOT, 'oracle', 'libchx_cpu.so')}").read()
    assert "chxo_" not in exported                      # the oracle's own entry points stay internal
    assert lib.chx_abi_version_cpu() == int(re.search(r"#define CHX_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "chx.h")).read()).group(1))


def test_twins_give_the_oracles_numbers(oracle):
    lib = _lib()
    rng = np.random.default_rng(7)
    # maps: quadrupole with tilt + misalignment, fp32 and fp64 storage
    for dtype, code in ((np.float64, 1), (np.float32, 0)):
        params = np.array([[0.2, 4.2, 0.1, 1e-4, -2e-4], [0.3, -3.0, 0.0, 0.0, 0.0]], dtype=dtype)
        energy = np.array([1e8], dtype=dtype)
        R = np.empty((2, 7, 7), dtype=dtype)
        assert lib.chx_build_rmatrix_cpu(2, _ptr(params), _ptr(energy), dbl(oracle.ELECTRON_MASS_EV), dbl(-1.0), i64(2), i64(2), i64(1),
                                         code, _ptr(R), None) == 0
        ref = oracle.build_rmatrix("quadrupole", params.astype(np.float64), energy.astype(np.float64))
        assert np.allclose(R, ref.astype(dtype), rtol=0, atol=0)
        # compose (one shared map, one per row)
        drift = oracle.build_rmatrix("drift", [0.8], 1e8).astype(dtype)
        ptrs = (vp * 3)(_ptr(R).value, _ptr(drift).value, _ptr(R).value)
        bc = (ctypes.c_uint8 * 3)(0, 1, 0)
        out = np.empty((2, 7, 7), dtype=dtype)
        assert lib.chx_compose_maps_cpu(ptrs, bc, i64(3), i64(2), code, _ptr(out), None) == 0
        want = np.stack([R[b].astype(np.float64) @ drift[0].astype(np.float64) @ R[b].astype(np.float64) for b in range(2)])
        assert np.allclose(out, want, rtol=1e-6 if code == 0 else 1e-14)
        # apply: the kernels' fma chain (bit-identical to the oracle's mode 1)
        x = (rng.standard_normal((1, 1000, 7)) * 1e-3).astype(dtype)
        x[..., 6] = 1
        y = np.empty((2, 1000, 7), dtype=dtype)
        assert lib.chx_apply_affine7_cpu(_ptr(x), _ptr(out), _ptr(y), i64(2), i64(1), i64(2), i64(1000), code, None) == 0
        assert np.array_equal(y, oracle.apply(x, out, mode=1))
        # moments
        w = rng.random((2, 1000)).astype(dtype)
        mom = np.empty((2, 29))
        assert lib.chx_moments_cpu(_ptr(y), _ptr(w), i64(2), i64(2), i64(2), i64(1000), code, _ptr(mom), None, ctypes.c_size_t(0), None) == 0
        assert np.allclose(mom, oracle.moments(y, w)["raw"], rtol=1e-13, atol=0)
    # cloud in cell through the struct of the ABI
    from cheetah_amd._lib import CicArgs

    x = (rng.standard_normal((1, 4000, 7)) * [1e-3, 0, 2e-3, 0, 0, 0, 0]).astype(np.float32)
    q = rng.random((1, 4000)).astype(np.float32)
    extent = np.array([[[-3e-3, 3e-3], [-5e-3, 5e-3]]], dtype=np.float32)
    grid = np.zeros((1, 16, 24), dtype=np.float32)
    a = CicArgs()
    a.ndim = 2
    a.cols[0], a.cols[1] = 0, 2
    a.bins[0], a.bins[1] = 16, 24
    a.B = a.Bx = a.Bq = a.Be = a.Bs = a.Bsc = a.Bsh = 1
    a.N, a.dtype, a.abs_charge = 4000, 0, 0
    a.x, a.charge, a.extent, a.grid = _ptr(x).value, _ptr(q).value, _ptr(extent).value, _ptr(grid).value
    assert lib.chx_cic_deposit_cpu(ctypes.byref(a), None) == 0
    ref = oracle.cic_deposit(x, (0, 2), (16, 24), extent[0], charge=q)
    assert np.array_equal(grid, ref.reshape(grid.shape))
    # argument checking like the GPU library: negative status, no exception across the ABI
    assert lib.chx_apply_affine7_cpu(None, _ptr(out), _ptr(y), i64(2), i64(1), i64(2), i64(1000), 0, None) == -1
    assert lib.chx_moments_cpu(_ptr(y), None, i64(2), i64(2), i64(1), i64(1000), 7, _ptr(mom), None, ctypes.c_size_t(0), None) == -2


def test_tracking_cavity_and_histogram_twins_give_the_oracles_numbers(oracle):
    """chx_track_elementwise_cpu / chx_cavity_coeffs_cpu / chx_cavity_track_cpu / chx_hist2d_cpu (segment.py:571-572,
    cavity.py:100-251, screen.py:292-311) through the argument lists of their chx.h namesakes."""
    from cheetah_amd._lib import Hist2dArgs

    lib = _lib()
    rng = np.random.default_rng(11)
    for dtype, code in ((np.float32, 0), (np.float64, 1)):
        # element by element, no merging: 7 maps on a beam shared by two rows of settings (Bx = 1, BR = B = 2)
        E, B, N = 7, 2, 1500
        maps = np.stack([np.stack([oracle.build_rmatrix("quadrupole", [0.2, 4.2 * (-1) ** e * (1 + b), 0, 0, 0], 1e8)[0] if e % 2 == 0
                                   else oracle.build_rmatrix("drift", [0.5 + 0.1 * b], 1e8)[0] for b in range(B)]) for e in range(E)]).astype(dtype)
        x = (rng.standard_normal((1, N, 7)) * [2e-4, 1e-5, 2e-4, 1e-5, 1e-4, 1e-3, 0]).astype(dtype)
        x[..., 6] = 1
        out = np.empty((B, N, 7), dtype=dtype)
        assert lib.chx_track_elementwise_cpu(_ptr(x), _ptr(maps), _ptr(out), None, i64(E), i64(B), i64(1), i64(B), i64(N), code, None) == 0
        for b in range(B):
            assert np.array_equal(out[b], oracle.track_elementwise(x[0], maps[:, b]))
        assert lib.chx_track_elementwise_cpu(_ptr(out), _ptr(maps), _ptr(out), None, i64(E), i64(B), i64(B), i64(B), i64(N), code, None) == -1
        # a cavity: coefficient rows and outgoing energies for two phases, then the particle update
        params = np.array([[1.0377, 18e6, -10.0, 1.3e9], [1.0377, 18e6, 170.0, 1.3e9]], dtype=dtype)
        energy = np.array([6e7], dtype=dtype)
        coeffs, e_out = np.empty((2, 8)), np.empty(2, dtype=dtype)
        assert lib.chx_cavity_coeffs_cpu(_ptr(params), _ptr(energy), dbl(oracle.ELECTRON_MASS_EV), dbl(-1.0), i64(2), i64(2), i64(1), code,
                                         _ptr(coeffs), _ptr(e_out), None) == 0
        c_ref, e_ref = oracle.cavity_coeffs(params.astype(np.float64), energy.astype(np.float64))
        assert np.array_equal(coeffs, c_ref) and np.array_equal(e_out, e_ref.astype(dtype))
        assert e_out[0] > 7.7e7 and e_out[1] < 4.3e7                                    # one row gains, the other loses energy
        R = oracle.build_rmatrix("cavity_sw", params.astype(np.float64), energy.astype(np.float64)).astype(dtype)
        y = np.empty((2, N, 7), dtype=dtype)
        assert lib.chx_cavity_track_cpu(_ptr(x), _ptr(R), _ptr(coeffs), _ptr(y), i64(2), i64(1), i64(N), code, None) == 0
        assert np.array_equal(y, oracle.cavity_track(x, R, coeffs))
        # the screen histogram through the struct of the ABI (misaligned screen, |q| * survival weights)
        edges_x, edges_y = np.linspace(-6e-4, 6e-4, 25).astype(dtype), np.linspace(-5e-4, 5e-4, 17).astype(dtype)
        q = (rng.standard_normal((1, N)) * 1e-15).astype(dtype)
        w = rng.random((1, N)).astype(dtype)
        shift = np.array([[5e-5, -2e-5]], dtype=dtype)
        image = np.zeros((1, 16, 24), dtype=dtype)
        a = Hist2dArgs()
        a.B = a.Bx = a.Bq = a.Bs = a.Bsh = 1
        a.N, a.nx, a.ny, a.dtype = N, 24, 16, code
        a.x, a.charge, a.survival, a.shift = _ptr(x).value, _ptr(q).value, _ptr(w).value, _ptr(shift).value
        a.edges_x, a.edges_y, a.image = _ptr(edges_x).value, _ptr(edges_y).value, _ptr(image).value
        assert lib.chx_hist2d_cpu(ctypes.byref(a), None) == 0
        ref, _ = oracle.hist2d(x, edges_x, edges_y, charge=q, survival=w, shift=shift)
        assert np.array_equal(image, ref) and image.sum() > 0
        a.dtype = 5
        assert lib.chx_hist2d_cpu(ctypes.byref(a), None) == -2


def test_space_charge_kick_twin_gives_the_oracles_numbers(oracle):
    """chx_sc_kick_cpu (space_charge_kick.py:477-586) through chx_sc_kick's argument list: the oracle's pipeline
    (`chx_oracle.space_charge_kick`, pinned to the reference by tests/golden/space_charge*.npz in tests/test_oracle_golden.py) with the
    cyclic convolution done by the twin's own radix-2 transforms instead of numpy.fft."""
    lib = _lib()
    lib.chx_sc_kick_workspace_bytes_cpu.restype = ctypes.c_size_t
    rng = np.random.default_rng(21)
    B, N, g = 2, 4000, (16, 16, 32)
    bins = (ctypes.c_int32 * 3)(*g)
    x64 = rng.standard_normal((B, N, 7)) * np.array([3e-4, 2e-5, 2e-4, 3e-5, 1e-4, 1e-3, 0.0])
    x64[..., 6] = 1.0
    x64[0, :20, 0] *= 8.0                                             # a few particles beyond the grid
    q64 = np.full((B, N), -1e-9 / N)
    w64 = np.where(rng.random((B, N)) < 0.1, 0.5, 1.0)
    for dtype, code, tol in ((np.float64, 1, 1e-9), (np.float32, 0, 2e-5)):
        x, q, w = x64.astype(dtype), q64.astype(dtype), w64.astype(dtype)
        energy = np.array([5e7, 8e7], dtype=dtype)
        length = np.array([0.2, 0.35], dtype=dtype)
        ext = np.array([[3.0, 3.0, 3.0]], dtype=dtype)
        ref = np.stack([oracle.space_charge_kick(x[b:b + 1], float(energy[b]), q[b:b + 1], w[b:b + 1], float(length[b]), grid_shape=g)[0]
                        for b in range(B)])
        assert lib.chx_sc_kick_workspace_bytes_cpu(i64(B), i64(N), bins, code) > 0
        out = np.empty_like(x)
        rc = lib.chx_sc_kick_cpu(_ptr(x), _ptr(q), _ptr(w), _ptr(energy), _ptr(length), _ptr(ext), dbl(oracle.ELECTRON_MASS_EV), i64(B), i64(B),
                                 i64(B), i64(B), i64(1), i64(N), bins, code, _ptr(out), None, ctypes.c_size_t(0), None, None, None, i64(1))
        assert rc == 0
        kick_ref, kick = ref.astype(np.float64) - x, out.astype(np.float64) - x
        scale = np.abs(kick_ref).max(axis=(0, 1))
        assert scale[1] > 0 and scale[3] > 0 and scale[5] > 0           # the kick acts on px, py, delta
        err = (np.abs(kick - kick_ref).max(axis=(0, 1))[[1, 3, 5]] / scale[[1, 3, 5]]).max()
        assert err < tol, err
        assert np.array_equal(out[..., [0, 2, 4, 6]], ref[..., [0, 2, 4, 6]])        # positions: the same round trip through SI units
        # the linear run behind the kick, applied by the same call (fma chain of chx_apply_affine7_cpu)
        R = (np.eye(7) + 0.05 * rng.standard_normal((7, 7))).astype(dtype)
        R[6] = 0
        R[6, 6] = 1
        out2, chained = np.empty_like(x), np.empty_like(x)
        assert lib.chx_sc_kick_cpu(_ptr(x), _ptr(q), _ptr(w), _ptr(energy), _ptr(length), _ptr(ext), dbl(oracle.ELECTRON_MASS_EV), i64(B), i64(B),
                                   i64(B), i64(B), i64(1), i64(N), bins, code, _ptr(out2), None, ctypes.c_size_t(0), None, None, _ptr(R), i64(1)) == 0
        assert lib.chx_apply_affine7_cpu(_ptr(out), _ptr(R), _ptr(chained), i64(B), i64(B), i64(1), i64(N), code, None) == 0
        assert np.array_equal(out2, chained)
    # argument checks of the namesake: a grid edge that is not a power of two >= 16, a dtype code that does not exist
    bad = (ctypes.c_int32 * 3)(16, 24, 16)
    assert lib.chx_sc_kick_cpu(_ptr(x), _ptr(q), _ptr(w), _ptr(energy), _ptr(length), _ptr(ext), dbl(oracle.ELECTRON_MASS_EV), i64(B), i64(B), i64(B),
                               i64(B), i64(1), i64(N), bad, 0, _ptr(out), None, ctypes.c_size_t(0), None, None, None, i64(1)) == -1
    assert lib.chx_sc_kick_cpu(_ptr(x), _ptr(q), _ptr(w), _ptr(energy), _ptr(length), _ptr(ext), dbl(oracle.ELECTRON_MASS_EV), i64(B), i64(B), i64(B),
                               i64(B), i64(1), i64(N), bins, 7, _ptr(out), None, ctypes.c_size_t(0), None, None, None, i64(1)) == -2


def test_backward_and_gather_twins_against_torch_autograd_and_the_oracle(oracle):
    """The round-6 twins. The backward entry points are checked against torch autograd of the reference's own tensor expressions
    (element.py:180-191, utils/statistics.py:4-62, utils/cloud_in_cell.py written out in float64) — independent of the C they test;
    the fused track and the gather against the twins / the oracle call they must equal."""
    import torch

    from cheetah_amd._lib import CicArgs

    lib = _lib()
    rng = np.random.default_rng(11)
    B, N, E = 2, 700, 5
    # ---- chx_track_fused_cpu == chx_track_elementwise_cpu, bit for bit (float32 storage: the rounding between elements matters)
    x = (rng.standard_normal((B, N, 7)) * 1e-3).astype(np.float32)
    x[..., 6] = 1
    R = (np.eye(7) + 0.05 * rng.standard_normal((E, B, 7, 7))).astype(np.float32)
    R[..., 6, :] = [0, 0, 0, 0, 0, 0, 1]
    a, b = np.empty_like(x), np.empty_like(x)
    assert lib.chx_track_elementwise_cpu(_ptr(x), _ptr(R), _ptr(a), None, i64(E), i64(B), i64(B), i64(B), i64(N), 0, None) == 0
    assert lib.chx_track_fused_cpu(_ptr(x), _ptr(R), _ptr(b), i64(E), i64(B), i64(B), i64(B), i64(N), 0, None) == 0
    assert np.array_equal(a, b) and not np.array_equal(a, x)
    # ---- chx_apply_affine7_bwd_cpu: y = x @ R.mT under autograd
    xt = torch.from_numpy(rng.standard_normal((B, N, 7))).requires_grad_(True)
    Rt = torch.from_numpy(np.eye(7) + 0.1 * rng.standard_normal((B, 7, 7))).requires_grad_(True)
    dY = torch.from_numpy(rng.standard_normal((B, N, 7)))
    (xt @ Rt.mT).backward(dY)
    dX, dR = np.empty((B, N, 7)), np.empty((B, 49))
    assert lib.chx_apply_affine7_bwd_cpu(_ptr(dY.numpy()), _ptr(Rt.detach().numpy()), _ptr(xt.detach().numpy()), _ptr(dX), _ptr(dR), i64(B), i64(B),
                                         i64(B), i64(N), 1, None, ctypes.c_size_t(0), None) == 0
    assert np.allclose(dX, xt.grad.numpy(), rtol=1e-13, atol=1e-15) and np.allclose(dR.reshape(B, 7, 7), Rt.grad.numpy(), rtol=1e-12, atol=1e-13)
    assert lib.chx_apply_bwd_workspace_bytes_cpu(i64(B), i64(N)) == 0
    # ---- chx_moments_bwd_w_cpu: the reference's weighted statistics (utils/statistics.py:4-62) under autograd
    xs = torch.from_numpy(rng.standard_normal((B, N, 7)) * [1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0] + [5e-3, 0, -1e-3, 0, 0, 0, 1]).requires_grad_(True)
    ws = torch.from_numpy(0.1 + rng.random((B, N))).requires_grad_(True)
    W, W2 = ws.sum(-1), ws.square().sum(-1)
    mu = (ws.unsqueeze(-1) * xs[..., :6]).sum(-2) / W.unsqueeze(-1)
    d = xs[..., :6] - mu.unsqueeze(-2)
    cov = torch.einsum("bn,bni,bnj->bij", ws, d, d) / (W - W2 / W)[:, None, None]
    iu = torch.triu_indices(6, 6)
    out_t = torch.cat([W[:, None], W2[:, None], mu, cov[:, iu[0], iu[1]]], dim=-1)          # the 29-vector of chx_moments
    g = torch.from_numpy(rng.standard_normal((B, 29)))
    out_t.backward(g)
    mom = np.empty((B, 29))
    assert lib.chx_moments_cpu(_ptr(xs.detach().numpy()), _ptr(ws.detach().numpy()), i64(B), i64(B), i64(B), i64(N), 1, _ptr(mom), None,
                               ctypes.c_size_t(0), None) == 0
    assert np.allclose(mom, out_t.detach().numpy(), rtol=1e-10, atol=1e-22)
    dXm, dWm = np.empty((B, N, 7)), np.empty((B, N))
    assert lib.chx_moments_bwd_w_cpu(_ptr(xs.detach().numpy()), _ptr(ws.detach().numpy()), _ptr(mom), _ptr(g.numpy()), i64(B), i64(B), i64(B),
                                     i64(N), 1, _ptr(dXm), _ptr(dWm), None) == 0
    scale = np.abs(xs.grad.numpy()).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(dXm - xs.grad.numpy()) / scale) < 1e-9 and np.all(dXm[..., 6] == 0)
    assert np.max(np.abs(dWm - ws.grad.numpy())) < 1e-9 * np.abs(ws.grad.numpy()).max()
    dXo = np.empty((B, N, 7))
    assert lib.chx_moments_bwd_cpu(_ptr(xs.detach().numpy()), _ptr(ws.detach().numpy()), _ptr(mom), _ptr(g.numpy()), i64(B), i64(B), i64(B), i64(N),
                                   1, _ptr(dXo), None) == 0
    assert np.array_equal(dXo, dXm)
    # ---- chx_cic_deposit_bwd_cpu: the 2-D cloud-in-cell deposit (utils/cloud_in_cell.py:178-239) under autograd
    n2, bins = 900, (12, 10)
    pos = torch.from_numpy(rng.standard_normal((n2, 2)) * [1.2e-3, 2.0e-3]).requires_grad_(True)      # some land outside the extent
    qq = torch.from_numpy(rng.random(n2)).requires_grad_(True)
    ext = torch.tensor([[-3e-3, 3e-3], [-4e-3, 4e-3]], dtype=torch.float64)
    bw = (ext[:, 1] - ext[:, 0]) / torch.tensor(bins, dtype=torch.float64)
    pb = (pos - ext[:, 0]) / bw - 0.5
    i0 = pb.detach().floor().long()
    f = pb - i0
    inside = ((pos >= ext[:, 0]) & (pos <= ext[:, 1])).all(-1).to(torch.float64).detach()
    grid_t = torch.zeros(bins[0] * bins[1], dtype=torch.float64)
    for ox in (0, 1):
        for oy in (0, 1):
            ix, iy = i0[:, 0] + ox, i0[:, 1] + oy
            ok = ((ix >= 0) & (ix < bins[0]) & (iy >= 0) & (iy < bins[1])).to(torch.float64)
            wgt = (f[:, 0] if ox else 1 - f[:, 0]) * (f[:, 1] if oy else 1 - f[:, 1])
            grid_t = grid_t.scatter_add(0, ix.clamp(0, bins[0] - 1) * bins[1] + iy.clamp(0, bins[1] - 1), qq * inside * ok * wgt)
    dgrid = torch.from_numpy(rng.standard_normal(bins[0] * bins[1]))
    grid_t.backward(dgrid)
    x7 = np.zeros((1, n2, 7))
    x7[0, :, 0], x7[0, :, 2] = pos.detach().numpy()[:, 0], pos.detach().numpy()[:, 1]
    qn, en = np.ascontiguousarray(qq.detach().numpy()[None]), np.ascontiguousarray(ext.numpy()[None])
    a = CicArgs()
    a.ndim = 2
    a.cols[0], a.cols[1] = 0, 2
    a.bins[0], a.bins[1] = bins
    a.B = a.Bx = a.Bq = a.Be = a.Bs = a.Bsc = a.Bsh = 1
    a.N, a.dtype, a.abs_charge = n2, 1, 0
    gridn = np.zeros((1, *bins))
    a.x, a.charge, a.extent, a.grid = _ptr(x7).value, _ptr(qn).value, _ptr(en).value, _ptr(gridn).value
    assert lib.chx_cic_deposit_cpu(ctypes.byref(a), None) == 0
    assert np.allclose(gridn.reshape(-1), grid_t.detach().numpy(), rtol=1e-12, atol=1e-18)         # same forward first
    dwt, dps = np.empty((1, n2)), np.empty((1, n2, 2))
    assert lib.chx_cic_deposit_bwd_cpu(ctypes.byref(a), _ptr(dgrid.numpy()), _ptr(dwt), _ptr(dps), None) == 0
    assert np.allclose(dwt[0], qq.grad.numpy(), rtol=1e-11, atol=1e-16)
    assert np.allclose(dps[0], pos.grad.numpy(), rtol=1e-11, atol=1e-12 * np.abs(pos.grad.numpy()).max())
    assert (inside == 0).any() and np.all(dwt[0][inside.numpy() == 0] == 0)
    # ---- chx_sc_gather_kick_cpu: the oracle's own gather on the oracle's own force grid (the last step of its space_charge_kick)
    n3 = 3000
    xk = (rng.standard_normal((1, n3, 7)) * [2e-4, 3e-6, 2e-4, 3e-6, 1e-5, 1e-3, 0]).astype(np.float64)
    xk[..., 6] = 1
    qk, wk = np.full((1, n3), 1e-13), np.ones((1, n3))
    want, det = oracle.space_charge_kick(xk, 1e8, qk, wk, 0.3, grid_shape=(16, 16, 16), details=True)
    F4 = np.zeros((1, 16, 16, 16, 4))
    F4[..., :3] = det["F"]
    got = np.empty_like(xk)
    b3 = (ctypes.c_int32 * 3)(16, 16, 16)
    assert lib.chx_sc_gather_kick_cpu(_ptr(xk), _ptr(F4), _ptr(np.ascontiguousarray(det["half"].astype(np.float64))),
                                      _ptr(np.ascontiguousarray(det["cell"].astype(np.float64))), _ptr(np.array([1e8])),
                                      _ptr(np.ascontiguousarray(det["dt"])), dbl(oracle.ELECTRON_MASS_EV), i64(1), i64(1), i64(1), i64(n3), b3, 1,
                                      _ptr(got), None) == 0
    assert np.array_equal(got, want) and not np.array_equal(got, xk)
    assert lib.chx_sc_gather_kick_cpu(_ptr(xk), None, None, None, None, None, dbl(1.0), i64(1), i64(1), i64(1), i64(n3), b3, 1, _ptr(got), None) == -1


def test_property_and_scalar_run_twins(oracle):
    """The second round-6 batch: pooled moments of shards, a beam property of y = R x and its gradient with respect to R (checked
    against torch autograd of the reference's statistics, utils/statistics.py:4-62, through y = x @ R.mT), the maps of a run whose
    settings are scalars read where they live and their product (against the batched builder twin / the oracle)."""
    import torch

    lib = _lib()
    rng = np.random.default_rng(23)
    N = 900
    x = rng.standard_normal((1, N, 7)) * [1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0] + [2e-3, 0, -1e-3, 0, 0, 0, 1]
    w = 0.2 + rng.random((1, N))

    def moments(xa, wa):
        out = np.empty((xa.shape[0], 29))
        assert lib.chx_moments_cpu(_ptr(np.ascontiguousarray(xa)), _ptr(np.ascontiguousarray(wa)), i64(xa.shape[0]), i64(xa.shape[0]), i64(xa.shape[0]),
                                   i64(xa.shape[1]), 1, _ptr(out), None, ctypes.c_size_t(0), None) == 0
        return out

    # ---- chx_merge_moments_cpu: three unequal shards (one of them without weight) pool to the moments of the whole
    whole = moments(x, w)
    cuts = [(0, 250), (250, 251), (251, N)]
    w_cut = w.copy()
    w_cut[:, 250:251] = 0.0                                         # a shard whose particles are all lost
    whole = moments(x, w_cut)
    parts = np.stack([moments(x[:, a:b], w_cut[:, a:b]) for a, b in cuts])       # (R, B, 29)
    merged = np.empty((1, 29))
    assert lib.chx_merge_moments_cpu(_ptr(np.ascontiguousarray(parts)), ctypes.c_int32(3), i64(1), _ptr(merged), None) == 0
    assert np.allclose(merged, whole, rtol=1e-11, atol=1e-24)
    # ---- a property of y = R x and d / dR, against torch autograd of the statistics themselves
    R = np.eye(7) + 0.2 * rng.standard_normal((1, 7, 7))
    R[:, 6, :] = [0, 0, 0, 0, 0, 0, 1]
    Rt = torch.from_numpy(R.copy()).requires_grad_(True)
    xs, ws = torch.from_numpy(x), torch.from_numpy(w)
    y = xs @ Rt.mT
    W, W2 = ws.sum(-1), ws.square().sum(-1)
    mu = (ws.unsqueeze(-1) * y[..., :6]).sum(-2) / W.unsqueeze(-1)
    d = y[..., :6] - mu.unsqueeze(-2)
    cov = torch.einsum("bn,bni,bnj->bij", ws, d, d) / (W - W2 / W)[:, None, None]
    iu = torch.triu_indices(6, 6)
    out_t = torch.cat([W[:, None], W2[:, None], mu, cov[:, iu[0], iu[1]]], dim=-1)
    g = rng.standard_normal((1, 29))
    out_t.backward(torch.from_numpy(g))
    mom_x, mom_y = moments(x, w), moments(y.detach().numpy(), w)
    dR = np.empty((1, 49))
    assert lib.chx_moments_mapped_bwd_cpu(_ptr(g), _ptr(R), _ptr(mom_x), i64(1), i64(1), i64(1), 1, _ptr(dR), None) == 0
    want = Rt.grad.numpy().reshape(1, 49)
    assert np.allclose(dR, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    for index, take_sqrt in ((8, 1), (3, 0), (19, 1), (10, 0)):
        picked = np.empty(1)
        assert lib.chx_moment_entry_cpu(_ptr(mom_y), i64(1), index, take_sqrt, 1, _ptr(picked), None) == 0
        assert picked[0] == (np.sqrt(mom_y[0, index]) if take_sqrt else mom_y[0, index])
        Rt.grad = None
        y = xs @ Rt.mT
        mu = (ws.unsqueeze(-1) * y[..., :6]).sum(-2) / W.unsqueeze(-1)
        d = y[..., :6] - mu.unsqueeze(-2)
        cov = torch.einsum("bn,bni,bnj->bij", ws, d, d) / (W - W2 / W)[:, None, None]
        out_t = torch.cat([W[:, None], W2[:, None], mu, cov[:, iu[0], iu[1]]], dim=-1)
        entry = out_t[0, index].sqrt() if take_sqrt else out_t[0, index]
        (0.7 * entry).backward()
        one = np.empty((1, 49))
        assert lib.chx_moment_entry_mapped_bwd_cpu(_ptr(np.array([0.7])), _ptr(mom_y), index, take_sqrt, _ptr(R), _ptr(mom_x), i64(1), i64(1),
                                                   i64(1), 1, _ptr(one), 1, None) == 0
        want = Rt.grad.numpy().reshape(1, 49)
        assert np.allclose(one, want, rtol=1e-8, atol=1e-11 * np.abs(want).max()), (index, take_sqrt)
        as_f32 = np.empty((1, 49), dtype=np.float32)               # (dR in the beam dtype: the same values rounded)
        R32, g32 = R.astype(np.float32), np.array([0.7], dtype=np.float32)
        assert lib.chx_moment_entry_mapped_bwd_cpu(_ptr(g32), _ptr(mom_y), index, take_sqrt, _ptr(R32), _ptr(mom_x), i64(1), i64(1), i64(1), 0,
                                                   _ptr(as_f32), 0, None) == 0
        assert np.allclose(as_f32, one, rtol=2e-5, atol=1e-6 * np.abs(one).max())
    assert lib.chx_moment_entry_mapped_bwd_cpu(_ptr(np.array([0.7])), _ptr(mom_y), 1, 0, _ptr(R), _ptr(mom_x), i64(1), i64(1), i64(1), 1,
                                               _ptr(one), 1, None) == -1
    # ---- the maps of a run of scalar settings and their product: the batched builder twin element by element, the compose twin
    kinds = np.array([1, 2, 4, 1, 2], dtype=np.int32)              # drift, quadrupole, horizontal corrector, drift, quadrupole
    params = [[0.3], [0.12, 4.2, 0.05, 1e-4, -2e-4], [0.05, 1.5e-4], [0.7], [0.12, -3.9, 0.0, 0.0, 0.0]]
    for dtype, code in ((np.float64, 1), (np.float32, 0)):
        scalars = [[np.array([v], dtype=dtype) for v in p] for p in params]
        ptrs = (vp * (5 * 9))()
        for e, row in enumerate(scalars):
            for k, a in enumerate(row):
                ptrs[e * 9 + k] = a.ctypes.data
        energy = np.array([1.3e8], dtype=dtype)
        maps, comp, maps2 = np.empty((5, 7, 7), dtype=dtype), np.empty((7, 7), dtype=dtype), np.empty((5, 7, 7), dtype=dtype)
        assert lib.chx_run_build_compose_cpu(_ptr(kinds), ptrs, i64(5), _ptr(energy), dbl(oracle.ELECTRON_MASS_EV), dbl(-1.0), code, _ptr(maps),
                                             _ptr(comp), None) == 0
        assert lib.chx_build_rmatrix_scalars_cpu(_ptr(kinds), ptrs, i64(5), _ptr(energy), dbl(oracle.ELECTRON_MASS_EV), dbl(-1.0), code,
                                                 _ptr(maps2), None) == 0
        assert np.array_equal(maps, maps2)
        for e, (kind, p) in enumerate(zip(("drift", "quadrupole", "hcor", "drift", "quadrupole"), params)):
            want = oracle.build_rmatrix(kind, np.array([p], dtype=dtype).astype(np.float64), energy.astype(np.float64))[0]
            assert np.array_equal(maps[e], want.astype(dtype)), (dtype, e)
        prod = np.eye(7)
        for e in range(5):
            prod = maps[e].astype(np.float64) @ prod
        assert np.allclose(comp, prod.astype(dtype), rtol=4 * np.finfo(dtype).eps, atol=1e-30)
