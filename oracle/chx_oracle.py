"""numpy front-end of the CPU oracle (oracle/chx_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``cheetah_amd`` never does (it fails loudly without its HIP
library instead of falling back to anything here).

The heavy per-particle work lives in the C file; the 3-D FFT convolution of the space-charge
solver (reference: cheetah/accelerator/space_charge_kick.py:293-322) is restated with
``numpy.fft`` here.  Parity against the real reference is pinned by ``tests/golden``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libchx_oracle.so")

KIND = {
    "identity": 0,
    "drift": 1,
    "quadrupole": 2,
    "dipole": 3,
    "hcor": 4,
    "vcor": 5,
    "ccor": 6,
    "cavity_sw": 7,
    "cavity_tw": 8,
    "solenoid": 9,
    "undulator": 10,
}
NUM_PARAMS = [0, 1, 5, 9, 2, 2, 3, 4, 4, 4, 4]

EPSILON_0 = 8.8541878188e-12  # scipy.constants.epsilon_0 (CODATA 2022)
SPEED_OF_LIGHT = 299792458.0
ELEMENTARY_CHARGE = 1.602176634e-19
ELECTRON_MASS_EV = 510998.95069
PROTON_MASS_EV = 938272089.4300001


def build(force: bool = False) -> str:
    """Compile the C oracle next to its source (gcc, OpenMP)."""
    src = os.path.join(_HERE, "chx_oracle.c")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "chx_oracle_nonlinear.inc")))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(
            # -mavx2 -mfma: fmaf()/fma() become single instructions (x86-64-v3, any EPYC/Xeon host);
            # -ffp-contract=off: no implicit contraction, the index arithmetic rounds op by op
            ["gcc", "-O3", "-mavx2", "-mfma", "-fPIC", "-shared", "-fopenmp", "-fvisibility=hidden",
             "-ffp-contract=off", "-o", _SO, src, "-lm"]
        )
    return _SO


_CPU_ABI_SO = os.path.join(_HERE, "libchx_cpu.so")


def build_cpu_abi(force: bool = False) -> str:
    """Compile the host twins of the core C-ABI entry points (include/chx_cpu.h; oracle/chx_cpu_abi.c includes the oracle's C)
    into oracle/libchx_cpu.so: what a downstream project links to exercise its binding without a GPU."""
    srcs = [os.path.join(_HERE, n) for n in ("chx_cpu_abi.c", "chx_oracle.c", "chx_oracle_nonlinear.inc")]
    hdrs = [os.path.join(_HERE, "..", "include", n) for n in ("chx.h", "chx_cpu.h")]
    newest = max(os.path.getmtime(f) for f in srcs + hdrs)
    if force or not os.path.exists(_CPU_ABI_SO) or os.path.getmtime(_CPU_ABI_SO) < newest:
        subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-fPIC", "-shared", "-fopenmp", "-fvisibility=hidden",
                               "-ffp-contract=off", "-o", _CPU_ABI_SO, srcs[0], "-lm"])
    return _CPU_ABI_SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _dt(a: np.ndarray) -> int:
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError(f"unsupported dtype {a.dtype}")


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


i64 = ctypes.c_int64


def build_rmatrix(kind, params, energy, mass_eV=ELECTRON_MASS_EV, n_charges=-1.0) -> np.ndarray:
    k = KIND[kind] if isinstance(kind, str) else int(kind)
    P = NUM_PARAMS[k]
    params = np.asarray(params, dtype=np.float64).reshape(-1, P) if P else np.zeros((1, 0))
    energy = np.asarray(energy, dtype=np.float64).reshape(-1)
    Bp, Be = max(params.shape[0], 1), energy.shape[0]
    B = max(Bp, Be)
    assert Bp in (1, B) and Be in (1, B)
    params, energy = _c(params), _c(energy)
    out = np.empty((B, 7, 7), dtype=np.float64)
    st = lib().chxo_build_rmatrix(k, _p(params), _p(energy), ctypes.c_double(mass_eV),
                                  ctypes.c_double(n_charges), i64(B), i64(Bp), i64(Be), _p(out))
    assert st == 0
    return out


def compose(maps) -> np.ndarray:
    """maps: list of (B?,7,7) arrays in element order; returns R_E ... R_1 (segment.py:534-543)."""
    maps = [np.asarray(m, dtype=np.float64).reshape(-1, 7, 7) for m in maps]
    B = max(m.shape[0] for m in maps)
    stack = _c(np.stack([np.broadcast_to(m, (B, 7, 7)) for m in maps]))
    out = np.empty((B, 7, 7), dtype=np.float64)
    st = lib().chxo_compose(_p(stack), i64(len(maps)), i64(B), i64(B), _p(out))
    assert st == 0
    return out


def apply(x, R, mode: int = 0) -> np.ndarray:
    """x (Bx,N,7), R (BR,7,7) of the same dtype -> (B,N,7) (element.py:182)."""
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    R = _c(np.asarray(R, dtype=x.dtype)).reshape(-1, 7, 7)
    Bx, BR = x.shape[0], R.shape[0]
    B = max(Bx, BR)
    out = np.empty((B, x.shape[1], 7), dtype=x.dtype)
    st = lib().chxo_apply(_p(x), _p(R), _p(out), i64(B), i64(Bx), i64(BR), i64(x.shape[1]), _dt(x), mode)
    assert st == 0
    return out


def track_elementwise(x, maps, out=None, tmp=None) -> np.ndarray:
    """E linear elements one after the other, no merging (segment.py:571-572); x (N,7), maps (E,7,7)
    of the same dtype. `out` / `tmp` may be passed to keep allocations out of a timed region."""
    x = _c(x).reshape(-1, 7)
    maps = _c(np.asarray(maps, dtype=x.dtype)).reshape(-1, 7, 7)
    out = np.empty_like(x) if out is None else out
    tmp = np.empty_like(x) if tmp is None else tmp
    st = lib().chxo_track_elementwise(_p(x), _p(maps), _p(out), _p(tmp), i64(maps.shape[0]), i64(x.shape[0]), _dt(x))
    assert st == 0
    return out


def cavity_coeffs(params, energy, mass_eV=ELECTRON_MASS_EV, n_charges=-1.0):
    params = _c(np.asarray(params, dtype=np.float64).reshape(-1, 4))
    energy = _c(np.asarray(energy, dtype=np.float64).reshape(-1))
    Bp, Be = params.shape[0], energy.shape[0]
    B = max(Bp, Be)
    coeffs = np.empty((B, 8), dtype=np.float64)
    e_out = np.empty((B,), dtype=np.float64)
    st = lib().chxo_cavity_coeffs(_p(params), _p(energy), ctypes.c_double(mass_eV),
                                  ctypes.c_double(n_charges), i64(B), i64(Bp), i64(Be), _p(coeffs), _p(e_out))
    assert st == 0
    return coeffs, e_out


def cavity_track(x, R, coeffs) -> np.ndarray:
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    coeffs = _c(np.asarray(coeffs, dtype=np.float64).reshape(-1, 8))
    B = coeffs.shape[0]
    R = _c(np.broadcast_to(np.asarray(R, dtype=x.dtype).reshape(-1, 7, 7), (B, 7, 7)))
    out = np.empty((B, x.shape[1], 7), dtype=x.dtype)
    st = lib().chxo_cavity_track(_p(x), _p(R), _p(coeffs), _p(out), i64(B), i64(x.shape[0]),
                                 i64(x.shape[1]), _dt(x))
    assert st == 0
    return out


def moments(x, w=None) -> dict:
    """All weighted first and second moments (particle_beam.py:1699-1943, statistics.py)."""
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    N = x.shape[1]
    if w is not None:
        w = _c(np.asarray(w, dtype=x.dtype)).reshape(-1, N)
    Bx, Bw = x.shape[0], (w.shape[0] if w is not None else 1)
    B = max(Bx, Bw)
    out = np.empty((B, 29), dtype=np.float64)
    st = lib().chxo_moments(_p(x), _p(w), i64(B), i64(Bx), i64(Bw), i64(N), _dt(x), _p(out))
    assert st == 0
    cov = np.zeros((B, 6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            cov[:, i, j] = cov[:, j, i] = out[:, 8 + k]
            k += 1
    return {"W": out[:, 0], "W2": out[:, 1], "mu": out[:, 2:8], "cov": cov, "raw": out}


class _CicArgs(ctypes.Structure):
    _fields_ = [
        ("ndim", ctypes.c_int32), ("cols", ctypes.c_int32 * 3), ("bins", ctypes.c_int32 * 3),
        ("B", i64), ("Bx", i64), ("Bq", i64), ("Bs", i64), ("Be", i64), ("Bsc", i64), ("Bsh", i64),
        ("N", i64), ("dtype", ctypes.c_int32), ("abs_charge", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("charge", ctypes.c_void_p), ("survival", ctypes.c_void_p),
        ("extent", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
    ]


def _cic_args(x, cols, bins, extent, charge, survival, scale, shift, abs_charge):
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    dt = x.dtype
    N = x.shape[1]
    nd = len(cols)
    keep = [x]
    extent = _c(np.asarray(extent, dtype=dt)).reshape(-1, nd, 2)
    charge = None if charge is None else _c(np.asarray(charge, dtype=dt)).reshape(-1, N)
    survival = None if survival is None else _c(np.asarray(survival, dtype=dt)).reshape(-1, N)
    scale = None if scale is None else _c(np.asarray(scale, dtype=dt)).reshape(-1, nd)
    shift = None if shift is None else _c(np.asarray(shift, dtype=dt)).reshape(-1, nd)
    keep += [extent, charge, survival, scale, shift]
    B = max(a.shape[0] for a in keep if a is not None)
    a = _CicArgs()
    a.ndim = nd
    for d in range(nd):
        a.cols[d] = int(cols[d])
        a.bins[d] = int(bins[d])
    a.B, a.Bx, a.Be, a.N = B, x.shape[0], extent.shape[0], N
    a.Bq = 1 if charge is None else charge.shape[0]
    a.Bs = 1 if survival is None else survival.shape[0]
    a.Bsc = 1 if scale is None else scale.shape[0]
    a.Bsh = 1 if shift is None else shift.shape[0]
    a.dtype = 0 if dt == np.float32 else 1
    a.abs_charge = int(bool(abs_charge))
    a.x = x.ctypes.data
    a.charge = None if charge is None else charge.ctypes.data
    a.survival = None if survival is None else survival.ctypes.data
    a.extent = extent.ctypes.data
    a.scale = None if scale is None else scale.ctypes.data
    a.shift = None if shift is None else shift.ctypes.data
    return a, keep, B, N, dt


def cic_deposit(x, cols, bins, extent, charge=None, survival=None, scale=None, shift=None,
                abs_charge=False) -> np.ndarray:
    """utils/cloud_in_cell.py:8-451 on positions taken from columns `cols` of the 7-vectors."""
    a, keep, B, N, dt = _cic_args(x, cols, bins, extent, charge, survival, scale, shift, abs_charge)
    grid = np.zeros((B, *[int(b) for b in bins]), dtype=dt)
    st = lib().chxo_cic_deposit(ctypes.byref(a), _p(grid))
    assert st == 0
    return grid


def cic_indices(x, cols, bins, extent, scale=None, shift=None):
    a, keep, B, N, dt = _cic_args(x, cols, bins, extent, None, None, scale, shift, False)
    idx = np.empty((B, N, len(cols)), dtype=np.int32)
    frac = np.empty((B, N, len(cols)), dtype=dt)
    st = lib().chxo_cic_indices(ctypes.byref(a), _p(idx), _p(frac))
    assert st == 0
    return idx, frac


def hist2d(x, edges_x, edges_y, charge=None, survival=None, shift=None, want_image=True):
    """screen.py:292-311 (torch.histogramdd semantics). Returns (image[B,ny,nx], ij[B,N,2])."""
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    dt, N = x.dtype, x.shape[1]
    edges_x, edges_y = _c(np.asarray(edges_x, dtype=dt)), _c(np.asarray(edges_y, dtype=dt))
    nx, ny = edges_x.shape[0] - 1, edges_y.shape[0] - 1
    charge = None if charge is None else _c(np.asarray(charge, dtype=dt)).reshape(-1, N)
    survival = None if survival is None else _c(np.asarray(survival, dtype=dt)).reshape(-1, N)
    shift = None if shift is None else _c(np.asarray(shift, dtype=dt)).reshape(-1, 2)
    Bs_ = [x.shape[0]] + [a.shape[0] for a in (charge, survival, shift) if a is not None]
    B = max(Bs_)
    image = np.zeros((B, ny, nx), dtype=dt) if want_image else None
    ij = np.empty((B, N, 2), dtype=np.int32)
    st = lib().chxo_hist2d(_p(x), _p(charge), _p(survival), _p(shift), _p(edges_x), _p(edges_y),
                           i64(B), i64(x.shape[0]), i64(1 if charge is None else charge.shape[0]),
                           i64(1 if survival is None else survival.shape[0]),
                           i64(1 if shift is None else shift.shape[0]), i64(N), nx, ny,
                           0 if dt == np.float32 else 1, _p(image), _p(ij))
    assert st == 0
    return image, ij


def to_xyz_pxpypz(x, energy, mass_eV=ELECTRON_MASS_EV):
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    energy = _c(np.asarray(energy, dtype=np.float64).reshape(-1))
    B = max(x.shape[0], energy.shape[0])
    out = np.empty((B, x.shape[1], 7), dtype=x.dtype)
    lib().chxo_to_xyz_pxpypz(_p(x), _p(energy), ctypes.c_double(mass_eV), i64(B), i64(x.shape[0]),
                             i64(energy.shape[0]), i64(x.shape[1]), _dt(x), _p(out))
    return out


def from_xyz_pxpypz(xp, energy, mass_eV=ELECTRON_MASS_EV):
    xp = _c(xp)
    xp = xp.reshape(-1, xp.shape[-2], 7)
    energy = _c(np.asarray(energy, dtype=np.float64).reshape(-1))
    B = max(xp.shape[0], energy.shape[0])
    out = np.empty((B, xp.shape[1], 7), dtype=xp.dtype)
    lib().chxo_from_xyz_pxpypz(_p(xp), _p(energy), ctypes.c_double(mass_eV), i64(B), i64(xp.shape[0]),
                               i64(energy.shape[0]), i64(xp.shape[1]), _dt(xp), _p(out))
    return out


def igf(cell_scaled, bins) -> np.ndarray:
    """Integrated Green function on the doubled grid; cell_scaled = (hx, hy, htau*gamma) per row."""
    cell_scaled = _c(np.asarray(cell_scaled, dtype=np.float64).reshape(-1, 3))
    B = cell_scaled.shape[0]
    b = (ctypes.c_int32 * 3)(*[int(v) for v in bins])
    G = np.empty((B, 2 * bins[0], 2 * bins[1], 2 * bins[2]), dtype=np.float64)
    lib().chxo_sc_igf(_p(cell_scaled), i64(B), b, _p(G))
    return G


def aperture_mask(x, survival, x_max, y_max, shape="rectangular") -> np.ndarray:
    """aperture.py:104-128 in numpy, every operation rounded in x.dtype like the reference's tensor ops:
    rectangular -x_max < x < x_max and -y_max < y < y_max (strict); elliptical x^2/x_max^2 + y^2/y_max^2 <= 1."""
    x = np.asarray(x)
    dt = x.dtype
    px, py = x[..., 0], x[..., 2]
    xm = np.asarray(x_max, dtype=dt)[..., None]
    ym = np.asarray(y_max, dtype=dt)[..., None]
    if shape == "rectangular":
        inside = np.logical_and(np.logical_and(px > -xm, px < xm), np.logical_and(py > -ym, py < ym))
    elif shape == "elliptical":
        inside = (np.square(px) / np.square(xm) + np.square(py) / np.square(ym)) <= dt.type(1.0)
    else:
        raise AssertionError(f"Unknown aperture shape {shape}")
    return (np.asarray(survival, dtype=dt) * inside).astype(dt)


DKD_KIND = {"drift": 0, "quadrupole": 1, "dipole": 2, "tdc": 3}
DKD_NUM_PARAMS = {0: 1, 1: 5, 2: 9, 3: 7}
T_KIND = {"drift": 0, "quadrupole": 1, "dipole": 2, "sextupole": 3}
T_NUM_PARAMS = {0: 1, 1: 5, 2: 9, 3: 5}


def dkd_track(kind, x, params, energy, mass_eV=ELECTRON_MASS_EV, n_charges=-1.0, num_steps=1, fringe_at=3):
    """Drift-kick-drift (Bmad-X) tracking of one element; x (Bx,N,7); params (Bp,P) and energy (Be,) are cast to
    x.dtype like the reference's element buffers. Returns (x_out (B,N,7), ref_energy (B,) float64)."""
    k = DKD_KIND[kind] if isinstance(kind, str) else int(kind)
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    params = _c(np.asarray(params, dtype=x.dtype).reshape(-1, DKD_NUM_PARAMS[k]))
    energy = _c(np.asarray(energy, dtype=x.dtype).reshape(-1))
    Bx, Bp, Be = x.shape[0], params.shape[0], energy.shape[0]
    B = max(Bx, Bp, Be)
    assert Bx in (1, B) and Bp in (1, B) and Be in (1, B)
    out = np.empty((B, x.shape[1], 7), dtype=x.dtype)
    e_out = np.empty((B,), dtype=np.float64)
    st = lib().chxo_dkd_track(k, _p(x), _p(params), _p(energy), ctypes.c_double(mass_eV), ctypes.c_double(n_charges),
                              int(num_steps), int(fringe_at), i64(B), i64(Bx), i64(Bp), i64(Be), i64(x.shape[1]),
                              _dt(x), _p(out), _p(e_out))
    assert st == 0
    return out, e_out


def build_ttensor(kind, params, energy, mass_eV=ELECTRON_MASS_EV) -> np.ndarray:
    """Second-order transfer tensor (B,7,7,7), float64 (track_methods.py:80-296 + element dressing)."""
    k = T_KIND[kind] if isinstance(kind, str) else int(kind)
    params = _c(np.asarray(params, dtype=np.float64).reshape(-1, T_NUM_PARAMS[k]))
    energy = _c(np.asarray(energy, dtype=np.float64).reshape(-1))
    Bp, Be = params.shape[0], energy.shape[0]
    B = max(Bp, Be)
    assert Bp in (1, B) and Be in (1, B)
    out = np.empty((B, 7, 7, 7), dtype=np.float64)
    st = lib().chxo_build_ttensor(k, _p(params), _p(energy), ctypes.c_double(mass_eV), i64(B), i64(Bp), i64(Be),
                                  _p(out))
    assert st == 0
    return out


def apply_second_order(x, T) -> np.ndarray:
    """x_out_i = sum_jk T_ijk x_j x_k (element.py:207-217); x (Bx,N,7), T (BT,7,7,7)."""
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    T = _c(np.asarray(T, dtype=np.float64).reshape(-1, 343))
    Bx, BT = x.shape[0], T.shape[0]
    B = max(Bx, BT)
    out = np.empty((B, x.shape[1], 7), dtype=x.dtype)
    st = lib().chxo_apply_second_order(_p(x), _p(T), _p(out), i64(B), i64(Bx), i64(BT), i64(x.shape[1]), _dt(x))
    assert st == 0
    return out


def space_charge_kick(x, energy, charge, survival, effect_length, grid_shape=(32, 32, 32),
                      grid_extent=(3.0, 3.0, 3.0), mass_eV=ELECTRON_MASS_EV, details: bool = False):
    """SpaceChargeKick.track (space_charge_kick.py:477-586) for x (B,N,7); fp64 grid solve.

    Index/weight arithmetic of the deposit is done in x.dtype (like the reference), the Poisson
    solve in float64.
    """
    x = _c(x)
    x = x.reshape(-1, x.shape[-2], 7)
    dt = x.dtype
    B, N = x.shape[0], x.shape[1]
    energy = np.broadcast_to(np.asarray(energy, dtype=np.float64).reshape(-1), (B,)).copy()
    charge = np.broadcast_to(np.asarray(charge, dtype=dt).reshape(-1, N), (B, N)).copy()
    survival = np.broadcast_to(np.asarray(survival, dtype=dt).reshape(-1, N), (B, N)).copy()
    L = np.broadcast_to(np.asarray(effect_length, dtype=np.float64).reshape(-1), (B,))
    g = [int(v) for v in grid_shape]
    mom = moments(x, survival)
    sig = np.sqrt(np.stack([mom["cov"][:, 0, 0], mom["cov"][:, 2, 2], mom["cov"][:, 4, 4]], axis=-1))
    # the reference forms sigma, grid_dimensions and cell_size in the beam dtype
    sig = sig.astype(dt)
    half = (np.asarray(grid_extent, dtype=dt) * sig).astype(dt)           # :531-538
    cell = (2 * half / np.asarray(g, dtype=dt)).astype(dt)                # :539-547
    gamma = energy / mass_eV
    beta = np.sqrt(1.0 - 1.0 / gamma**2)
    dtk = L / (SPEED_OF_LIGHT * beta)                                     # :548-550
    scale = np.stack([np.ones(B), np.ones(B), -beta], axis=-1).astype(dt)  # z = tau * -beta
    extent = np.stack([-half, half], axis=-1)                             # (B,3,2)
    rho = cic_deposit(x, (0, 2, 4), g, extent, charge=charge, survival=survival, scale=scale)
    inv_vol = 1.0 / np.prod(cell.astype(np.float64), axis=-1)
    rho64 = rho.astype(np.float64) * inv_vol[:, None, None, None]        # :144-146
    pad = np.zeros((B, 2 * g[0], 2 * g[1], 2 * g[2]))
    pad[:, : g[0], : g[1], : g[2]] = rho64
    cell_scaled = cell.astype(np.float64).copy()
    cell_scaled[:, 2] = (cell[:, 2] * gamma.astype(dt)).astype(dt)        # :170-176 (beam dtype product)
    G = igf(cell_scaled, g)
    phi = np.fft.irfftn(np.fft.rfftn(pad, axes=(1, 2, 3)) * np.fft.rfftn(G, axes=(1, 2, 3)),
                        s=pad.shape[1:], axes=(1, 2, 3)) / (4 * np.pi * EPSILON_0)  # :306-316
    phi = np.ascontiguousarray(phi[:, : g[0], : g[1], : g[2]])
    F = np.empty((B, g[0], g[1], g[2], 3))
    b3 = (ctypes.c_int32 * 3)(*g)
    cell64 = _c(cell.astype(np.float64))
    gam64 = _c(gamma)
    lib().chxo_sc_gradient(_p(phi), _p(cell64), _p(gam64), i64(B), b3, _p(F))
    out = np.empty_like(x)
    forces = np.empty((B, N, 3))
    half64, dtk64, en64 = _c(half.astype(np.float64)), _c(dtk), _c(energy)
    lib().chxo_sc_gather_kick(_p(x), _p(F), _p(half64), _p(cell64), _p(en64), _p(dtk64),
                              ctypes.c_double(mass_eV), i64(B), i64(B), i64(B), i64(N), b3, _dt(x),
                              _p(out), _p(forces))
    if details:
        return out, {"rho": rho, "G": G, "phi": phi, "F": F, "forces": forces, "half": half,
                     "cell": cell, "dt": dtk, "sigma": sig}
    return out
