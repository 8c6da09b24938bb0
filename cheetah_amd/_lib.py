"""ctypes binding of libchx.so (include/chx.h).

The library is the ONLY compute backend of this package: if it is missing or cannot be loaded the
import of any tracking entry point fails loudly — there is no CPU / eager fallback.

PyTorch is imported first on purpose: libchx.so needs ``libamdhip64.so.7``; PyTorch-ROCm ships and
loads its own copy with the same soname, and the dynamic loader then binds libchx to that copy, so
streams and device pointers handed over from torch are valid inside the kernels.
"""

from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede loading libchx.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libchx.so")

c_void_p, c_int, c_i64, c_double, c_size_t = (
    ctypes.c_void_p,
    ctypes.c_int,
    ctypes.c_int64,
    ctypes.c_double,
    ctypes.c_size_t,
)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_u8_p = ctypes.POINTER(ctypes.c_uint8)
c_vpp = ctypes.POINTER(ctypes.c_void_p)
c_double_p = ctypes.POINTER(ctypes.c_double)


class ChxError(RuntimeError):
    """A libchx entry point returned a negative chx_status."""


class CicArgs(ctypes.Structure):
    """struct chx_cic_args (include/chx.h)."""

    _fields_ = [
        ("ndim", ctypes.c_int32),
        ("cols", ctypes.c_int32 * 3),
        ("bins", ctypes.c_int32 * 3),
        ("grid_strides", ctypes.c_int64 * 3),
        ("grid_batch_stride", ctypes.c_int64),
        ("B", c_i64), ("Bx", c_i64), ("Bq", c_i64), ("Bs", c_i64), ("Be", c_i64),
        ("Bsc", c_i64), ("Bsh", c_i64), ("N", c_i64),
        ("dtype", ctypes.c_int32),
        ("abs_charge", ctypes.c_int32),
        ("x", c_void_p), ("charge", c_void_p), ("survival", c_void_p), ("extent", c_void_p),
        ("scale", c_void_p), ("shift", c_void_p), ("grid", c_void_p),
    ]


class LatticeScreen(ctypes.Structure):
    """struct chx_lattice_screen (include/chx.h): one active Screen's output buffers of a stretch call."""

    _fields_ = [
        ("rows", c_void_p), ("charges", c_void_p), ("survival", c_void_p), ("energy", c_void_p), ("s", c_void_p),
        ("image", c_void_p), ("image_bytes", c_i64), ("map", c_void_p), ("element_maps", c_void_p),
        ("mu", c_void_p), ("cov", c_void_p), ("geom", c_void_p), ("shift", c_void_p), ("total_charge", c_void_p),
        ("total_charge_out", c_void_p), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("mom_partials", c_void_p),
    ]


class Hist2dArgs(ctypes.Structure):
    """struct chx_hist2d_args (include/chx.h)."""

    _fields_ = [
        ("B", c_i64), ("Bx", c_i64), ("Bq", c_i64), ("Bs", c_i64), ("Bsh", c_i64), ("N", c_i64),
        ("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("x", c_void_p), ("charge", c_void_p), ("survival", c_void_p), ("shift", c_void_p),
        ("edges_x", c_void_p), ("edges_y", c_void_p), ("image", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/chx.h declares
SIGNATURES = {
    "chx_kind_num_params": (c_int, [c_int]),
    "chx_abi_version": (c_int, []),
    "chx_status_string": (ctypes.c_char_p, [c_int]),
    "chx_build_rmatrix": (c_int, [c_int, c_void_p, c_void_p, c_double, c_double, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_build_rmatrix_vjp": (c_int, [c_int, c_void_p, c_void_p, c_double, c_double, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_compose_maps": (c_int, [c_vpp, c_u8_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_compose_maps_vjp_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_compose_maps_vjp": (c_int, [c_vpp, c_u8_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_compose_prefix": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_run_state_bytes": (c_size_t, [c_i64]),
    "chx_run_map": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "chx_run_track": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                              c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "chx_apply_affine7": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "chx_apply_bwd_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_apply_affine7_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_size_t, c_void_p]),
    "chx_track_elementwise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "chx_track_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "chx_cavity_coeffs": (c_int, [c_void_p, c_void_p, c_double, c_double, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_cavity_track": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "chx_cavity_track_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p,
                                     c_size_t, c_void_p]),
    "chx_moments_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_moment_sums": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_moment_centred": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_moment_finalize": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_moments": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_moments_entry": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "chx_moments_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_moments_bwd_w": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "chx_moments_mapped_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_moment_entry": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chx_moment_entry_mapped_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int,
                                            c_void_p, c_int, c_void_p]),
    "chx_cic_deposit": (c_int, [ctypes.POINTER(CicArgs), c_void_p]),
    "chx_cic_deposit_mapped": (c_int, [ctypes.POINTER(CicArgs), c_void_p, c_i64, c_void_p]),
    "chx_cic_sorted_workspace_bytes": (c_size_t, [ctypes.POINTER(CicArgs)]),
    "chx_cic_deposit_sorted": (c_int, [ctypes.POINTER(CicArgs), c_void_p, c_size_t, c_void_p]),
    "chx_cic_deposit_sorted_overwrite": (c_int, [ctypes.POINTER(CicArgs), c_void_p, c_size_t, c_void_p]),
    "chx_cic_indices": (c_int, [ctypes.POINTER(CicArgs), c_void_p, c_void_p, c_void_p]),
    "chx_cic_deposit_bwd": (c_int, [ctypes.POINTER(CicArgs), c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_hist2d": (c_int, [ctypes.POINTER(Hist2dArgs), c_void_p]),
    "chx_hist2d_indices": (c_int, [ctypes.POINTER(Hist2dArgs), c_void_p, c_void_p]),
    "chx_sc_igf_workspace_bytes": (c_size_t, [c_i64, c_i32_p]),
    "chx_sc_igf": (c_int, [c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_i64, c_void_p, c_size_t, c_void_p]),
    "chx_sc_igf_from_table": (c_int, [c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_i64, c_void_p]),
    "chx_sc_pruned_supported": (c_int, [c_i32_p, c_int]),
    "chx_sc_igf_table": (c_int, [c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p]),
    "chx_sc_green_workspace_bytes": (c_size_t, [c_i64, c_i32_p, c_int]),
    "chx_sc_green_fast_workspace_bytes": (c_size_t, [c_i64, c_i32_p, c_int]),
    "chx_sc_green_spectrum_fast": (c_int, [c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_green_spectrum": (c_int, [c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_convolve_workspace_bytes": (c_size_t, [c_i64, c_i32_p, c_int]),
    "chx_sc_convolve": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_fft_plan_create": (c_int, [c_i64, c_i32_p, c_int, c_vpp]),
    "chx_sc_fft_plan_destroy": (c_int, [c_void_p]),
    "chx_sc_fft_exec": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "chx_sc_spectral_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "chx_sc_phi_halo_elements": (c_size_t, [c_i64, c_i32_p]),
    "chx_sc_convolve_halo": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_convolve_halo_after": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                           c_void_p]),
    "chx_sc_gather_kick_phi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i64,
                                       c_i64, c_i64, c_i32_p, c_int, c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_sc_gradient": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_i64, c_int, c_void_p, c_void_p]),
    "chx_sc_gather_kick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64, c_i64, c_i32_p, c_int, c_void_p, c_void_p]),
    "chx_sc_gather_kick_mapped": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64,
                                          c_i64, c_i32_p, c_int, c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_cavity_prepare_scalars": (c_int, [c_void_p, c_void_p, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "chx_cavity_track_scalars_workspace_bytes": (c_size_t, []),
    "chx_cavity_track_scalars": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_i64, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_lattice_state_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_lattice_prepare": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "chx_lattice_track": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                                  c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_lattice_diag_workspace_bytes": (c_size_t, [c_i64, c_i64, c_i64]),
    "chx_lattice_track_diag": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                                       c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_i64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_lattice_track_screens": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t, c_void_p,
                                          c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_i64, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_i64, c_void_p]),
    "chx_table_store_max_words": (c_i64, []),
    "chx_table_store": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_lattice_moment_blocks": (c_i64, [c_i64, c_i64]),
    "chx_lattice_screen_moments": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "chx_lattice_prepare_screens": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_double, c_double, c_int, c_void_p,
                                            c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "chx_screen_extent": (c_int, [c_void_p, ctypes.c_int32, ctypes.c_int32, c_int, c_void_p, c_void_p]),
    "chx_parameter_lattice_track_screens": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t,
                                                    c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    "chx_lattice_state_bytes_batched": (c_size_t, [c_i64, c_i64, c_i64]),
    "chx_lattice_prepare_batched": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_lattice_prepare_rows": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_double, c_double, c_int, c_void_p,
                                         c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_parameter_lattice_track": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_size_t,
                                            c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_copy_arrays": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int32, c_void_p]),
    "chx_to_xyz_pxpypz": (c_int, [c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_from_xyz_pxpypz": (c_int, [c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_parameter_track": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_parameter_track_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "chx_screen_gaussian": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, ctypes.c_int32, ctypes.c_int32, c_int, c_int, c_void_p, c_void_p]),
    "chx_track_moments_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_track_moments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int,
                                  c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_geometry": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_i64, c_i64, c_i64, c_i64, c_i64,
                                c_i32_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "chx_aperture_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p,
                                  c_void_p]),
    "chx_dkd_num_params": (c_int, [c_int]),
    "chx_dkd_track": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_double, ctypes.c_int32, ctypes.c_int32,
                              c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_dkd_track_p": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_double, ctypes.c_int32, ctypes.c_int32, c_i64, c_i64,
                                c_i64, c_i64, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_dkd_chain": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_double, c_double, c_i64,
                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_dkd_chain_mixed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_double,
                                    c_double, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_dkd_energy_chain": (c_int, [c_void_p, c_i64, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_t_num_params": (c_int, [c_int]),
    "chx_build_ttensor": (c_int, [c_int, c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "chx_apply_second_order": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "chx_second_order_chain": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_second_order_chain_mixed": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p]),
    "chx_dkd_bwd_partials_count": (c_i64, [c_int, c_i64, c_i64]),
    "chx_dkd_track_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, ctypes.c_int32,
                                  ctypes.c_int32, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_build_ttensor_vjp": (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p,
                                      c_void_p, c_void_p]),
    "chx_special": (c_int, [c_int, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_second_order_bwd_partials_count": (c_i64, [c_i64]),
    "chx_apply_second_order_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64,
                                           c_int, c_void_p]),
    "chx_sc_igf_table_grad": (c_int, [c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p]),
    "chx_sc_gradient_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p]),
    "chx_sc_gather_kick_bwd_partials_count": (c_i64, [c_i64, c_i64]),
    "chx_sc_gather_kick_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64,
                                       c_i64, c_i64, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_sc_beam_geometry_workspace_bytes": (c_size_t, [c_i64, c_i64]),
    "chx_sc_beam_geometry": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_i64, c_i64, c_i64,
                                     c_i64, c_i64, c_i64, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_sc_kick_workspace_bytes": (c_size_t, [c_i64, c_i64, c_i32_p, c_int]),
    "chx_sc_kick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i64, c_i64, c_i64,
                            c_i64, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_i64]),
    "chx_sc_tile_state_bytes": (c_size_t, [c_i64, c_i32_p, c_int]),
    "chx_sc_kick_sorted_workspace_bytes": (c_size_t, [c_i64, c_i32_p, c_int]),
    "chx_sc_kick_sorted": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i32_p, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_sc_kick_sorted_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64, c_i32_p, c_int,
                                         c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p, ctypes.c_int32, c_vpp, c_void_p,
                                         c_void_p]),
    "chx_sc_kick_sorted_finish": (c_int, [c_void_p, c_void_p, c_double, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                          c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "chx_sc_tile_beam_moments": (c_int, [c_void_p, c_size_t, c_i64, c_i32_p, c_int, c_void_p, c_void_p]),
    "chx_sc_partials_moments": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "chx_sc_geometry_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_i64, c_i64, c_i64, c_i64, c_i64,
                                      c_i32_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, ctypes.c_int32, c_void_p]),
    "chx_sc_beam_geometry_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_i64, c_i64,
                                           c_i64, c_i64, c_i64, c_i64, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "chx_sc_geometry_from_partials": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_double, c_double, c_i32_p, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chx_sc_tile_sort": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "chx_sc_tile_deposit": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                    c_void_p]),
    "chx_sc_tile_deposit_acc": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_size_t, c_int, c_vpp, c_void_p]),
    "chx_sc_convolve_halo_consume": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                             c_void_p]),
    "chx_sc_tile_gather_kick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_i64,
                                        c_i32_p, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "chx_kde_values": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64, c_i64,
                               c_i64, c_i64, c_i64, c_i64, ctypes.c_int32, c_int, c_void_p, c_void_p]),
    "chx_merge_moments": (c_int, [c_void_p, ctypes.c_int32, c_i64, c_void_p, c_void_p]),
    "chx_run_vjp_workspace_bytes": (c_size_t, [c_i64]),
    "chx_run_vjp": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_size_t, c_void_p]),
    "chx_run_vjp_entry_workspace_bytes": (c_size_t, [c_i64]),
    "chx_run_vjp_entry": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_run_vjp_masked": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "chx_run_map_batched_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int]),
    "chx_run_map_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_int, c_double, c_double, c_int, c_void_p, c_size_t,
                                    c_void_p, c_void_p]),
    "chx_run_build_compose": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p,
                                      c_void_p]),
    "chx_build_rmatrix_scalars": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libchx.so (once). Raises ImportError with build instructions when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP library is the only backend of cheetah_amd. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C cheetah_amd/csrc` (needs hipcc, --offload-arch=gfx950)."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.chx_abi_version() != 9:  # CHX_ABI_VERSION of include/chx.h
            raise ImportError("libchx.so ABI version mismatch; rebuild the library")
        _lib = handle
    return _lib


def check(status: int, what: str = "libchx") -> None:
    if status != 0:
        msg = lib().chx_status_string(status).decode()
        raise ChxError(f"{what} failed: {msg} (status {status})")


_host = None


def host():
    """`cheetah_amd._chxhost` (csrc/chx_host.c), bound to libchx's chx_run_track: the host step of a merged Segment.track in C.
    Built by the same Makefile as libchx.so; missing = a broken build, there is no slower stand-in."""
    global _host
    if _host is None:
        try:
            from . import _chxhost
        except ImportError as exc:  # pragma: no cover - broken build
            raise ImportError("cheetah_amd._chxhost is not built: run `make -C cheetah_amd/csrc` "
                              "(or `python -c 'import __graft_entry__ as g; g.build()'`)") from exc
        fn = lib().chx_run_track
        _chxhost.bind(ctypes.cast(fn, ctypes.c_void_p).value, torch.empty_like, torch._C._cuda_getCurrentRawStream, ChxError)
        _chxhost.bind_lattice(ctypes.cast(lib().chx_lattice_track_diag, ctypes.c_void_p).value)
        _chxhost.bind_table_store(ctypes.cast(lib().chx_table_store, ctypes.c_void_p).value, int(lib().chx_table_store_max_words()))
        _host = _chxhost
    return _host


_torch_host = None


def torch_host():
    """`cheetah_amd._chxtorch` (csrc/chx_torch_host.cpp), bound to libchx's stretch entry points: the host step of a stretch with
    active Screens in C++ against ATen. Built by the same Makefile as libchx.so; missing = a broken build."""
    global _torch_host
    if _torch_host is None:
        try:
            from . import _chxtorch
        except ImportError as exc:  # pragma: no cover - broken build
            raise ImportError("cheetah_amd._chxtorch is not built: run `make -C cheetah_amd/csrc` "
                              "(or `python -c 'import __graft_entry__ as g; g.build()'`)") from exc
        h = lib()
        names = ("chx_lattice_track_screens", "chx_parameter_lattice_track_screens", "chx_run_build_compose", "chx_run_vjp_masked", "chx_run_vjp_entry",
                 "chx_run_vjp_entry_workspace_bytes",
                 "chx_run_vjp_workspace_bytes", "chx_apply_affine7_bwd", "chx_apply_bwd_workspace_bytes", "chx_moments_entry",
                 "chx_moments_workspace_bytes", "chx_moment_entry", "chx_moment_entry_mapped_bwd", "chx_lattice_moment_blocks",
                 "chx_lattice_screen_moments")
        _chxtorch.bind({n: ctypes.cast(getattr(h, n), ctypes.c_void_p).value for n in names}, ChxError)
        _torch_host = _chxtorch
    return _torch_host
