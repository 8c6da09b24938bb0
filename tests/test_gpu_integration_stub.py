"""The reference-side binding documented in INTEGRATION.md section B is EXECUTED: the Python block a maintainer would add as
`cheetah/utils/chx.py` is extracted from the document, run against the built libchx.so, and its three functions —
`apply_affine7` (element.py:182), `base_rmatrix` (track_methods.py:17-77) and `moments` (particle_beam.py:1699-1943) — are
checked against the CPU oracle. A stub that drifts from the header fails here, not at a downstream user."""
import os
import re
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_module():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "cheetah/utils/chx.py" in b)
    assert 'ctypes.CDLL("libchx.so")' in stub
    import cheetah_amd  # noqa: F401  (PyTorch-ROCm first: libchx binds to its libamdhip64)
    from cheetah_amd import _lib

    stub = stub.replace('ctypes.CDLL("libchx.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    mod = types.ModuleType("chx_stub")
    exec(compile(stub, "INTEGRATION.md#B", "exec"), mod.__dict__)
    return mod


def test_documented_binding_runs_and_matches_the_oracle(oracle):
    chx = _stub_module()
    torch.manual_seed(0)
    species = types.SimpleNamespace(mass_eV=oracle.ELECTRON_MASS_EV, num_elementary_charges=-1.0)
    for dt, rtol in ((torch.float64, 1e-12), (torch.float32, 2e-6)):
        kw = {"dtype": dt, "device": "cuda"}
        params = torch.tensor([[0.2, 4.2, 0.05, 1e-4, -2e-4], [0.2, -4.2, 0.0, 0.0, 0.0]], **kw)
        energy = torch.tensor([1e8], **kw)
        R = chx.base_rmatrix(2, params, energy, species)                       # CHX_QUADRUPOLE
        ref_R = oracle.build_rmatrix("quadrupole", params.double().cpu().numpy(), [1e8])
        assert np.allclose(R.cpu().numpy(), ref_R, rtol=rtol, atol=1e-12 if dt == torch.float64 else 1e-7)
        x = torch.randn(1, 5000, 7, **kw) * 1e-3
        x[..., 6] = 1.0
        y = chx.apply_affine7(x, R)
        assert y.shape == (2, 5000, 7)
        assert np.array_equal(y.cpu().numpy(), oracle.apply(x.cpu().numpy(), R.cpu().numpy(), mode=1))   # the kernels' fma chain
        w = torch.rand(2, 5000, **kw)
        mom = chx.moments(y, w)
        ref = oracle.moments(y.cpu().numpy(), w.cpu().numpy())["raw"]
        assert np.allclose(mom.cpu().numpy(), ref, rtol=1e-9 if dt == torch.float64 else 1e-5, atol=1e-30)
    # error behaviour: a negative status becomes the stub's RuntimeError, nothing crosses the ABI as an exception
    with pytest.raises(RuntimeError, match="chx_apply_affine7"):
        chx.apply_affine7(torch.zeros(3, 10, 7, device="cuda"), torch.zeros(2, 7, 7, device="cuda"))
