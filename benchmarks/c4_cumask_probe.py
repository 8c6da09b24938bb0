"""C4 with the side stream (Green chain, run maps) restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): does keeping
it off part of the chip help the deposit / FFT kernels it runs beside? usage: python benchmarks/c4_cumask_probe.py
Measured (MI355X, ROCm 7.0): no — 1.98 ms per track with the ordinary side stream, 4.1 ms with a mask of every 2nd / every 4th /
3 of 4 CUs, 12.8 ms with the lower half of every 32: a masked queue costs far more than the contention it avoids."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cheetah_amd as ca
from cheetah_amd.accelerator.space_charge_kick import SpaceChargeKick

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
g = 128
els = []
for i in range(10):
    els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1)),
            ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
seg = ca.Segment(els)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)


def bench(label):
    for _ in range(4):
        seg.track(beam)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        seg.track(beam)
    torch.cuda.synchronize(); print(label, "ms/track", round((time.perf_counter() - t0) * 50, 3))


bench("side stream on all CUs          ")
hip = ctypes.CDLL("libamdhip64.so")
for name, word in (("every 2nd CU", 0x55555555), ("every 4th CU", 0x11111111), ("lower half of each 32", 0x0000FFFF), ("3 of 4", 0x77777777)):
    mask = (ctypes.c_uint32 * 8)(*([word] * 8))
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
    if rc != 0:
        print(name, "hipExtStreamCreateWithCUMask ->", rc)
        continue
    ext = torch.cuda.ExternalStream(stream.value)
    SpaceChargeKick._side_streams.clear()
    SpaceChargeKick._side_stream = classmethod(lambda cls, device, ext=ext: ext)
    bench(f"side stream on {name:22s}")
