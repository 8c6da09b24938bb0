import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from benchmarks import run_configs as rc
import cheetah_amd as ca
from cheetah_amd import _lib
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
k1 = torch.nn.Parameter(rc.t(3.142, dt))
seg = ca.Segment([ca.Drift(rc.t(1.0, dt)), ca.Quadrupole(rc.t(0.2, dt), k1=k1, **kw), ca.Drift(rc.t(1.0, dt)), ca.Screen(is_active=True, name="scr", **kw)])
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")
lib = ctypes.CDLL(os.path.join("/root/repo/cheetah_amd", "libchx.so"))
buf = (ctypes.c_ulonglong * 64)()
for it in range(6):
    k1.grad = None
    seg.track(beam)
    seg.scr.get_read_beam().sigma_x.backward()
    torch.cuda.synchronize()
    lib.chx_debug_read(buf, 64)
    v = list(buf[:5])
    print("vjp:", [v[i + 1] - v[i] for i in range(4)], "quad:", [buf[i + 1] - buf[i] for i in range(10, 19)])
