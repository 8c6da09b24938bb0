import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc
import cheetah_amd as ca
which = sys.argv[1]
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
k1 = torch.nn.Parameter(rc.t(3.142, dt))
seg = ca.Segment([ca.Drift(rc.t(1.0, dt)), ca.Quadrupole(rc.t(0.2, dt), k1=k1, **kw), ca.Drift(rc.t(1.0, dt)),
                  ca.Screen(is_active=True, name="scr", **kw)])
beam = ca.ParticleBeam.from_parameters(num_particles=int(os.environ.get("NPART", "100000")), dtype=dt, device="cuda")
w = torch.nn.Parameter(torch.randn(64, 64, device="cuda"))
xin = torch.randn(64, 64, device="cuda")
def step():
    if which == "torch":
        loss = (xin @ w).square().sum(); loss.backward(); return loss
    if which == "fwd_nograd":
        with torch.no_grad():
            seg.track(beam); return seg.scr.get_read_beam().sigma_x
    if which == "fwd":
        seg.track(beam); return seg.scr.get_read_beam().sigma_x
    if which == "track_only":
        return seg.track(beam).particles
    seg.track(beam); loss = seg.scr.get_read_beam().sigma_x; loss.backward(); return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        k1.grad = None; w.grad = None; step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); k1.grad = None; w.grad = None
with torch.cuda.graph(g):
    out = step()
g.replay(); torch.cuda.synchronize()
print(which, "OK", float(out.reshape(-1)[0]))
