import sys, os, torch, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv=[sys.argv[0]]
import run_configs as rc
import cheetah_amd as ca
for dt in (torch.float32,):
    seg=rc.ares_subcell(dt, rc.t(8.2, dt))
    beam=ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14,dt), beta_y=rc.t(42.0,dt), num_particles=10_000, dtype=dt, device="cuda")
    pbeam=ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14,dt), beta_y=rc.t(42.0,dt), dtype=dt, device="cuda")
    seg.AREABSCR1.is_active=True
    acts=torch.randn(300,5,device="cuda",dtype=dt)
    def step(i, b):
        a=acts[i%300]
        seg.AREAMQZM1.k1=a[0]*10; seg.AREAMQZM2.k1=a[1]*10; seg.AREAMCVM1.angle=a[2]*1e-4
        seg.AREAMQZM3.k1=a[3]*10; seg.AREAMCHM1.angle=a[4]*1e-4
        out=seg.track(b)
        return seg.AREABSCR1.reading
    for name,b in (("ParticleBeam",beam),("ParameterBeam",pbeam)):
        for i in range(20): step(i,b)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for i in range(500): step(i,b)
        torch.cuda.synchronize(); print(name, "RL-style step (5 settings + track + reading):", (time.perf_counter()-t0)/500*1e3, "ms")
    pr=cProfile.Profile(); pr.enable()
    for i in range(300): step(i,beam)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
