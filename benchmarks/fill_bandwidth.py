import torch, time
x = torch.empty(4096*100000*7, dtype=torch.float32, device="cuda")
for fn, name in ((lambda: x.zero_(), "zero_"), (lambda: x.fill_(1.5), "fill_")):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(name, f"{dt*1e3:.2f} ms", f"{x.numel()*4/dt/1e12:.2f} TB/s")
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
print("copy", f"{dt*1e3:.2f} ms", f"{2*x.numel()*4/dt/1e12:.2f} TB/s (read+write)")
