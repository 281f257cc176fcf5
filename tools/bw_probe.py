import torch, time
x = torch.empty(1024*1536*256, dtype=torch.float32, device="cuda")   # 1.6 GB
y = torch.empty_like(x)
for name, fn in [("fill_", lambda: x.fill_(1.5)), ("copy_", lambda: y.copy_(x)), ("read(sum)", lambda: x.sum())]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(name, f"{dt*1e3:.3f} ms", f"{x.numel()*4/dt/1e12:.2f} TB/s (x2 for copy)")
out = torch.empty((1024, 128, 3000), dtype=torch.float32, device="cuda")
for name, fn, nbytes in [("strided fill out[:, :, 528:]", lambda: out[:, :, 528:].fill_(1.5), 1024*128*2472*4),
                         ("row-block fill out[:, 64:, :]", lambda: out[:, 64:, :].fill_(1.5), 1024*64*3000*4),
                         ("full fill", lambda: out.fill_(1.5), out.numel()*4)]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(name, f"{dt*1e3:.3f} ms", f"{nbytes/dt/1e12:.2f} TB/s")
