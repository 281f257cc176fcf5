"""Streaming floor for the decimator's traffic shape on this box: read 2 n floats, write n floats (config 3: n = 60 M),
plus a 1 : 1 copy and a read-only pass of the same bytes for comparison."""
import time
import torch
import sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000      # 60 M: config 3's sizes (partly served by the 256 MB Infinity Cache when looped)
x = torch.randn(2 * n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
z = torch.empty(2 * n, dtype=torch.float32, device="cuda")
for name, fn, nbytes in [("add(x[:n], x[n:]) -> y   (2 : 1)", lambda: torch.add(x[:n], x[n:], out=y), 12 * n),
                         ("copy 2n                 (1 : 1)", lambda: z.copy_(x), 16 * n),
                         ("copy n                  (1 : 1)", lambda: y.copy_(x[:n]), 8 * n),
                         ("sum 2n               (read only)", lambda: x.sum(), 8 * n)]:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt * 1e3:.3f} ms  {nbytes / dt / 1e12:.2f} TB/s", flush=True)
