import torch, time
for mb in (1, 8, 64, 256):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64).pin_memory(); d = torch.empty(n, dtype=torch.float64, device="cuda")
    d.copy_(a, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): d.copy_(a, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"H2D pinned {mb} MB: {mb/1024/dt:.1f} GB/s", flush=True)
    t = time.perf_counter()
    for _ in range(10): a.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"D2H pinned {mb} MB: {mb/1024/dt:.1f} GB/s", flush=True)
