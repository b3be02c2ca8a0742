"""Known-volume kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: a 1 GiB fill (writes 1 GiB),
a 1 GiB copy (reads 1 GiB, writes 1 GiB), with 8-byte elements like the solver's loads and stores."""
import torch
n = (1 << 30) // 8
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
a.fill_(1.0)          # FillFunctor kernel: write 1 GiB
torch.cuda.synchronize()
b.copy_(a)            # copy kernel / blit: read 1 GiB + write 1 GiB
torch.cuda.synchronize()
c = a + b             # elementwise add: read 2 GiB, write 1 GiB
torch.cuda.synchronize()
print("ok", float(c[0]))
