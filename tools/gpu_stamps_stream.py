"""Per-phase s_memtime shares of the large-site kernel on the configs[4] bench leg; needs a diagnostic build:
build_hip_library(extra_flags=["-DACNQP_STAMPS"], out=".../libacn_qp_hip_stamps.so")
    python tools/gpu_stamps_stream.py [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ACNQP_LIBRARY"] = os.path.join(ROOT, "adacharge_amd", "lib", "libacn_qp_hip_stamps.so")
import numpy as np, torch
import bench
from adacharge_amd.backend import DeviceBatch, SiteHandle, load_library
from adacharge_amd.builder import ProblemBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch, opts, _, note = bench.other_workloads()["cfg4_synth512_T48_b2048"]()
if B < batch.B:
    batch = batch.take(np.arange(B)) if hasattr(batch, "take") else batch
h = SiteHandle(batch.site, 0)
dev = DeviceBatch(batch, "cuda:0")
st = torch.cuda.current_stream().cuda_stream
h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
ms = h.last_kernel_ms()
it = dev.iters.cpu().numpy()
lib = load_library()
buf = (C.c_ulonglong * (1024 * 16 * 12))()
lib.acnqp_debug_read_stamps_stream(buf, 1024 * 16 * 12)
nb = min(batch.B, 1024)
nw = 4 if batch.B >= 384 else 8
s_ = np.array(buf, dtype=np.float64).reshape(1024, 16, 12)[:nb, :nw]
per = s_ / it[:nb, None, None]
names = ["eigen step + site rows", "front: loads, MFMA, x", "back: bounds, fill, z1/y1", "residual terms", "round: slab, barrier, owners", "closing barrier", "Anderson event iteration", "check / certificate / output"]
tot = per.sum(-1).mean()
print(note, "batch", batch.B, "kernel_ms %.1f" % ms, "iters mean %.0f" % it.mean())
for k, n in enumerate(names):
    print("   %-30s %9.0f ticks/iter  w0 %.0f w%d %.0f   %.1f%%" % (n, per[:, :, k].mean(), per[:, 0, k].mean(), nw - 1, per[:, nw - 1, k].mean(), 100 * per[:, :, k].mean() / tot))
print("   total %.0f ticks per iteration" % tot)
