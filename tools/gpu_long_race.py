"""Where do two copies of the same problem first part ways inside ONE launch of the long-horizon kernel?

Diagnostic for the run-to-run nondeterminism of the workspace placement with scalar-resident block ids (DESIGN.md
section 3.6).  Needs a diagnostic build: build_hip_library(extra_flags=["-DACNQP_DEBUG_WS", "-DACNQP_LONG_IDS_OPAQUE=1"],
out=".../libacn_qp_hip_race.so").  The batch holds every problem twice (i and i + nb); after K iterations (max_iter = K,
one residual check at the end, no retry) the kernel's workspace is copied back and the two copies are compared array by
array.  Prints, per K, the arrays that differ and where (EVSE tile, column tile, register pair, lane).

    ACNQP_NO_RZL=1 python tools/gpu_long_race.py [T] [nb] [K ...]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ACNQP_LIBRARY", os.path.join(ROOT, "adacharge_amd", "lib", "libacn_qp_hip_race.so"))
import numpy as np
import torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options, load_library
from adacharge_amd.builder import build_batch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 144
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Ks = [int(a) for a in sys.argv[3:]] or [1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 40]
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, T, nb, seed=100 + T, demand_range=(5.0, 60.0))
batch = build_batch(snaps + snaps, infra, iface, obj, "SOC")
B = batch.B
lib = load_library()
lib.acnqp_debug_copy_workspace.restype = C.c_int64
lib.acnqp_debug_copy_workspace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
NP = 64
CTL = {96: 6, 144: 9}.get(T, (T + 15) // 16)
NE = NP // 16
NT = NE * CTL * 256
MT = None


def decode(region, idx):
    if region in ("X", "Z1", "Y1", "Q", "LB", "UB", "RZ"):
        tile, rem = divmod(idx, 256)
        e, c = divmod(tile, CTL)
        half, rem = divmod(rem, 128)
        lane, odd = divmod(rem, 2)
        return f"e{e} c{c} r{2 * half + odd} lane{lane}"
    return str(idx)


for K in Ks:
    h = SiteHandle(batch.site, 0)
    dev = DeviceBatch(batch, "cuda:0")
    o = default_options()
    o.max_iter = K; o.check_every = 1 << 20; o.retry_passes = 0; o.stall_iters = 0
    h.solve_device(dev, o, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    have = int(lib.acnqp_debug_copy_workspace(h._h, None, 0))
    per = have // B
    W = np.empty(have, np.float64)
    lib.acnqp_debug_copy_workspace(h._h, W.ctypes.data_as(C.c_void_p), have)
    W = W[: per * B].reshape(B, per)
    Kses = batch.K
    regions = [("X", 0, NT), ("Z1", NT, 2 * NT), ("Y1", 2 * NT, 3 * NT), ("Q", 3 * NT, 4 * NT), ("LB", 4 * NT, 5 * NT),
               ("UB", 5 * NT, 6 * NT), ("RZ", 6 * NT, 7 * NT), ("MU", 7 * NT, 7 * NT + Kses * NP), ("rest", 7 * NT + Kses * NP, per)]
    x = dev.x.cpu().numpy()
    it = dev.iters.cpu().numpy()
    nd_x = int((x[:nb] != x[nb:]).any(axis=(1, 2)).sum())
    line = [f"K={K:3d} per-problem ws {per} doubles; problems whose x differs between the copies: {nd_x}/{nb}"]
    for name, a, b_ in regions:
        A_, B_ = W[:nb, a:b_], W[nb:, a:b_]
        ne = (A_.view(np.uint64) != B_.view(np.uint64))
        if ne.any():
            probs = np.nonzero(ne.any(axis=1))[0]
            p0 = int(probs[0])
            idx = np.nonzero(ne[p0])[0]
            line.append(f"   {name:4s}: {int(ne.sum())} words differ in {len(probs)} problems; problem {p0}: {len(idx)} words, first "
                        + ", ".join(decode(name, int(i)) + f" ({A_[p0, i]:.6g} vs {B_[p0, i]:.6g})" for i in idx[:6]))
    print("\n".join(line), flush=True)
    h.close()
