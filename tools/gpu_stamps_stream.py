import sys, os, ctypes as C
sys.path.insert(0, '.')
os.environ["ACNQP_LIBRARY"] = os.path.abspath("adacharge_amd/lib/libacn_qp_hip_stamps.so")
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, load_library
from adacharge_amd.builder import build_batch
infra = sites.synth512(); iface = Interface({"infrastructure_info": infra, "period": 5})
T = 48; ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0), ObjectiveComponent(equal_share, 1e-3)]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
snaps = sites.snapshot_batch(infra, T, NB, seed=512, min_sessions=200)
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
res = h.solve(batch, default_options(eps_abs=1e-6, eps_rel=1e-6))
lib = load_library()
buf = (C.c_ulonglong * (1024*16*12))()
lib.acnqp_debug_read_stamps(buf, 1024*16*12)
st = np.array(buf, dtype=np.float64).reshape(1024,16,12)[:NB, :8]
per_iter = st / res.iters[:,None,None]
names = ["reduce rounds","eigen A+barrier","site rows+barrier","load+x~ mfma+relax","project rows+stores","P accumulate","check terms"]
tot = per_iter.sum(-1).mean()
print("kernel_ms %.2f iters %s" % (res.kernel_ms, np.unique(res.iters)))
for k,n in enumerate(names):
    print("   %-24s %9.0f ticks/iter (wave mean)  w0 %.0f w7 %.0f  %.1f%%"%(n, per_iter[:,:,k].mean(), per_iter[:,0,k].mean(), per_iter[:,7,k].mean(), 100*per_iter[:,:,k].mean()/tot))
print("   total %.0f ticks/iter" % tot)
