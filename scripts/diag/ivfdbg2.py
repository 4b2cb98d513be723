import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.LIB_PATH = os.path.abspath('scratch/stamps/libvsearch_hip.so')
os.environ["VSEARCH_DIAG"] = "128"
L = pkg.lib()
L.vs_debug_ivf_wide_stats.restype = C.c_int
L.vs_debug_ivf_wide_stats.argtypes = [C.c_void_p, C.c_void_p]
NB = 256
base = pkg.synth_sift(1_000_000, seed=20251205)
q = np.tile(pkg.synth_sift(4096, seed=20251206), (2, 1))
vr, off, r2o, cents, it = pkg.ivf_build(base, 1024, max_iter=20, seed=42)
dev = torch.device('cuda', 0)
qd = torch.from_numpy(q).to(dev)
s = torch.cuda.current_stream().cuda_stream
od = torch.zeros((NB * 32, 5), dtype=torch.float32, device=dev)
oi = torch.zeros((NB * 32, 5), dtype=torch.int32, device=dev)
with pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
    for nprobe in (32, 8):
        ivf.search_dev_multi(qd.data_ptr(), NB, 32, 5, nprobe, oi.data_ptr(), od.data_ptr(), s)
        torch.cuda.synchronize()
        out = (C.c_int64 * 8)()
        print(L.vs_debug_ivf_wide_stats(ivf._h, out), "nprobe", nprobe, "overflow", out[0], "slow", out[1], "candidates", out[2], "per query %.1f" % (out[2] / (NB * 32)),
              "max sublist", out[4], "tau inf", out[5], "units sb0", out[6], "queries>256 / max per query", out[7] // 100000, out[7] % 100000)
