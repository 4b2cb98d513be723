import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.LIB_PATH = os.path.abspath('scratch/stamps/libvsearch_hip.so')
L = pkg.lib()
L.vs_debug_buffer.argtypes = [C.c_void_p]
NB = 256
base = pkg.synth_sift(1_000_000, seed=20251205)
q = np.tile(pkg.synth_sift(4096, seed=20251206), (2, 1))
vr, off, r2o, cents, it = pkg.ivf_build(base, 1024, max_iter=20, seed=42)
dev = torch.device('cuda', 0)
qd = torch.from_numpy(q).to(dev)
s = torch.cuda.current_stream().cuda_stream
od = torch.zeros((NB * 32, 5), dtype=torch.float32, device=dev)
oi = torch.zeros((NB * 32, 5), dtype=torch.int32, device=dev)
dbg = torch.zeros((32768, 16), dtype=torch.int32, device=dev)
L.vs_debug_buffer(dbg.data_ptr())
with pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
    for rep in range(3):
        dbg.zero_()
        ivf.search_dev_multi(qd.data_ptr(), NB, 32, 5, 32, oi.data_ptr(), od.data_ptr(), s)
        torch.cuda.synchronize()
    t = dbg.cpu().numpy()[4096 + 20480:4096 + 20480 + 512].astype(np.int64)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    st = (t[:, :9] - t0) / 100.0
    for i, n in ((0, "start"), (1, "queries+norms"), (2, "tile 0 done"), (3, "t1 after wait"), (5, "t1 mfma issued"), (6, "t1 epilogue"), (7, "end")):
        print("%-14s mean %.2f min %.2f max %.2f" % (n, st[:, i].mean(), st[:, i].min(), st[:, i].max()))
    print("per tile (end - tile0 done)/7: %.2f us" % ((st[:, 7] - st[:, 2]).mean() / 7))
