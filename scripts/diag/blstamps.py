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
    t = dbg.cpu().numpy()[16384:16384 + 1024].astype(np.int64)
    live = t[:, 3] != 0
    t0 = t[:, 0].min()
    st = (t[:, :4] - t0) / 100.0
    print("bounds WGs (part 0) with work:", live.sum(), "n max", t[:, 8].max())
    print("start: mean %.1f max %.1f | counter read: mean %.1f max %.1f" % (st[:, 0].mean(), st[:, 0].max(), (st[:, 1] - st[:, 0]).mean(), (st[:, 1] - st[:, 0]).max()))
    print("scored (first... last unit of wave 0): mean %.1f max %.1f  (since counter: mean %.1f max %.1f)" % (st[live, 2].mean(), st[live, 2].max(), (st[live, 2] - st[live, 1]).mean(), (st[live, 2] - st[live, 1]).max()))
    print("selected: mean %.1f max %.1f (since scored: mean %.1f max %.1f)" % (st[live, 3].mean(), st[live, 3].max(), (st[live, 3] - st[live, 2]).mean(), (st[live, 3] - st[live, 2]).max()))
    tp = dbg.cpu().numpy()[20480:20480 + 128].astype(np.int64)
    tp = tp[tp[:, 0] != 0]
    stp = (tp[:, :4] - t0) / 100.0
    print('plan starts', np.round(np.sort(stp[:,0])[:8],1), 'ends', np.round(np.sort(stp[:,3])[-8:],1))
    print("plan: class pass (2->4) %.1f, records (4->3) %.1f" % ((tp[:,4]-tp[:,2]).mean()/100, (tp[:,3]-tp[:,4]).mean()/100))
    print("plan phases per WG (us): loaded %.1f totals %.1f classes+records %.1f | life mean %.1f max %.1f" % ((tp[:,1]-tp[:,0]).mean()/100, (tp[:,2]-tp[:,1]).mean()/100, (tp[:,3]-tp[:,2]).mean()/100, (tp[:,3]-tp[:,0]).mean()/100, (tp[:,3]-tp[:,0]).max()/100))
    print("plan WGs", len(tp), "start mean %.1f loaded %.1f totals %.1f end mean %.1f max %.1f" % (stp[:, 0].mean(), stp[:, 1].mean(), stp[:, 2].mean(), stp[:, 3].mean(), stp[:, 3].max()))
    order = np.argsort(st[:, 0])
    print("start quantiles:", np.round(np.quantile(st[:, 0], [0, .1, .25, .5, .75, .9, 1]), 1))
    print("life (start->selected) quantiles:", np.round(np.quantile(st[live, 3] - st[live, 0], [0, .25, .5, .75, 1]), 1))
    xcc = t[:, 9]; hw = t[:, 10]
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    u, cnt = np.unique(key, return_counts=True)
    print("distinct (xcc,se,sh,cu):", len(u), "WGs per CU min/max", cnt.min(), cnt.max())
    # concurrency: max overlapping lifetimes per CU
    mx = 0
    for k in u[:40]:
        sel = key == k
        ev = sorted([(a, 1) for a in st[sel, 0]] + [(b, -1) for b in st[sel, 3]])
        c = 0
        for _, d in ev:
            c += d; mx = max(mx, c)
    print("max concurrent part-0 WGs on a CU (first 40 CUs):", mx)
    for i in order[:5].tolist() + order[-5:].tolist():
        print(i, np.round(st[i], 1), "n", t[i, 8], "xcc", xcc[i], "se", se[i], "cu", cu[i])
