import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.LIB_PATH = os.path.abspath('scratch/stamps/libvsearch_hip.so')
L = pkg.lib()
L.vs_debug_buffer.argtypes = [C.c_void_p]
q = pkg.synth_sift(4096, seed=20251206)
qd = torch.from_numpy(q).cuda(); st = torch.cuda.current_stream().cuda_stream
o_d = torch.zeros((32, 6), dtype=torch.float32, device='cuda'); o_i = torch.zeros((32, 6), dtype=torch.int32, device='cuda')
fl = torch.zeros((32,), dtype=torch.int32, device='cuda')
dbg = torch.zeros((256, 16), dtype=torch.int32, device='cuda')
L.vs_debug_buffer(dbg.data_ptr())
for n in (1_000_000, 10_000):
    base = pkg.synth_sift(n, seed=20251205)
    with pkg.BruteForceIndex(base) as idx:
        idx.set_precision(1)
        for B in (1, 32):
            for rep in range(4):
                dbg.zero_()
                idx.search_dev(qd.data_ptr() + rep * B * 512, B, 5, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(), st)
                torch.cuda.synchronize()
            t = dbg.cpu().numpy().astype(np.int64)
            live = t[:, 0] != 0
            t = t[live]
            t0 = t[:, 0].min()
            s = (t[:, :7] - t0) / 100.0
            last = t[:, 7] != 0
            print(f"rows {n} B {B}: WGs {live.sum()}  stamps mean (us): entry {s[:,0].mean():.1f} prologue {s[:,1].mean():.1f} loop {s[:,2].mean():.1f} (min {s[:,2].min():.1f} max {s[:,2].max():.1f}) "
                  f"compact {s[:,3].mean():.1f} ranked {s[:,4].mean():.1f} barrier {s[:,5].mean():.1f} ticket {s[:,6].mean():.1f} max {s[:,6].max():.1f}; last WG: ticket {((t[last,6]-t0)/100.0)} fence {((t[last,8]-t0)/100.0)} firsts-issued {((t[last,9]-t0)/100.0)} M0 {t[last,11]} end {((t[last,7]-t0)/100.0)}")
