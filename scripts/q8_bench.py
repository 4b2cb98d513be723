"""Micro-benchmark of the UFIXED_POINT_8 runner (vs_q8_*) on SIFT-1M-shaped data: HIP-event time per batch of the score
matrix alone and of the search (score matrix + top-k).  Used under rocprofv3 for the q8 rows of profiles/."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
N = 1_000_000
base = pkg.synth_sift(N, seed=20251205); q = pkg.synth_sift(4096, seed=20251206)
in_s = float(q.max())/255; w_s = float(base.max())/255
o_s = 1.05*float((q[:64].astype(np.float64) @ base[::244].astype(np.float64).T).max())/255
r = pkg.Q8Runner(base, in_s, w_s, 0, o_s)
qd = torch.from_numpy(q).cuda(); st = torch.cuda.current_stream().cuda_stream
npad = (N+63)//64*64
sc = torch.empty((32*npad,), dtype=torch.uint8, device='cuda')
ids = torch.empty((32*32,5), dtype=torch.int32, device='cuda'); top = torch.empty((32*32,5), dtype=torch.uint8, device='cuda')
def ev(fn, n):
    for i in range(4): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
print("execute us/batch", ev(lambda i: r.execute_dev(qd.data_ptr()+(i%128)*32*128*4, 32, sc.data_ptr(), npad, st), 64))
print("search us/batch", ev(lambda i: r.search_dev(qd.data_ptr(), 32, 32, 5, ids.data_ptr(), top.data_ptr(), st), 8)/32)
