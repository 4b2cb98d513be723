"""Micro-benchmark of the wide IVF pipeline on SIFT-1M-shaped data (nlist 1024, nprobe 32, launch groups of 256 batches):
device time per call of 256 batches on the exact-int8 rows (precision 0) or on the fp32 rows (precision 1,
IVFIndex.cpp:270-358's arithmetic).  Used under rocprofv3 (--kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE) for the IVF
rows of profiles/.   python scripts/ivf_bench.py [precision] [nprobe]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("VSEARCH_LIB"):  # (a variant build of the library, for A/B runs)
    pkg.LIB_PATH = os.path.abspath(os.environ["VSEARCH_LIB"])
precision = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nprobe = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K, B, NB = 5, 32, 256
base = pkg.synth_sift(1_000_000, seed=20251205)
q = np.tile(pkg.synth_sift(4096, seed=20251206), (2, 1))
vr, off, r2o, cents, it = pkg.ivf_build(base, 1024, max_iter=20, seed=42)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q).to(dev); st = torch.cuda.current_stream().cuda_stream
o_i = torch.zeros((NB * B, K), dtype=torch.int32, device=dev); o_d = torch.zeros((NB * B, K), dtype=torch.float32, device=dev)
with pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
    ivf.set_precision(precision)
    call = lambda: ivf.search_dev_multi(qd.data_ptr(), NB, B, K, nprobe, o_i.data_ptr(), o_d.data_ptr(), st)
    call(); call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 6
    print(f"precision {precision} nprobe {nprobe}: {us:.1f} us per call of {NB} batches = {us / (NB * B / 1024):.1f} us per 1024 queries = {NB * B / us:.2f} M QPS")
