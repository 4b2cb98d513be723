"""Micro-benchmark of the single-call brute-force path (one query per call, cpu_baseline.cpp:222-254's loop shape) on
SIFT-1M- and SIFT-small-shaped data: HIP-event time per call back to back and host-clock latency of one synchronised call.
Used under rocprofv3 --kernel-trace for the scan_one_kernel rows of profiles/.   python scripts/b1_bench.py [B]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
q = pkg.synth_sift(4096, seed=20251206)
qd = torch.from_numpy(q).cuda(); st = torch.cuda.current_stream().cuda_stream
o_d = torch.zeros((32, 6), dtype=torch.float32, device='cuda'); o_i = torch.zeros((32, 6), dtype=torch.int32, device='cuda')
fl = torch.zeros((32,), dtype=torch.int32, device='cuda')
for n in (1_000_000, 10_000):
    base = pkg.synth_sift(n, seed=20251205)
    with pkg.BruteForceIndex(base) as idx:
        idx.set_precision(1)
        call = lambda i: idx.search_dev(qd.data_ptr() + (i % 64) * B * 128 * 4, B, 5, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(), st)
        for i in range(8): call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(64): call(i)
        e1.record(); torch.cuda.synchronize()
        lat = []
        for i in range(40):
            torch.cuda.synchronize(); t = time.perf_counter(); call(i); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
        print(f"rows {n} B {B}: {e0.elapsed_time(e1) * 1e3 / 64:.2f} us per call back to back, {sorted(lat)[len(lat) // 2] * 1e6:.1f} us synchronised call")
