"""What a rank of a cluster-sharded IVF job does per launch group, measured on ONE GPU (SURVEY 8e / BASELINE configs[4]):
SIFT-1M-shaped index, nlist 1024, nprobe 32.  (1) the unsharded index: device time per 1024 queries for launch groups of
32 / 64 / 128 / 256 batches on one and on two streams; (2) virtual ranks (vs_ivf_search_dev_vshards): world = 2, 4, 8
shards of the same index on this GPU run the sliced pipeline one after the other; per-rank device time (front + back half)
per launch group of 32 * world batches, and per 1024 queries.   python scripts/ivf_shard_bench.py [nprobe]"""
import sys, os, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, B, NLIST = 5, 32, 1024
base = pkg.synth_sift(1_000_000, seed=20251205)
q = pkg.synth_sift(8192 + 4096, seed=20251206)
vr, off, r2o, cents, it = pkg.ivf_build(base, NLIST, max_iter=20, seed=42)
dev = torch.device("cuda", 0)
qd = torch.from_numpy(q).to(dev)
st = torch.cuda.current_stream().cuda_stream
NB = 256
o_i = torch.zeros((NB * B, K), dtype=torch.int32, device=dev)
o_d = torch.zeros((NB * B, K), dtype=torch.float32, device=dev)
out = {"nprobe": nprobe, "nlist": NLIST, "rows": len(base), "batch": B, "k": K}

def ev_time(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]

def make(group=None, lanes=None, **kw):
    env = {"VSEARCH_IVF_GROUP": group, "VSEARCH_IVF_WIDE_LANES": lanes}
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is not None:
            os.environ[k] = str(v)
    try:
        return pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

unsh = {}
for lanes in (1, 2):
    for group in (32, 64, 128, 256):
        with make(group, lanes) as ivf:
            us = ev_time(lambda: ivf.search_dev_multi(qd.data_ptr(), NB, B, K, nprobe, o_i.data_ptr(), o_d.data_ptr(), st), 4)
            unsh[f"group{group}_lanes{lanes}"] = round(us / (NB * B / 1024), 2)
            print(f"unsharded, groups of {group} batches, {lanes} stream(s): {us / (NB * B / 1024):.1f} us per 1024 queries ({NB * B / us:.2f} M QPS)", flush=True)
out["unsharded_us_per_1024_queries"] = unsh
ref = unsh["group32_lanes1"]
want_i, want_d = None, None
with make(32, 1) as ivf:
    ivf.search_dev_multi(qd.data_ptr(), NB, B, K, nprobe, o_i.data_ptr(), o_d.data_ptr(), st)
    torch.cuda.synchronize()
    want_d = o_d.clone()
sh = {}
for world in (2, 4, 8):
    shards = [make(rank=r, world=world) for r in range(world)]
    nb = 32 * world
    g_i = torch.zeros((nb * B, K), dtype=torch.int32, device=dev)
    g_d = torch.zeros((nb * B, K), dtype=torch.float32, device=dev)
    call = lambda timed=False: pkg.IVFIndex.search_dev_vshards(shards, qd.data_ptr(), nb, B, K, nprobe, g_i.data_ptr(), g_d.data_ptr(), st, timed=timed)
    call(); call()
    torch.cuda.synchronize()
    assert torch.equal(g_d, want_d[:nb * B]), "sliced pipeline differs from the unsharded index"
    ms = np.array([call(True) for _ in range(5)])
    per_rank = np.median(ms, axis=0) * 1e3  # us per launch group of nb batches, per rank
    worst = float(per_rank.max())
    sh[f"world{world}"] = {"batches_per_group": nb, "rank_us_per_group": [round(float(x), 1) for x in per_rank],
                           "slowest_rank_us_per_1024_queries": round(worst / world, 2),
                           "vs_unsharded_one_stream": round(ref / (worst / world), 2)}
    print(f"world {world}: per-rank device time per group of {nb} batches: {np.round(per_rank, 1).tolist()} us -> "
          f"{worst / world:.1f} us per 1024 queries on the slowest rank = {ref / (worst / world):.2f}x the unsharded index "
          f"({ref:.1f} us, one stream, groups of 32)", flush=True)
    for s_ in shards:
        s_.close()
out["sliced_virtual_ranks"] = sh
print(json.dumps(out))
