"""IVF path on the GPU against the oracle's restatement of IVFIndex::searchBatch (L2).
The reference's IVF code cannot be built and holds no golden vectors ("parity unpinned"): the
pins are (i) nprobe == nlist must equal exact search, (ii) agreement with the oracle."""
import json
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


_INDEX_CACHE = {}


def _make_index(pkg, n=20000, nlist=64, seed=3, jitter=0.0):
    key = (n, nlist, seed, jitter)
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = _build_index(pkg, n, nlist, seed, jitter)
    return _INDEX_CACHE[key]


def _build_index(pkg, n, nlist, seed, jitter):
    base = pkg.synth_sift(n, seed=seed)
    rng = np.random.default_rng(seed)
    cents = base[rng.choice(n, nlist, replace=False)].copy()
    for _ in range(2):  # a few Lloyd steps on the host (index building is not the path under test)
        d = (base ** 2).sum(1)[:, None] - 2 * base @ cents.T + (cents ** 2).sum(1)[None]
        a = d.argmin(1)
        for c in range(nlist):
            if (a == c).any():
                cents[c] = base[a == c].mean(0)
    cents = (cents + jitter).astype(np.float32)
    d = (base ** 2).sum(1)[:, None] - 2 * base @ cents.T + (cents ** 2).sum(1)[None]
    a = d.argmin(1)
    vr, off, r2o = pkg.ivf_layout_from_assignment(base, a, nlist)
    return base, cents, vr, off, r2o


def test_full_probe_equals_exact(gpu_pkg):
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=12000, nlist=32)
    q = gpu_pkg.synth_sift(50, seed=99)
    oi, od = oracle.search_bf(base, q, 5)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        assert ivf.getNumVectors() == 12000 and ivf.getNumClusters() == 32 and ivf.getDim() == 128
        ids, d, total = ivf.searchBatch(q, len(q), 5, 32)
        ids2, d2, total2 = ivf.searchBatch(q, len(q), 5, 1000)  # nprobe clamps to nlist (IVFIndex.cpp:647)
    assert total == total2 == len(q) * len(base)
    assert np.array_equal(d, od) and np.array_equal(d2, od)
    ex = oracle.exact_int_dists(q, base)
    assert np.array_equal(np.take_along_axis(ex, ids.astype(np.int64), 1).astype(np.float32), od)
    # where the oracle has no distance ties (inside the top-k or at its boundary) the ids are identical too
    top6 = np.sort(ex, axis=1)[:, :6]
    notie = (top6[:, 1:] != top6[:, :-1]).all(1)
    assert notie.mean() > 0.9 and np.array_equal(ids[notie], oi[notie])


@pytest.mark.parametrize("nprobe", [100, 256])
def test_large_nprobe_on_a_launch_group_of_several_super_batches(gpu_pkg, nprobe):
    """nprobe above 64: the bound tables take a query's segments from its nearest 64 probes only, the slot tables take all
    of them, a list is probed by hundreds of a super-batch's queries.  nprobe = nlist = 256 is the exact search: distances
    must equal the brute-force oracle's; nprobe = 100 must equal the IVF oracle's."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=60000, nlist=256, seed=16)
    nb, B, k = 40, 32, 5
    q = gpu_pkg.synth_sift(nb * B, seed=321)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        got_i = torch.full((nb * B, k), -7, dtype=torch.int32, device=dev)
        got_d = torch.zeros((nb * B, k), dtype=torch.float32, device=dev)
        ivf.search_dev_multi(qd.data_ptr(), nb, B, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
        torch.cuda.synchronize()
    gi, gd = got_i.cpu().numpy(), got_d.cpu().numpy()
    sub = np.r_[0:48, 32 * 20:32 * 20 + 48, 32 * 39:32 * 40]
    if nprobe == 256:
        _, od = oracle.search_bf(base, q[sub], k)
        assert np.array_equal(gd[sub], od)
    else:
        _, od, _ = oracle.ivf_search(vr, off, r2o, cents, q[sub], k, nprobe)
        same = np.array([np.array_equal(gd[sub][i], od[i]) for i in range(len(sub))])
        assert same.mean() >= 0.97  # (probe sets can differ by a last-bit coarse tie, see test_matches_oracle_ivf)
    ex = oracle.exact_int_dists(q[sub], base)
    assert np.array_equal(np.take_along_axis(ex, gi[sub].astype(np.int64), 1).astype(np.float32), gd[sub])


@pytest.mark.parametrize("nprobe", [1, 8, 32])
@pytest.mark.parametrize("k", [1, 5, 10])
def test_matches_oracle_ivf(gpu_pkg, nprobe, k):
    base, cents, vr, off, r2o = _make_index(gpu_pkg)
    q = gpu_pkg.synth_sift(70, seed=77)
    oi, od, ototal, oprobes = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe, return_probes=True)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, d, total = ivf.searchBatch(q, len(q), k, nprobe)
    # Centroids are not integer valued, so a coarse distance can differ in the last bit and swap
    # two nearly tied probes: require identical results on >= 97 % of the queries and identical
    # candidate counts wherever the probe sets agree.
    same = np.array([np.array_equal(d[i], od[i]) for i in range(len(q))])
    assert same.mean() >= 0.97
    assert abs(total - ototal) <= 0.02 * ototal
    ex = oracle.exact_int_dists(q, base)
    valid = ids >= 0
    assert np.array_equal(np.take_along_axis(ex, np.where(valid, ids, 0).astype(np.int64), 1).astype(np.float32)[valid], d[valid])
    gt, _ = oracle.search_bf(base, q, k)
    assert abs(oracle.recall(ids, gt, k) - oracle.recall(oi, gt, k)) < 0.02


def test_recall_at_1_on_clustered_data(gpu_pkg):
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=40000, nlist=128, seed=5)
    q = gpu_pkg.synth_sift(200, seed=55)
    gt, _ = oracle.search_bf(base, q, 5)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        r = {}
        for nprobe in (1, 4, 16, 128):
            ids, _, _ = ivf.searchBatch(q, len(q), 5, nprobe)
            r[nprobe] = (oracle.recall(ids[:, :1], gt[:, :1], 1), oracle.recall(ids, gt, 5))
    assert r[128] == (1.0, 1.0)
    assert r[1][0] <= r[4][0] <= r[16][0] <= 1.0
    assert r[16][0] >= 0.91  # the north-star's bar, at nprobe/nlist = 1/8


def test_empty_lists_ragged_batches_and_small_k_pool(gpu_pkg):
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=3000, nlist=16, seed=8)
    # add 4 empty lists at the end (centroids far away) -- offsets repeat
    cents2 = np.concatenate([cents, np.full((4, 128), 1e4, dtype=np.float32)])
    off2 = np.concatenate([off, np.full(4, off[-1], dtype=np.int32)])
    q = gpu_pkg.synth_sift(37, seed=9)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents2, cluster_offsets=off2, reorder_to_original=r2o) as ivf:
        for batch in (1, 5, 32):
            ivf.set_batch(batch)
            ids, d, total = ivf.searchBatch(q, len(q), 5, 20)
            assert total == len(q) * len(base)
            assert np.array_equal(d, oracle.search_bf(base, q, 5)[1])
        one_i, one_d, _ = ivf.search(q[0], 5, 20)
        assert np.array_equal(one_d, d[0])
    # fewer candidates than k: the reference clamps k (IVFIndex.cpp:735); unused slots are -1 / inf
    tiny = base[:3]
    with gpu_pkg.IVFIndex(vectors_reordered=tiny, centroids=tiny[:1], cluster_offsets=[0, 3]) as ivf:
        ids, d, total = ivf.searchBatch(q[:2], 2, 5, 1)
    assert total == 6 and (ids[:, 3:] == -1).all() and np.isinf(d[:, 3:]).all() and (ids[:, :3] >= 0).all()


def test_index_directory_roundtrip(gpu_pkg, tmp_path):
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=5000, nlist=16, seed=4)
    q = gpu_pkg.synth_sift(20, seed=44)
    d1 = str(tmp_path / "saved")
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        want = ivf.searchBatch(q, len(q), 5, 4)
        ivf.save(d1)
    cfg = json.load(open(os.path.join(d1, "ivf_config.json")))
    assert cfg["n_vectors"] == 5000 and cfg["n_clusters"] == 16 and cfg["dim"] == 128 and cfg["reordered"] is True
    assert np.array_equal(np.load(os.path.join(d1, "vectors_reordered.npy")), vr)
    assert np.array_equal(np.load(os.path.join(d1, "cluster_offsets.npy")), off)
    assert np.array_equal(np.load(os.path.join(d1, "reorder_to_original.npy")), r2o)
    with gpu_pkg.IVFIndex(d1) as ivf2:
        got = ivf2.searchBatch(q, len(q), 5, 4)
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]) and want[2] == got[2]
    # a directory written the way the reference's builder writes it (numpy np.save + json.dump), plain mode
    d2 = tmp_path / "plain"
    d2.mkdir()
    np.save(d2 / "vectors.npy", base)
    np.save(d2 / "cluster_indices.npy", r2o)
    np.save(d2 / "cluster_offsets.npy", off)
    np.save(d2 / "centroids.npy", cents)
    json.dump({"n_vectors": 5000, "n_clusters": 16, "dim": 128, "avg_cluster_size": 312.5}, open(d2 / "ivf_config.json", "w"))
    with gpu_pkg.IVFIndex(str(d2)) as ivf3:
        got3 = ivf3.searchBatch(q, len(q), 5, 4)
    assert np.array_equal(want[0], got3[0]) and np.array_equal(want[1], got3[1])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cluster_shards_merge_to_unsharded(gpu_pkg, world):
    """Virtual shards on one GPU: every rank runs the same coarse stage, scans only its lists,
    and the merge of the per-shard top-k equals the unsharded result."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=16000, nlist=64, seed=6)
    q = gpu_pkg.synth_sift(32, seed=66)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)
    k, nprobe = 5, 16
    def run(rank, w):
        with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o,
                              rank=rank, world=w) as ivf:
            ids = torch.zeros((32, k), dtype=torch.int32, device=dev)
            d = torch.zeros((32, k), dtype=torch.float32, device=dev)
            ivf.search_dev(qd.data_ptr(), 32, k, nprobe, ids.data_ptr(), d.data_ptr(), s)
            torch.cuda.synchronize()
            return d, ids
    d_all, i_all = run(0, 1)
    parts = [run(r, world) for r in range(world)]
    gd = torch.stack([p[0] for p in parts]).contiguous()
    gi = torch.stack([p[1] for p in parts]).contiguous()
    od = torch.zeros((32, k), dtype=torch.float32, device=dev)
    oi = torch.zeros((32, k), dtype=torch.int32, device=dev)
    gpu_pkg.topk_merge_dev(gd.data_ptr(), gi.data_ptr(), world, 32, k, k, od.data_ptr(), oi.data_ptr(), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(od, d_all)
    # ids: the unsharded IVF orders equal distances by reordered position, the merged one by
    # original id; compare as sets per distance value
    a, b = i_all.cpu().numpy(), oi.cpu().numpy()
    for i in range(32):
        assert sorted(a[i].tolist()) == sorted(b[i].tolist())


def test_gpu_index_builder(gpu_pkg):
    """vs_ivf_build: Lloyd k-means on the MFMA scan kernel + deterministic update, then the
    reordered layout of create_ivf_model_reordered.py:108-128."""
    base = gpu_pkg.synth_sift(30000, seed=21)
    vr, off, r2o, cents, n_iter = gpu_pkg.ivf_build(base, 64, max_iter=25, seed=42)
    assert cents.shape == (64, 128) and off.shape == (65,) and off[0] == 0 and off[-1] == 30000
    assert 1 <= n_iter <= 25
    assert np.array_equal(np.sort(r2o), np.arange(30000))          # a permutation
    assert np.array_equal(vr, base[r2o])                           # vectors[sorted_indices] (:112)
    # every row sits in the list of its nearest centroid (final E-step), up to fp32 rounding of the distance
    d = (base.astype(np.float64) ** 2).sum(1)[:, None] - 2 * base.astype(np.float64) @ cents.astype(np.float64).T \
        + (cents.astype(np.float64) ** 2).sum(1)[None]
    assign = np.empty(30000, dtype=np.int64)
    for c in range(64):
        assign[r2o[off[c]:off[c + 1]]] = c
    best = d.min(1)
    chosen = d[np.arange(30000), assign]
    assert np.all(chosen - best <= 1e-3 * np.maximum(best, 1.0))
    assert (assign == d.argmin(1)).mean() > 0.999
    # stable argsort: original ids ascend inside every list (:111)
    for c in (0, 17, 63):
        seg = r2o[off[c]:off[c + 1]]
        assert np.all(np.diff(seg) > 0)
    # deterministic: same seed -> bit-identical index; other seed -> different
    vr2, off2, r2o2, cents2, _ = gpu_pkg.ivf_build(base, 64, max_iter=25, seed=42)
    assert np.array_equal(cents, cents2) and np.array_equal(r2o, r2o2) and np.array_equal(off, off2)
    _, _, _, cents3, _ = gpu_pkg.ivf_build(base, 64, max_iter=25, seed=7)
    assert not np.array_equal(cents, cents3)
    # k-means quality: inertia far below that of random centres, lists reasonably balanced
    inertia = chosen.sum()
    rnd = base[np.random.default_rng(0).choice(30000, 64, replace=False)].astype(np.float64)
    d0 = (base.astype(np.float64) ** 2).sum(1)[:, None] - 2 * base.astype(np.float64) @ rnd.T + (rnd ** 2).sum(1)[None]
    assert inertia < 0.9 * d0.min(1).sum()
    # the nlist clamp of create_ivf_model_reordered.py:92-94
    _, off_c, _, cents_c, _ = gpu_pkg.ivf_build(base[:2000], 1024, max_iter=3)
    assert cents_c.shape[0] == gpu_pkg.clamp_nlist(2000, 1024) == 20
    # and the index it builds searches correctly
    q = gpu_pkg.synth_sift(64, seed=22)
    gt, _ = oracle.search_bf(base, q, 5)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, _, _ = ivf.searchBatch(q, len(q), 5, 64)
        assert np.array_equal(np.sort(ids, 1), np.sort(gt, 1)) or oracle.recall(ids, gt, 5) > 0.995
        ids8, _, _ = ivf.searchBatch(q, len(q), 5, 8)
        ids32, _, _ = ivf.searchBatch(q, len(q), 5, 32)
    r8, r32 = oracle.recall(ids8[:, :1], gt[:, :1], 1), oracle.recall(ids32[:, :1], gt[:, :1], 1)
    assert 0.5 < r8 <= r32 and r32 >= 0.91


def test_multi_batch_equals_single_batches(gpu_pkg):
    """vs_ivf_search_dev_multi (two alternating streams, two scratch sets) returns what batch-by-batch calls
    return, for an odd number of batches and on repeated calls."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=20000, nlist=64, seed=3)
    nb, k, nprobe = 7, 5, 8
    q = gpu_pkg.synth_sift(32 * nb, seed=77)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        want_i = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
        want_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
        for b in range(nb):
            ivf.search_dev(qd.data_ptr() + b * 32 * 128 * 4, 32, k, nprobe, want_i.data_ptr() + b * 32 * k * 4,
                           want_d.data_ptr() + b * 32 * k * 4, s)
        torch.cuda.synchronize()
        for _ in range(3):
            got_i = torch.full((nb * 32, k), -7, dtype=torch.int32, device=dev)
            got_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
            torch.cuda.synchronize()
            assert torch.equal(got_d, want_d) and torch.equal(got_i, want_i)
        # and against the host API (which the oracle tests pin)
        ids, dists, _ = ivf.searchBatch(q, len(q), k, nprobe)
        assert np.array_equal(ids, want_i.cpu().numpy()) and np.array_equal(dists, want_d.cpu().numpy())


def test_sweep_driver_end_to_end(gpu_pkg, tmp_path):
    """the nprobe sweep driver builds an index (GPU k-means, reference directory format), runs the CLI
    per nprobe and writes the CSV; recall grows with nprobe and reaches 100 % at nprobe = nlist."""
    import csv
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = gpu_pkg.synth_sift(6000, seed=11)
    q = gpu_pkg.synth_sift(64, seed=12)
    with gpu_pkg.BruteForceIndex(base) as bf:
        gt, _ = bf.search(q, 5)
    gpu_pkg.write_fvecs(str(tmp_path / "b.fvecs"), base)
    gpu_pkg.write_fvecs(str(tmp_path / "q.fvecs"), q)
    gpu_pkg.write_ivecs(str(tmp_path / "gt.ivecs"), gt)
    out = tmp_path / "res"
    subprocess.run([sys.executable, os.path.join(root, "scripts", "sweep_ivf.py"), "--dataset", "synth", "--base",
                    str(tmp_path / "b.fvecs"), "--queries", str(tmp_path / "q.fvecs"), "--groundtruth", str(tmp_path / "gt.ivecs"),
                    "--index-dir", str(tmp_path / "idx"), "--nlist", "64", "--max-iter", "10", "--nprobes", "1", "8", "64",
                    "--top-k", "5", "--batch", "32", "--out", str(out)], check=True, cwd=str(tmp_path))
    files = [f for f in os.listdir(out) if f.endswith(".csv")]
    assert len(files) == 1
    rows = list(csv.DictReader(open(out / files[0])))
    assert [r["nprobe"] for r in rows] == ["1", "8", "64"]
    rec = [float(r["recall"]) for r in rows]
    assert rec[0] <= rec[1] <= rec[2] and rec[2] == 100.0
    assert all(float(r["qps"]) > 0 and float(r["avg_candidates"]) > 0 for r in rows)
    assert os.path.exists(tmp_path / "idx" / "ivf_config.json")


def test_int8_rows_and_fp32_fallback_agree(gpu_pkg):
    """The list scan reads the exact int8 copy of byte-valued rows when the batch's queries are byte valued too, and
    the fp32 rows otherwise (decided per batch inside the kernel).  (i) integer batch: distances equal the exact
    integer distances; (ii) the same queries with one made non-integer (whole batch on the fp32 rows): every other
    query keeps its result bit for bit; (iii) a non-byte-valued base has no int8 copy and still matches."""
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=20000, nlist=64, seed=3)
    q = gpu_pkg.synth_sift(32, seed=91)
    k, nprobe = 5, 16
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, d, _ = ivf.searchBatch(q, 32, k, nprobe)
        ex = oracle.exact_int_dists(q, base)
        assert np.array_equal(np.take_along_axis(ex, ids.astype(np.int64), 1).astype(np.float32), d)
        q2 = q.copy()
        q2[5, 7] += 0.5
        ids2, d2, _ = ivf.searchBatch(q2, 32, k, nprobe)
        keep = np.arange(32) != 5
        assert np.array_equal(ids2[keep], ids[keep]) and np.array_equal(d2[keep], d[keep])
        oi, od, _ = oracle.ivf_search(vr, off, r2o, cents, q2, k, nprobe)
        assert np.array_equal(ids2[5], oi[5]) and np.allclose(d2[5], od[5], rtol=0, atol=1e-2)
    # scaled base: integers up to 436, no int8 copy
    with gpu_pkg.IVFIndex(vectors_reordered=vr * 2.0, centroids=cents * 2.0, cluster_offsets=off, reorder_to_original=r2o) as ivf2:
        ids3, d3, _ = ivf2.searchBatch(q * 2.0, 32, k, nprobe)
        assert np.array_equal(ids3, ids) and np.array_equal(d3, d * 4.0)


@pytest.mark.parametrize("nlist", [1024, 1500, 2500, 5000])
def test_large_nlist_paths(gpu_pkg, nlist):
    """nlist <= 1024, <= 2048 and <= 4096 take the three instantiations of the pick kernel of the wide list-major
    pipeline; nlist > 4096 the query-major fallback (coarse scores on the MFMA scan kernel, probe pick, one workgroup per
    (query, probe)).  All must agree with the oracle's restatement (same probes up to last-bit coarse ties) and return
    exact integer distances; multi-batch == batch by batch."""
    import torch
    base = gpu_pkg.synth_sift(60000, seed=31)
    vr, off, r2o, cents, _ = gpu_pkg.ivf_build(base, nlist, max_iter=3, seed=7)
    assert len(off) == nlist + 1
    q = gpu_pkg.synth_sift(3 * 32, seed=32)
    k, nprobe = 5, 24
    oi, od, _ = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, d, _ = ivf.searchBatch(q, len(q), k, nprobe)
        same = np.array([np.array_equal(d[i], od[i]) for i in range(len(q))])
        assert same.mean() >= 0.95
        ex = oracle.exact_int_dists(q, base)
        valid = ids >= 0
        assert np.array_equal(np.take_along_axis(ex, np.where(valid, ids, 0).astype(np.int64), 1).astype(np.float32)[valid], d[valid])
        dev = torch.device("cuda:0")
        qd = torch.from_numpy(q).to(dev)
        s = torch.cuda.current_stream().cuda_stream
        gi = torch.zeros((96, k), dtype=torch.int32, device=dev)
        gd = torch.zeros((96, k), dtype=torch.float32, device=dev)
        ivf.search_dev_multi(qd.data_ptr(), 3, 32, k, nprobe, gi.data_ptr(), gd.data_ptr(), s)
        torch.cuda.synchronize()
        assert np.array_equal(gd.cpu().numpy(), d) and np.array_equal(gi.cpu().numpy(), ids)


def test_cluster_shards_multi_batch(gpu_pkg):
    """Sharded index + multi-batch launches (what bench.py --gpus N runs per rank): lists that are not resident on a
    shard have length 0 there and must drop out of the work plan; merged shards equal the unsharded result."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=16000, nlist=64, seed=6)
    nb, k, nprobe, world = 5, 5, 16, 4
    q = gpu_pkg.synth_sift(32 * nb, seed=67)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)

    def run(rank, w):
        with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o,
                              rank=rank, world=w) as ivf:
            ids = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
            d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, ids.data_ptr(), d.data_ptr(), s)
            torch.cuda.synchronize()
            return d, ids
    d_all, i_all = run(0, 1)
    parts = [run(r, world) for r in range(world)]
    gd = torch.stack([p[0] for p in parts]).contiguous()
    gi = torch.stack([p[1] for p in parts]).contiguous()
    od = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
    oi = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
    gpu_pkg.topk_merge_dev(gd.data_ptr(), gi.data_ptr(), world, nb * 32, k, k, od.data_ptr(), oi.data_ptr(), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(od, d_all)
    a, b = i_all.cpu().numpy(), oi.cpu().numpy()
    for i in range(nb * 32):
        assert sorted(a[i].tolist()) == sorted(b[i].tolist())


@pytest.mark.parametrize("nprobe", [8, 32])
def test_nlist1024_weakly_clustered_data_recall_below_one(gpu_pkg, nprobe):
    """The same shape on the SECOND synthetic distribution (WEAK_MIXTURE: 65 536 centres, sixteen times as many as lists, so
    a query's neighbours straddle list boundaries): recall is well below 1 here, so that "GPU recall == oracle recall"
    says something about the scan and the probe selection -- on the strongly clustered default set every method scores
    ~ 1.  GPU recall@1 / @5 within 1 % of the oracle's, both clearly below 1 at nprobe 8; the candidate statistic
    (main_ivf.cpp:198) near nprobe * N / nlist; exact integer distances for every returned id."""
    n, nlist, k = 150_000, 1024, 5
    key = ("weak", n, nlist)
    if key not in _INDEX_CACHE:
        base = gpu_pkg.synth_mixture(n, 51, **gpu_pkg.WEAK_MIXTURE)
        _INDEX_CACHE[key] = (base,) + tuple(gpu_pkg.ivf_build(base, nlist, max_iter=8, seed=42))
    base, vr, off, r2o, cents, _ = _INDEX_CACHE[key]
    q = gpu_pkg.synth_mixture(8 * 32, 53, **gpu_pkg.WEAK_MIXTURE)
    oi, od, ototal = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe)
    gt, _ = oracle.search_bf(base, q, k)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, d, total = ivf.searchBatch(q, len(q), k, nprobe)
    same = np.array([np.array_equal(d[i], od[i]) and np.array_equal(ids[i], oi[i]) for i in range(len(q))])
    assert same.mean() >= 0.95, same.mean()  # (the rest: last-bit coarse ties on non-integer centroids pick another list)
    assert abs(total - ototal) <= 0.02 * ototal
    per_query = total / len(q)
    assert 0.6 * nprobe * n / nlist <= per_query <= 2.0 * nprobe * n / nlist, per_query  # (queries sit in the denser lists)
    ex = oracle.exact_int_dists(q, base)
    assert (ids >= 0).all() and np.array_equal(np.take_along_axis(ex, ids.astype(np.int64), 1).astype(np.float32), d)
    r1, r1o = oracle.recall(ids[:, :1], gt[:, :1], 1), oracle.recall(oi[:, :1], gt[:, :1], 1)
    r5, r5o = oracle.recall(ids, gt, k), oracle.recall(oi, gt, k)
    assert abs(r1 - r1o) <= 0.01 and abs(r5 - r5o) <= 0.01, (r1, r1o, r5, r5o)
    if nprobe == 8:
        assert 0.3 <= r1o <= 0.93 and r5o <= 0.9, (r1o, r5o)  # the distribution does what it is for


@pytest.mark.parametrize("nprobe", [8, 32])
def test_nlist1024_nprobe_8_and_32(gpu_pkg, nprobe):
    """BASELINE.json config 4 in shape (nlist = 1024, nprobe in {8, 32}, k = 5, batch 32) on a base the oracle covers in
    seconds: index from the native builder; >= 97 % of the queries identical to the oracle's restatement of
    IVFIndex::searchBatch (the rest: last-bit coarse ties on non-integer centroids), exact integer distances for every
    returned id, recall@1 (main_ivf.cpp:52-59 with k = 1) against exact ground truth equal to the oracle's within 1 %,
    host API == device multi-batch API."""
    import torch
    n, nlist, k = 150_000, 1024, 5
    key = ("built", n, nlist)
    if key not in _INDEX_CACHE:
        base = gpu_pkg.synth_sift(n, seed=41)
        _INDEX_CACHE[key] = (base,) + tuple(gpu_pkg.ivf_build(base, nlist, max_iter=8, seed=42))
    base, vr, off, r2o, cents, _ = _INDEX_CACHE[key]
    assert len(off) == nlist + 1 and off[-1] == n
    q = gpu_pkg.synth_sift(8 * 32, seed=43)
    oi, od, ototal = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe)
    gt, _ = oracle.search_bf(base, q, k)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ids, d, total = ivf.searchBatch(q, len(q), k, nprobe)
        dev = torch.device("cuda:0")
        qd = torch.from_numpy(q).to(dev)
        gi = torch.zeros((len(q), k), dtype=torch.int32, device=dev)
        gd = torch.zeros((len(q), k), dtype=torch.float32, device=dev)
        ivf.search_dev_multi(qd.data_ptr(), 8, 32, k, nprobe, gi.data_ptr(), gd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert np.array_equal(gd.cpu().numpy(), d) and np.array_equal(gi.cpu().numpy(), ids)
    same = np.array([np.array_equal(d[i], od[i]) and np.array_equal(ids[i], oi[i]) for i in range(len(q))])
    assert same.mean() >= 0.97, same.mean()
    assert abs(total - ototal) <= 0.02 * ototal
    ex = oracle.exact_int_dists(q, base)
    valid = ids >= 0
    assert valid.all()
    assert np.array_equal(np.take_along_axis(ex, ids.astype(np.int64), 1).astype(np.float32), d)
    r1, r1o = oracle.recall(ids[:, :1], gt[:, :1], 1), oracle.recall(oi[:, :1], gt[:, :1], 1)
    assert abs(r1 - r1o) <= 0.01
    assert abs(oracle.recall(ids, gt, k) - oracle.recall(oi, gt, k)) <= 0.01
    if nprobe == 32:
        assert r1 >= 0.91  # the north-star's bar


def test_library_collective_world_1_ivf(gpu_pkg):
    """vs_ivf_search_dev_sharded / vs_ivf_search_sharded over a one-rank RCCL communicator == the unsharded calls."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=16000, nlist=64, seed=6)
    nb, k, nprobe = 37, 5, 16
    q = gpu_pkg.synth_sift(32 * nb, seed=68)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.Comm(gpu_pkg.Comm.unique_id(), 0, 1, 0) as comm, \
            gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        a_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
        a_i = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
        b_d, b_i = torch.zeros_like(a_d), torch.zeros_like(a_i)
        ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, a_i.data_ptr(), a_d.data_ptr(), s)
        ivf.search_dev_sharded(comm, qd.data_ptr(), nb, 32, k, nprobe, b_i.data_ptr(), b_d.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(a_d, b_d)
        a, b = a_i.cpu().numpy(), b_i.cpu().numpy()
        for i in range(nb * 32):  # equal distances may swap ids between the two merges: compare as sets
            assert sorted(a[i].tolist()) == sorted(b[i].tolist())
        ids, d, total = ivf.searchBatch(q[:200], 200, k, nprobe)
        sid, sd, stotal = ivf.searchBatch_sharded(comm, q[:200], k, nprobe)
        assert np.array_equal(sd, d) and stotal == total


def test_native_index_builder_end_to_end(gpu_pkg, tmp_path):
    """vs_ivf_build_index -> vs_ivf_save -> vs_ivf_load with no Python in the layout step (create_ivf_model_reordered.py:
    82-177): the directory holds the reference's files, the reordered layout is the stable sort by cluster of the
    assignment, and the loaded index searches like the built one."""
    base = gpu_pkg.synth_sift(30000, seed=61)
    ivf, n_it = gpu_pkg.IVFIndex.build(base, 64, max_iter=5, seed=42)
    with ivf:
        assert ivf.getNumClusters() == 64 and ivf.getNumVectors() == 30000 and n_it >= 1
        d = str(tmp_path / "idx")
        ivf.save(d)
        q = gpu_pkg.synth_sift(64, seed=62)
        ids, dd, total = ivf.searchBatch(q, 64, 5, 8)
    for f in ("ivf_config.json", "cluster_offsets.npy", "vectors_reordered.npy", "reorder_to_original.npy", "centroids.npy",
              "cluster_sizes.npy"):
        assert os.path.exists(os.path.join(d, f)), f
    off = np.load(os.path.join(d, "cluster_offsets.npy"))
    r2o = np.load(os.path.join(d, "reorder_to_original.npy"))
    vr = np.load(os.path.join(d, "vectors_reordered.npy"))
    cents = np.load(os.path.join(d, "centroids.npy"))
    assert off[0] == 0 and off[-1] == 30000 and np.all(np.diff(off) >= 0)
    assert np.array_equal(np.sort(r2o), np.arange(30000)) and np.array_equal(vr, base[r2o])
    # every row sits in the list of its nearest centroid (float64 check, ties aside), rows of a list keep their order
    x = base.astype(np.float64)
    dist = (x ** 2).sum(1)[:, None] - 2 * x @ cents.astype(np.float64).T + (cents.astype(np.float64) ** 2).sum(1)[None]
    lst = np.searchsorted(off, np.arange(30000), side="right") - 1
    assert (dist.argmin(1)[r2o] == lst).mean() > 0.999
    for c in range(64):
        seg = r2o[off[c]:off[c + 1]]
        assert np.all(np.diff(seg) > 0)
    with gpu_pkg.IVFIndex(d) as ivf2:
        ids2, dd2, total2 = ivf2.searchBatch(q, 64, 5, 8)
    assert np.array_equal(ids2, ids) and np.array_equal(dd2, dd) and total2 == total


def test_c5_shape_eight_list_shards(gpu_pkg):
    """BASELINE.json config 5 in shape: nlist = 1024, nprobe = 32, k = 5, batch 32, lists dealt to 8 ranks (longest
    first, round robin).  Eight shard indexes on one GPU stand in for the ranks; each runs the multi-batch device call the
    ranks run, the per-shard top-k lists are merged exactly as the all-gather + merge of vs_ivf_search_dev_sharded does
    (vs_topk_merge_dev over [world][n][k]) and must equal the unsharded result; every list is owned exactly once and the
    rows per rank are balanced within 5 %."""
    import torch
    n, nlist, k, nprobe, world, nb = 150_000, 1024, 5, 32, 8, 6
    key = ("built", n, nlist)
    if key not in _INDEX_CACHE:
        base = gpu_pkg.synth_sift(n, seed=41)
        _INDEX_CACHE[key] = (base,) + tuple(gpu_pkg.ivf_build(base, nlist, max_iter=8, seed=42))
    base, vr, off, r2o, cents, _ = _INDEX_CACHE[key]
    owners = gpu_pkg.ivf_list_owners(off, world)
    sizes = np.diff(off)
    rows_per_rank = np.array([sizes[owners == r].sum() for r in range(world)])
    assert rows_per_rank.sum() == n and rows_per_rank.max() <= 1.05 * rows_per_rank.mean()
    dev = torch.device("cuda:0")
    q = gpu_pkg.synth_sift(nb * 32, seed=44)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream

    def run(rank, w):
        with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o,
                              rank=rank, world=w) as ivf:
            ids = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
            d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, ids.data_ptr(), d.data_ptr(), s)
            torch.cuda.synchronize()
            return d, ids
    d_all, i_all = run(0, 1)
    parts = [run(r, world) for r in range(world)]
    gd = torch.stack([p[0] for p in parts]).contiguous()
    gi = torch.stack([p[1] for p in parts]).contiguous()
    od = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
    oi = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
    gpu_pkg.topk_merge_dev(gd.data_ptr(), gi.data_ptr(), world, nb * 32, k, k, od.data_ptr(), oi.data_ptr(), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(od, d_all)
    a, b = i_all.cpu().numpy(), oi.cpu().numpy()
    ex = oracle.exact_int_dists(q, base)
    assert np.array_equal(np.take_along_axis(ex, b.astype(np.int64), 1).astype(np.float32), od.cpu().numpy())
    for i in range(nb * 32):
        assert sorted(a[i].tolist()) == sorted(b[i].tolist())
    gt, _ = oracle.search_bf(base, q, k)
    assert oracle.recall(b[:, :1], gt[:, :1], 1) >= 0.91


def test_wide_groups_on_two_streams_hot_lists(gpu_pkg):
    """A call of more than 32 batches: the launch groups (32 + 32 + 6 batches) run on two internal streams with a
    scratch lane each.  Few lists, so every list is probed by hundreds of a group's queries: the plan splits their
    records by slot range and the scan's loop over more than four column blocks runs.  Must equal the same batches
    sent group by group (one stream), on repeated calls, and the oracle."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=30000, nlist=16, seed=21)
    nb, k, nprobe = 70, 5, 4
    q = gpu_pkg.synth_sift(32 * nb, seed=78)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        want_i = torch.zeros((nb * 32, k), dtype=torch.int32, device=dev)
        want_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
        for b0 in range(0, nb, 32):
            n = min(32, nb - b0)
            ivf.search_dev_multi(qd.data_ptr() + b0 * 32 * 128 * 4, n, 32, k, nprobe, want_i.data_ptr() + b0 * 32 * k * 4,
                                 want_d.data_ptr() + b0 * 32 * k * 4, s)
        torch.cuda.synchronize()
        for _ in range(3):
            got_i = torch.full((nb * 32, k), -7, dtype=torch.int32, device=dev)
            got_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
            torch.cuda.synchronize()
            assert torch.equal(got_d, want_d) and torch.equal(got_i, want_i)
    sub = np.r_[0:64, 32 * 40:32 * 40 + 64, 32 * 69:32 * 70]  # queries of all three groups
    oi, od, _ = oracle.ivf_search(vr, off, r2o, cents, q[sub], k, nprobe)
    gd = want_d.cpu().numpy()[sub]
    same = np.array([np.array_equal(gd[i], od[i]) for i in range(len(sub))])
    assert same.mean() >= 0.97  # (probe sets can differ by a last-bit coarse tie, see test_matches_oracle_ivf)
    ex = oracle.exact_int_dists(q[sub], base)
    assert np.array_equal(np.take_along_axis(ex, want_i.cpu().numpy()[sub].astype(np.int64), 1).astype(np.float32), gd)


@pytest.mark.parametrize("nb,B", [(70, 32), (40, 20)])
def test_fp32_rows_scan_kernel_hot_lists_equals_int8_rows(gpu_pkg, nb, B):
    """`vs_set_precision(h, 1)` sends the list scan to its fp32 kernel (`ivf_scan_wide_f32_kernel`: query fragments by
    LDS-DMA one column block ahead, rows one record ahead, hand-counted waits).  Few lists, so every list is probed by
    hundreds of a super-batch's queries: records with far more than the four column blocks whose slots travel with the
    rows, several launch groups and super-batches, a ragged batch size, waves without a record.  On integer data the
    fp32 distances are exact: ids and distances must equal the int8-rows scan's bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=30000, nlist=16, seed=23)
    k, nprobe = 5, 4
    q = gpu_pkg.synth_sift(B * nb, seed=79)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        out = []
        for precision in (0, 1, 1):
            ivf.set_precision(precision)
            got_i = torch.full((nb * B, k), -7, dtype=torch.int32, device=dev)
            got_d = torch.zeros((nb * B, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, B, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
            torch.cuda.synchronize()
            out.append((got_i.cpu().numpy(), got_d.cpu().numpy()))
    for gi, gd in out[1:]:
        assert np.array_equal(gd, out[0][1]) and np.array_equal(gi, out[0][0])
    ex = oracle.exact_int_dists(q[:200], base)
    assert np.array_equal(np.take_along_axis(ex, out[1][0][:200].astype(np.int64), 1).astype(np.float32), out[1][1][:200])


def test_wide_slow_path_duplicates_and_tiny_lists(gpu_pkg):
    """Queries the wide pipeline cannot bound or whose candidates do not fit their lists go through the exact slow path
    inside the ranking launch: (i) thousands of identical rows (every one of them is under the bound: the sub-lists
    overflow), (ii) lists with fewer than k rows as the two nearest (no bound at all)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    base = gpu_pkg.synth_sift(12000, seed=33)
    base[2000:8000] = base[1999]                      # 6000 copies of one row
    cents = base[rng.choice(12000, 24, replace=False)].copy()
    cents[0] = base[1999]
    # eight far-away centroids with two rows each: lists with fewer than k rows
    far = (base[:16].reshape(8, 2, 128).mean(1) * 0 + np.arange(8)[:, None] * 3 + 200).astype(np.float32)
    far_rows = np.repeat(far, 2, axis=0) + rng.integers(0, 2, (16, 128)).astype(np.float32)
    base = np.concatenate([base, np.clip(far_rows, 0, 255)]).astype(np.float32)
    cents = np.concatenate([cents, far]).astype(np.float32)
    d = (base ** 2).sum(1)[:, None] - 2 * base @ cents.T + (cents ** 2).sum(1)[None]
    a = d.argmin(1)
    vr, off, r2o = gpu_pkg.ivf_layout_from_assignment(base, a, len(cents))
    k, nprobe, nb = 5, 3, 4
    q = gpu_pkg.synth_sift(32 * nb, seed=34)
    q[:40] = np.clip(base[1999] + rng.integers(-2, 3, (40, 128)), 0, 255)   # next to the duplicates
    q[40:60] = np.clip(far[rng.integers(0, 8, 20)] + rng.integers(0, 2, (20, 128)), 0, 255)  # next to the tiny lists
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    oi, od, _ = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        got_i = torch.full((nb * 32, k), -7, dtype=torch.int32, device=dev)
        got_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
        ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
        torch.cuda.synchronize()
    gi, gd = got_i.cpu().numpy(), got_d.cpu().numpy()
    same = np.array([np.array_equal(gd[i], od[i]) for i in range(len(q))])
    assert same.mean() >= 0.97, np.nonzero(~same)[0][:10]
    ex = oracle.exact_int_dists(q, base)
    ok = gi >= 0
    assert np.array_equal(np.where(ok, np.take_along_axis(ex, np.maximum(gi, 0).astype(np.int64), 1).astype(np.float32), np.inf), gd)
    assert all(len(set(r[r >= 0])) == (r >= 0).sum() for r in gi)   # no row twice


@pytest.mark.parametrize("nprobe", [4, 32])
def test_probe_selection_with_masses_of_equal_scores(gpu_pkg, nprobe):
    """400 of the 512 centroids are copies of one point: every query near it sees 400 equal coarse scores, far more
    than nprobe.  The reference's order among equal scores is ascending list id (deterministic restatement of
    std::nth_element, IVFIndex.cpp:711); the selection's bound-then-rank must give exactly that, and the bound tables
    must cope with probed lists that are nearly all empty.  Results must equal the oracle's query by query."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    base = gpu_pkg.synth_sift(20000, seed=91)
    cents = base[rng.choice(20000, 512, replace=False)].copy()
    cents[100:500] = cents[100]                       # 400 identical centroids (lists 101 .. 499 end up empty)
    d = (base ** 2).sum(1)[:, None] - 2 * base @ cents.T + (cents ** 2).sum(1)[None]
    vr, off, r2o = gpu_pkg.ivf_layout_from_assignment(base, d.argmin(1), len(cents))
    nb, k = 33, 5
    q = gpu_pkg.synth_sift(32 * nb, seed=92)
    q[:600] = np.clip(cents[100] + rng.integers(-3, 4, (600, 128)), 0, 255)   # the copies are these queries' nearest lists
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    oi, od, _ = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        got_i = torch.full((nb * 32, k), -7, dtype=torch.int32, device=dev)
        got_d = torch.zeros((nb * 32, k), dtype=torch.float32, device=dev)
        ivf.search_dev_multi(qd.data_ptr(), nb, 32, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
        torch.cuda.synchronize()
    gd = got_d.cpu().numpy()
    same = np.array([np.array_equal(gd[i], od[i]) for i in range(len(q))])
    assert same[:600].all(), np.nonzero(~same[:600])[0][:10]   # (all-equal scores: no last-bit coarse ties to excuse)
    assert same.mean() >= 0.97


@pytest.mark.parametrize("world,nb,B", [(2, 5, 32), (2, 150, 32), (4, 37, 32), (8, 70, 20), (8, 3, 32), (8, 256, 32)])
def test_sliced_pipeline_virtual_ranks_equal_unsharded(gpu_pkg, world, nb, B):
    """The cluster-sharded pipeline (BASELINE configs[4]) on virtual ranks: `world` shards of one index on one GPU go
    through vs_ivf_search_dev_sharded's sliced pipeline -- rank r runs coarse + pick + bound for slice r only, the blocks
    are exchanged, every rank fills its slot tables for all slices, plans, scans its lists, ranks; top-k lists merged --
    with the two collectives replaced by writes into the gathered layout (vs_ivf_search_dev_vshards).  Distances must equal
    the unsharded index's bit for bit (ids as sets per query: equal distances come out in a different order).  Covers a
    launch group with fewer batches than ranks (8, 3), ragged slices, several groups per call and a full 8 x 32-batch group."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=60000, nlist=256, seed=16)
    k, nprobe = 5, 24
    q = gpu_pkg.synth_sift(nb * B, seed=500 + nb)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)
    want_i = torch.zeros((nb * B, k), dtype=torch.int32, device=dev)
    want_d = torch.zeros((nb * B, k), dtype=torch.float32, device=dev)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ivf.search_dev_multi(qd.data_ptr(), nb, B, k, nprobe, want_i.data_ptr(), want_d.data_ptr(), s)
        torch.cuda.synchronize()
    shards = [gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o, rank=r, world=world)
              for r in range(world)]
    try:
        for timed in (False, True):
            got_i = torch.full((nb * B, k), -7, dtype=torch.int32, device=dev)
            got_d = torch.full((nb * B, k), -7.0, dtype=torch.float32, device=dev)
            ms = gpu_pkg.IVFIndex.search_dev_vshards(shards, qd.data_ptr(), nb, B, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s, timed=timed)
            torch.cuda.synchronize()
            assert torch.equal(got_d, want_d), f"distances differ (world={world}, nb={nb}, B={B}, timed={timed})"
            a, b = want_i.cpu().numpy(), got_i.cpu().numpy()
            assert all(sorted(a[i].tolist()) == sorted(b[i].tolist()) for i in range(nb * B))
            if timed:
                assert len(ms) == world and all(m > 0 for m in ms)
    finally:
        for sh in shards:
            sh.close()


@pytest.mark.parametrize("world,nb,B", [(4, 37, 32), (8, 70, 20)])
def test_sliced_pipeline_on_fp32_rows_equals_unsharded(gpu_pkg, world, nb, B):
    """The sliced pipeline with `vs_set_precision(1)` on every shard: the front half's bounds come from the fp32 list heads,
    the back half's scan is the fp32-rows kernel reading the exchanged bounds (not the segment lists) -- the one combination
    of that kernel the unsharded tests do not reach.  On integer data: bit-identical to the unsharded int8-rows result."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=60000, nlist=256, seed=16)
    k, nprobe = 5, 24
    q = gpu_pkg.synth_sift(nb * B, seed=600 + nb)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)
    want_i = torch.zeros((nb * B, k), dtype=torch.int32, device=dev)
    want_d = torch.zeros((nb * B, k), dtype=torch.float32, device=dev)
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ivf.search_dev_multi(qd.data_ptr(), nb, B, k, nprobe, want_i.data_ptr(), want_d.data_ptr(), s)
        torch.cuda.synchronize()
    shards = [gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o, rank=r, world=world)
              for r in range(world)]
    try:
        for sh in shards:
            sh.set_precision(1)
        got_i = torch.full((nb * B, k), -7, dtype=torch.int32, device=dev)
        got_d = torch.full((nb * B, k), -7.0, dtype=torch.float32, device=dev)
        gpu_pkg.IVFIndex.search_dev_vshards(shards, qd.data_ptr(), nb, B, k, nprobe, got_i.data_ptr(), got_d.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(got_d, want_d)
        a, b = want_i.cpu().numpy(), got_i.cpu().numpy()
        assert all(sorted(a[i].tolist()) == sorted(b[i].tolist()) for i in range(nb * B))
    finally:
        for sh in shards:
            sh.close()


def test_launch_groups_of_several_super_batches_and_fp32_rows(gpu_pkg):
    """(i) VSEARCH_IVF_GROUP (read when an index is created): launch groups of 128 batches = 4 super-batches per kernel
    launch must give what groups of 32 give.  (ii) vs_set_precision(1) on an IVF index: the list scan on the fp32 rows
    (IVFIndex.cpp:270-358's arithmetic) equals the exact-int8 scan of the same byte-valued rows bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=60000, nlist=256, seed=16)
    nb, B, k, nprobe = 150, 32, 5, 16
    q = gpu_pkg.synth_sift(nb * B, seed=77)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)

    def run(group, precision):
        old = os.environ.get("VSEARCH_IVF_GROUP")
        os.environ["VSEARCH_IVF_GROUP"] = str(group)
        try:
            ivf = gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o)
        finally:
            if old is None:
                del os.environ["VSEARCH_IVF_GROUP"]
            else:
                os.environ["VSEARCH_IVF_GROUP"] = old
        with ivf:
            ivf.set_precision(precision)
            gi = torch.zeros((nb * B, k), dtype=torch.int32, device=dev)
            gd = torch.zeros((nb * B, k), dtype=torch.float32, device=dev)
            ivf.search_dev_multi(qd.data_ptr(), nb, B, k, nprobe, gi.data_ptr(), gd.data_ptr(), s)
            torch.cuda.synchronize()
            return gi.cpu().numpy(), gd.cpu().numpy()

    i32, d32 = run(32, 0)
    i128, d128 = run(128, 0)
    assert np.array_equal(d32, d128) and np.array_equal(i32, i128)
    i256, d256 = run(256, 0)
    assert np.array_equal(d32, d256) and np.array_equal(i32, i256)
    if32, df32 = run(64, 1)
    assert np.array_equal(d32, df32)
    assert all(sorted(i32[i].tolist()) == sorted(if32[i].tolist()) for i in range(nb * B))


def test_inner_product_metric_matches_the_reference_ranking(gpu_pkg):
    """vs_ivf_set_metric(VS_METRIC_IP): the reference's own IVF ranking (IVFIndex.cpp:449-496, :697-723: nprobe lists of
    largest q.c, k candidates of largest q.v) against the oracle's restatement of it; full probe == brute-force inner
    product.  Non-negative integer data: every dot product is an exact integer."""
    base, cents, vr, off, r2o = _make_index(gpu_pkg, n=20000, nlist=64, seed=3)
    q = gpu_pkg.synth_sift(70, seed=92)
    k = 5
    with gpu_pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ivf.set_metric(1)
        for nprobe in (4, 16, 64):
            ids, sc, total = ivf.searchBatch(q, len(q), k, nprobe)
            oi, od, ototal = oracle.ivf_search(vr, off, r2o, cents, q, k, nprobe, metric=1)
            same = np.array([np.array_equal(sc[i], -od[i]) for i in range(len(q))])
            assert same.mean() >= 0.95, (nprobe, same.mean())  # (the rest: last-bit ties between non-integer centroid scores)
            dots = (q.astype(np.float64) @ base.astype(np.float64).T)
            assert np.array_equal(np.take_along_axis(dots, ids.astype(np.int64), 1).astype(np.float32), sc)
            assert (np.diff(sc, axis=1) <= 0).all()
            if nprobe == 64:
                want = -np.sort(-dots, axis=1)[:, :k]
                assert np.array_equal(sc, want.astype(np.float32)) and total == len(q) * len(base)
        ivf.set_metric(0)
        ids2, d2, _ = ivf.searchBatch(q, len(q), k, 16)
        oi2, od2, _ = oracle.ivf_search(vr, off, r2o, cents, q, k, 16)
        assert np.mean([np.array_equal(d2[i], od2[i]) for i in range(len(q))]) >= 0.97
