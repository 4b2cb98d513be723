"""Parity of the HIP path with the CPU oracle, through the C ABI (libvsearch_hip.so).
Integer-valued data: bit-exact ids AND distances, including the reference's tie order.
Non-integer data: ids equal wherever the oracle's own distance gap exceeds fp32 re-association
error, distances within rtol 2e-6 * max(||q||^2 + ||b||^2) (the tolerance is in the test)."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _int_data(rng, n, nq, hi=219):
    return (rng.integers(0, hi, size=(n, 128)).astype(np.float32),
            rng.integers(0, hi, size=(nq, 128)).astype(np.float32))


def _check_exact(pkg, base, q, k, batch=None):
    """Both data paths: precision 1 = fp32 rows + fp32 MFMA (the reference arithmetic), 0 = auto, which
    scans integer-valued bases as bytes with int8 MFMA.  Both must reproduce the oracle bit for bit."""
    oi, od = oracle.search_bf(base, q, k)
    with pkg.BruteForceIndex(base) as idx:
        if batch:
            idx.set_batch(batch)
        for precision in (1, 0):
            idx.set_precision(precision)
            ids, d = idx.search(q, k)
            assert np.array_equal(ids, oi), f"ids differ (N={len(base)}, nq={len(q)}, k={k}, batch={batch}, precision={precision})"
            assert np.array_equal(d, od), f"dists differ (precision={precision})"


@pytest.mark.parametrize("tag", ["fwd", "rev"])
def test_reference_tie_probes(gpu_pkg, golden_dir, tag):
    b = gpu_pkg.read_fvecs(os.path.join(golden_dir, f"ref_ties_{tag}_base.fvecs"))
    q = gpu_pkg.read_fvecs(os.path.join(golden_dir, f"ref_ties_{tag}_query.fvecs"))
    rid, rd = oracle.parse_results_txt(os.path.join(golden_dir, f"ref_ties_{tag}_results.txt"))
    with gpu_pkg.BruteForceIndex(b) as idx:
        tm = gpu_pkg.Timing()
        ids, d = idx.search(q, 5, tm)
    assert np.array_equal(ids, np.array(rid)) and np.array_equal(d, np.array(rd, dtype=np.float32))
    assert tm.tie_queries == len(q)  # every query of this fixture has ties inside the top-(k+1)


def test_reference_synth10k(gpu_pkg, golden_dir, tmp_path):
    z = np.load(os.path.join(golden_dir, "ref_synth10k_inputs.npz"))
    base, query = z["base"].astype(np.float32), z["query"].astype(np.float32)
    with gpu_pkg.BruteForceIndex(base) as idx:
        ids, d = idx.search(query, 5)
    out = str(tmp_path / "siftsmall_results.txt")
    gpu_pkg.write_results(out, ids, d)
    # byte-identical to what the reference binary wrote
    assert open(out).read() == open(os.path.join(golden_dir, "ref_synth10k_results.txt")).read()


@pytest.mark.parametrize("n", [1, 4, 5, 6, 15, 16, 17, 127, 1000, 4099])
def test_small_and_ragged_bases(gpu_pkg, n):
    rng = np.random.default_rng(n)
    base, q = _int_data(rng, n, 7)
    for k in (1, 5):
        _check_exact(gpu_pkg, base, q, k)


@pytest.mark.parametrize("nq,batch", [(1, 32), (2, 32), (16, 32), (17, 32), (31, 32), (32, 32), (33, 32), (100, 32),
                                      (5, 1), (40, 8), (40, 16), (35, 17)])
def test_batch_padding_paths(gpu_pkg, nq, batch):
    rng = np.random.default_rng(100 + nq + batch)
    base, q = _int_data(rng, 10000, nq)
    _check_exact(gpu_pkg, base, q, 5, batch)


@pytest.mark.parametrize("k", [1, 2, 5, 7, 8, 10, 15])
def test_k_values(gpu_pkg, k):
    rng = np.random.default_rng(200 + k)
    base, q = _int_data(rng, 6000, 33)
    _check_exact(gpu_pkg, base, q, k)


def test_k_too_large_is_refused_not_faked(gpu_pkg):
    rng = np.random.default_rng(1)
    base, q = _int_data(rng, 500, 2)
    with gpu_pkg.BruteForceIndex(base) as idx:
        with pytest.raises(gpu_pkg.VSearchError) as e:
            idx.search(q, 16)
    assert e.value.status == -5


def test_heavy_ties_and_duplicates(gpu_pkg):
    # tiny alphabet -> many equal distances; duplicated rows like SIFT-1M has
    rng = np.random.default_rng(7)
    base = rng.integers(0, 3, size=(5000, 128)).astype(np.float32)
    base[1000:1200] = base[0:200]
    base[3000:3050] = base[0]
    q = np.concatenate([base[[0, 5, 1000, 3001]], rng.integers(0, 3, size=(20, 128)).astype(np.float32)])
    for k in (1, 5, 10):
        _check_exact(gpu_pkg, base, q, k)
    # all rows identical: every distance ties
    same = np.tile(base[:1], (300, 1))
    _check_exact(gpu_pkg, same, q[:3], 5)
    # descending distances: worst case for any running-threshold scheme
    ramp = np.zeros((4000, 128), dtype=np.float32)
    ramp[:, 0] = np.arange(4000, 0, -1) % 251
    _check_exact(gpu_pkg, ramp, np.zeros((3, 128), dtype=np.float32), 5)


def test_seeded_large_base_exact(gpu_pkg):
    # >= 262144 rows turns on the seed-threshold pass; check against the oracle bit for bit
    base = gpu_pkg.synth_sift(300000, seed=11)
    q = gpu_pkg.synth_sift(48, seed=12)
    q[3] = base[123456]            # exact hit
    base[250000] = base[17]        # duplicate far apart
    q[4] = base[17]
    _check_exact(gpu_pkg, base, q, 5)
    _check_exact(gpu_pkg, base, q[:5], 10)


def test_int8_path_selection_and_fallback(gpu_pkg):
    import torch
    rng = np.random.default_rng(11)
    base, q = _int_data(rng, 20000, 40)
    base[:, 5] = 255.0  # exercise the top of the byte range
    base[7] = 0.0
    q[3] = 255.0
    oi, od = oracle.search_bf(base, q, 5)
    with gpu_pkg.BruteForceIndex(base) as idx:
        idx.set_precision(2)  # int8 required: available for integer data in [0, 255]
        ids, d = idx.search(q, 5)
        assert np.array_equal(ids, oi) and np.array_equal(d, od)
        # a non-integer query in a batch: the device flags the batch (2) and vs_bf_search reruns it in fp32
        q2 = q.copy()
        q2[17, 3] += 0.5
        o2i, o2d = oracle.search_bf(base, q2, 5)
        idx.set_precision(0)
        ids2, d2 = idx.search(q2, 5)
        assert np.array_equal(ids2, o2i) and np.allclose(d2, o2d, rtol=0, atol=1e-2)
        dev = torch.device("cuda:0")
        qd = torch.from_numpy(q2[:32]).to(dev)
        oid = torch.zeros((32, 6), dtype=torch.int32, device=dev)
        odd = torch.zeros((32, 6), dtype=torch.float32, device=dev)
        fl = torch.zeros((32,), dtype=torch.int32, device=dev)
        idx.search_dev(qd.data_ptr(), 32, 5, oid.data_ptr(), odd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert (fl.cpu().numpy() == 2).all() and (oid.cpu().numpy() == -1).all()
        # out-of-range query values take the same route
        q3 = q.copy()
        q3[0, 0] = 300.0
        o3i, o3d = oracle.search_bf(base, q3, 5)
        ids3, d3 = idx.search(q3, 5)
        assert np.array_equal(ids3, o3i) and np.array_equal(d3, o3d)
    # a base that is not byte valued has no int8 copy
    with gpu_pkg.BruteForceIndex(base + 0.25) as idx:
        with pytest.raises(gpu_pkg.VSearchError) as e:
            idx.set_precision(2)
        assert e.value.status == -5
    with gpu_pkg.BruteForceIndex(base * 2.0) as idx:  # integers, but up to 436 > 255
        with pytest.raises(gpu_pkg.VSearchError):
            idx.set_precision(2)


def test_non_integer_data_within_tolerance(gpu_pkg):
    rng = np.random.default_rng(5)
    base = rng.normal(0, 1, size=(20000, 128)).astype(np.float32)
    q = rng.normal(0, 1, size=(37, 128)).astype(np.float32)
    with gpu_pkg.BruteForceIndex(base) as idx:
        ids, d = idx.search(q, 5)
    oi, od = oracle.search_bf(base, q, 5)
    scale = float((q ** 2).sum(1).max() + (base ** 2).sum(1).max())
    tol = 2e-6 * scale
    assert np.allclose(d, od, rtol=0, atol=tol)
    gaps_ok = np.ones_like(oi, dtype=bool)
    od6 = np.sort(np.stack([oracle.l2_row(q[i], base) for i in range(len(q))]), axis=1)[:, :7]
    for i in range(len(q)):
        for t in range(5):
            lo = od6[i, t] - od6[i, t - 1] if t > 0 else np.inf
            hi = od6[i, t + 1] - od6[i, t]
            gaps_ok[i, t] = min(lo, hi) > 4 * tol
    assert gaps_ok.mean() > 0.9
    assert np.array_equal(ids[gaps_ok], oi[gaps_ok])


def test_inner_product_metric(gpu_pkg):
    rng = np.random.default_rng(6)
    base, q = _int_data(rng, 3000, 10, hi=100)
    base[:, 0] += np.arange(3000) % 7  # break ties
    with gpu_pkg.BruteForceIndex(base, metric=gpu_pkg.METRIC_IP) as idx:
        ids, s = idx.search(q, 5)
    ip = q.astype(np.int64) @ base.astype(np.int64).T
    order = np.argsort(-ip, axis=1, kind="stable")[:, :5]
    assert np.array_equal(np.take_along_axis(ip, ids.astype(np.int64), 1), np.take_along_axis(ip, order, 1))
    assert np.array_equal(s, np.take_along_axis(ip, order, 1).astype(np.float32))


def test_device_level_api_and_score_matrix(gpu_pkg):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    base, q = _int_data(rng, 12345, 32)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.BruteForceIndex(base) as idx:
        for B in (1, 16, 17, 32):
            qd = torch.from_numpy(q[:B]).to(dev)
            ids = torch.full((B, 6), -7, dtype=torch.int32, device=dev)
            d = torch.zeros((B, 6), dtype=torch.float32, device=dev)
            fl = torch.zeros((B,), dtype=torch.int32, device=dev)
            idx.search_dev(qd.data_ptr(), B, 5, ids.data_ptr(), d.data_ptr(), fl.data_ptr(), s)
            torch.cuda.synchronize()
            ex = oracle.exact_int_dists(q[:B], base)
            want = np.sort(ex, axis=1)[:, :6].astype(np.float32)
            assert np.array_equal(d.cpu().numpy(), want)
            got_ids = ids.cpu().numpy().astype(np.int64)
            assert np.array_equal(np.take_along_axis(ex, got_ids, 1).astype(np.float32), want)
            flags = fl.cpu().numpy()
            assert np.array_equal(flags != 0, (want[:, 1:] == want[:, :-1]).any(1))
        # QnnRunner::executeBatchRaw analogue: the raw [B, ld] matrix
        ld = 12352
        sc = torch.full((32, ld), -1.0, dtype=torch.float32, device=dev)
        qd = torch.from_numpy(q).to(dev)
        idx.scores_dev(qd.data_ptr(), 32, sc.data_ptr(), ld, s)
        torch.cuda.synchronize()
        got = sc.cpu().numpy()
        assert np.array_equal(got[:, :12345], oracle.exact_int_dists(q, base).astype(np.float32))
        assert np.all(got[:, 12345:] == -1.0)  # padding columns untouched


def test_topk_merge_equals_unsharded(gpu_pkg):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    base, q = _int_data(rng, 9000, 32, hi=40)
    s = torch.cuda.current_stream().cuda_stream
    qd = torch.from_numpy(q).to(dev)
    K1 = 6
    def run(b, off):
        with gpu_pkg.BruteForceIndex(b, id_offset=off) as idx:
            ids = torch.zeros((32, K1), dtype=torch.int32, device=dev)
            d = torch.zeros((32, K1), dtype=torch.float32, device=dev)
            idx.search_dev(qd.data_ptr(), 32, 5, ids.data_ptr(), d.data_ptr(), 0, s)
            torch.cuda.synchronize()
            return d, ids
    d_all, i_all = run(base, 0)
    for G in (2, 3, 8):
        bounds = np.linspace(0, len(base), G + 1).astype(int)
        parts = [run(base[bounds[g]:bounds[g + 1]], int(bounds[g])) for g in range(G)]
        gd = torch.stack([p[0] for p in parts]).contiguous()   # [G, B, K1] = all-gather receive layout
        gi = torch.stack([p[1] for p in parts]).contiguous()
        od = torch.zeros((32, K1), dtype=torch.float32, device=dev)
        oi = torch.zeros((32, K1), dtype=torch.int32, device=dev)
        fl = torch.zeros((32,), dtype=torch.int32, device=dev)
        gpu_pkg.topk_merge_dev(gd.data_ptr(), gi.data_ptr(), G, 32, K1, K1, od.data_ptr(), oi.data_ptr(), fl.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(od, d_all) and torch.equal(oi, i_all)


def test_seeded_multi_batch_exact(gpu_pkg):
    """Launches of >= 4 batches on a shard of >= 65536 rows take their bounds from the seed launches (int8 seed on
    integer data) and stream from the first tile on.  7 full batches + a ragged one against the oracle, both
    precisions, plus exact hits, far-apart duplicates and a batch with a non-integer query (no bound from the int8
    seed for that batch; the int8 scan skips it and the host reruns it in fp32)."""
    base = gpu_pkg.synth_sift(150000, seed=21)
    q = gpu_pkg.synth_sift(7 * 32 + 9, seed=22)
    q[3] = base[123456]
    base[140000] = base[17]
    q[40] = base[17]
    _check_exact(gpu_pkg, base, q, 5)
    _check_exact(gpu_pkg, base, q[:160], 10)
    q2 = q.copy()
    q2[70, 5] += 0.5  # batch 2 is no longer byte valued
    oi, od = oracle.search_bf(base, q2, 5)
    with gpu_pkg.BruteForceIndex(base) as idx:
        for precision in (1, 0):
            idx.set_precision(precision)
            ids, d = idx.search(q2, 5)
            assert np.array_equal(ids, oi), f"precision {precision}"
            assert np.allclose(d, od, rtol=0, atol=1e-2)


def test_seeded_multi_batch_non_integer_base(gpu_pkg):
    """The fp32 seed kernel (no int8 copy): same tolerance statement as test_non_integer_data_within_tolerance,
    on a multi-batch call."""
    rng = np.random.default_rng(15)
    base = rng.normal(0, 1, size=(70000, 128)).astype(np.float32)
    q = rng.normal(0, 1, size=(5 * 32, 128)).astype(np.float32)
    with gpu_pkg.BruteForceIndex(base) as idx:
        ids, d = idx.search(q, 5)
    oi, od = oracle.search_bf(base, q, 5)
    scale = float((q ** 2).sum(1).max() + (base ** 2).sum(1).max())
    tol = 2e-6 * scale
    assert np.allclose(d, od, rtol=0, atol=tol)
    # ids agree wherever the oracle's own gap to the neighbouring distances exceeds the re-association error
    same = (ids == oi)
    gap = np.minimum(np.abs(np.diff(od, axis=1, prepend=-np.inf)), np.abs(np.diff(od, axis=1, append=np.inf)))
    assert same[gap > 4 * tol].all() and same.mean() > 0.99


def test_tie_resolver_candidate_pass(gpu_pkg):
    """Flagged queries on a base larger than the dense prefix of the tie resolver (4096 rows): the filtered
    candidate pass + slot replay must reproduce select_topk's history-dependent order (cpu_baseline.cpp:127-153),
    including ties between rows before and after the prefix, ties at the k-th position and duplicated rows."""
    rng = np.random.default_rng(21)
    base = rng.integers(0, 40, size=(60000, 128)).astype(np.float32)   # small alphabet: ties are common
    q = rng.integers(0, 40, size=(70, 128)).astype(np.float32)
    # planted duplicates: the same vector early, in the middle and late -> equal distances across the dense prefix
    for t, src in enumerate((5, 77, 4000, 4095, 4096, 30000)):
        base[4096 + 977 * (t + 1)] = base[src]
        base[59000 - 13 * t] = base[src]
        q[t] = base[src] + (t % 2)        # nearest rows are the planted copies (distance 0 or 128)
    q[10] = base[3]                       # exact hit in the prefix
    base[50001] = base[3]
    for k in (1, 5, 10):
        oi, od = oracle.search_bf(base, q, k)
        with gpu_pkg.BruteForceIndex(base) as idx:
            for precision in (1, 0):
                idx.set_precision(precision)
                tm = gpu_pkg.Timing()
                ids, d = idx.search(q, k, tm)
                assert np.array_equal(ids, oi) and np.array_equal(d, od), f"k={k} precision={precision}"
                assert tm.tie_queries > 0  # the resolver ran (otherwise this test proves nothing)


def test_tie_resolver_overflow_falls_back_to_full_rows(gpu_pkg):
    # tiny alphabet on a base beyond the dense prefix: far more rows under the bound than candidate slots -> full-row path
    rng = np.random.default_rng(22)
    base = rng.integers(0, 2, size=(40000, 128)).astype(np.float32)
    base[20000:30000] = base[0:10000]
    q = np.concatenate([base[[0, 9999, 25000]], rng.integers(0, 2, size=(5, 128)).astype(np.float32)])
    _check_exact(gpu_pkg, base, q, 5)


def test_sift1m_brute_force_exact(gpu_pkg):
    """BASELINE.json config 3 at full size: SIFT-1M-shaped base, batch 32, k = 5 -- 64 queries against the oracle,
    ids and distances bit for bit, on the fp32 path and on the exact int8 path, host API and device API."""
    import torch
    base = gpu_pkg.synth_sift(1_000_000, seed=20251205)
    q = gpu_pkg.synth_sift(64, seed=20251206)
    oi, od = oracle.search_bf(base, q, 5)
    with gpu_pkg.BruteForceIndex(base) as idx:
        for precision in (1, 0):
            idx.set_precision(precision)
            ids, d = idx.search(q, 5)
            assert np.array_equal(ids, oi) and np.array_equal(d, od), f"precision={precision}"
        # the device-level multi-batch call the bench times: k + 1 best by (dist, id); flags mark ties
        idx.set_precision(1)
        dev = torch.device("cuda", 0)
        qd = torch.from_numpy(q).to(dev)
        o_d = torch.zeros((64, 6), dtype=torch.float32, device=dev)
        o_i = torch.zeros((64, 6), dtype=torch.int32, device=dev)
        fl = torch.zeros((64,), dtype=torch.int32, device=dev)
        idx.search_dev_multi(qd.data_ptr(), 2, 32, 5, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        keep = fl.cpu().numpy() == 0
        assert keep.sum() >= 60
        assert np.array_equal(o_i.cpu().numpy()[keep, :5], oi[keep]) and np.array_equal(o_d.cpu().numpy()[keep, :5], od[keep])
        # BASELINE.json config 2's shape at full size: one query per call (cpu_baseline.cpp:222 runs one query per
        # iteration) -- the single-call scan, one launch each, back to back on one stream
        o_d.fill_(-1.0)
        o_i.fill_(-7)
        fl.fill_(-7)
        for i in range(12):
            idx.search_dev(qd.data_ptr() + i * 128 * 4, 1, 5, o_i.data_ptr() + i * 6 * 4, o_d.data_ptr() + i * 6 * 4,
                           fl.data_ptr() + i * 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        f = fl.cpu().numpy()[:12]
        assert set(np.unique(f)) <= {0, 1}
        assert np.array_equal(o_d.cpu().numpy()[:12, :5], od[:12])
        assert np.array_equal(o_i.cpu().numpy()[:12][f == 0, :5], oi[:12][f == 0])


def test_cli_no_arguments_reproduces_reference_run(gpu_pkg, golden_dir, tmp_path):
    """`vsearch_bf` with no arguments = the reference's hard-coded run (cpu_baseline.cpp:323-345): k = 5,
    siftsmall/siftsmall_{base,query}.fvecs relative to the CWD -> siftsmall_results.txt, byte for byte what the
    reference wrote for the same inputs; the missing sift/ dataset is reported and skipped, exit code 0 (:194-197, :351)."""
    import subprocess
    z = np.load(os.path.join(golden_dir, "ref_synth10k_inputs.npz"))
    os.makedirs(tmp_path / "siftsmall")
    gpu_pkg.write_fvecs(str(tmp_path / "siftsmall" / "siftsmall_base.fvecs"), z["base"].astype(np.float32))
    gpu_pkg.write_fvecs(str(tmp_path / "siftsmall" / "siftsmall_query.fvecs"), z["query"].astype(np.float32))
    exe = os.path.join(os.path.dirname(gpu_pkg.LIB_PATH), "vsearch_bf")
    assert os.path.exists(exe), "vsearch_bf not built (make -C hai-25-rag-on-edge_amd/csrc all)"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = gpu_pkg.hip_runtime_dir() + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Failed to load base file!" in r.stderr  # sift/ is absent, like a checkout without the 1M set
    got = open(tmp_path / "siftsmall_results.txt").read()
    assert got == open(os.path.join(golden_dir, "ref_synth10k_results.txt")).read()
    # brute-force metrics.txt in the layout of qidk_bruteforce main.cpp:321-390
    m = open(tmp_path / "siftsmall_metrics.txt").read()
    for needle in ("Performance Metrics (Batched) ===", "Dataset Information:", "Number of queries: 100", "Number of documents: 10000",
                   "Dimension: 128", "Batch size: 32", "Number of batches: 4", "Top-K: 5", "Operational Intensity Analysis:",
                   "Overall Performance:", "Throughput:", "queries/sec", "Avg graph execute time:", "P95 graph exec time:",
                   "P99 graph exec time:", "Avg GFLOPS:", "Per-Query Amortized Performance:", "Time Breakdown (% of end-to-end):"):
        assert needle in m, needle
    assert not os.path.exists(tmp_path / "sift_results.txt")


def test_library_collective_world_1(gpu_pkg):
    """vs_comm_* + vs_bf_search_dev_sharded / vs_bf_search_sharded with a one-rank RCCL communicator: the whole
    in-library path (local scan -> ncclAllGather on the communicator's stream -> device merge, double-buffered over
    launch groups) must reproduce the unsharded calls.  More ranks need more GPUs (driver's SCALE run)."""
    import torch
    dev = torch.device("cuda", 0)
    base = gpu_pkg.synth_sift(70000, seed=51)
    nb = 70  # three launch groups: 32 + 32 + 6 -> both exchange buffers are reused
    q = gpu_pkg.synth_sift(nb * 32, seed=52)
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.Comm(gpu_pkg.Comm.unique_id(), 0, 1, 0) as comm, gpu_pkg.BruteForceIndex(base) as idx:
        for precision in (1, 0):
            idx.set_precision(precision)
            a_d = torch.zeros((nb * 32, 6), dtype=torch.float32, device=dev)
            a_i = torch.zeros((nb * 32, 6), dtype=torch.int32, device=dev)
            a_f = torch.zeros((nb * 32,), dtype=torch.int32, device=dev)
            b_d, b_i, b_f = torch.zeros_like(a_d), torch.zeros_like(a_i), torch.zeros_like(a_f)
            idx.search_dev_multi(qd.data_ptr(), nb, 32, 5, a_i.data_ptr(), a_d.data_ptr(), a_f.data_ptr(), s)
            idx.search_dev_sharded(comm, qd.data_ptr(), nb, 32, 5, b_i.data_ptr(), b_d.data_ptr(), b_f.data_ptr(), s)
            torch.cuda.synchronize()
            assert torch.equal(a_d, b_d) and torch.equal(a_i, b_i) and torch.equal(a_f, b_f)
        ids, d = idx.search(q[:300], 5)
        tm = gpu_pkg.Timing()
        sid, sd = idx.search_sharded(comm, q[:300], 5, tm)
        assert np.array_equal(sd, d)
        unflagged = np.array([len(set(d[i])) == 5 for i in range(300)])
        assert np.array_equal(sid[unflagged], ids[unflagged])


def test_bf_batch_sweep_driver_end_to_end(gpu_pkg, golden_dir, tmp_path):
    """scripts/sweep_bf.py = qidk_bruteforce/scripts/run_all.sh on this backend: batch grid {1, 8, 16, 32, 64}, one CLI
    run each (qidk argument form), CSV with the reference's columns; every run's results.txt is the reference's."""
    import csv
    import subprocess
    import sys
    z = np.load(os.path.join(golden_dir, "ref_synth10k_inputs.npz"))
    os.makedirs(tmp_path / "siftsmall")
    gpu_pkg.write_fvecs(str(tmp_path / "siftsmall" / "siftsmall_base.fvecs"), z["base"].astype(np.float32))
    gpu_pkg.write_fvecs(str(tmp_path / "siftsmall" / "siftsmall_query.fvecs"), z["query"].astype(np.float32))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = gpu_pkg.hip_runtime_dir() + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "sweep_bf.py"), "--datasets", "siftsmall", "sift",
                        "--data-root", str(tmp_path), "--out", str(tmp_path / "results")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "dataset sift:" in r.stdout and "skipped" in r.stdout
    csvs = [f for f in os.listdir(tmp_path / "results") if f.endswith(".csv")]
    rows = list(csv.DictReader(open(tmp_path / "results" / csvs[0])))
    assert [int(x["batch_size"]) for x in rows] == [1, 8, 16, 32, 64]
    ref = open(os.path.join(golden_dir, "ref_synth10k_results.txt")).read()
    for x in rows:
        assert float(x["throughput_qps"]) > 0 and float(x["gflops"]) > 0 and float(x["avg_latency_ms"]) > 0
        b = int(x["batch_size"])
        rdir = tmp_path / "results" / ("siftsmall" if b == 1 else f"siftsmall_b{b}")
        assert open(rdir / "results.txt").read() == ref
        assert f"Batch size: {b}" in open(rdir / "metrics.txt").read()


def test_wide_int8_scan_equals_fp32_path(gpu_pkg):
    """Launches of >= 4 batches on byte-valued data take the wide int8 scan (several batches per pass over the rows,
    candidates appended to per-query lists, flat merge).  Its k+1 lists and tie flags must be bit-identical to the
    fp32 path's for every batch shape: 32-query batches (two column blocks each), <= 16-query batches (one block),
    batch counts that do not fill the last pass, k + 1 beyond the lane-list sizes of the per-batch kernels, an id offset."""
    import torch
    dev = torch.device("cuda", 0)
    base = gpu_pkg.synth_sift(150000, seed=71)
    base[149000] = base[11]
    q = gpu_pkg.synth_sift(13 * 32, seed=72)
    q[5] = base[11]          # exact hit with a far-apart duplicate: tie flag
    q[40] = base[77777]
    qd = torch.from_numpy(q).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.BruteForceIndex(base, id_offset=1000) as idx:
        for nb, B, k in ((7, 32, 5), (13, 32, 10), (9, 16, 5), (13, 5, 1), (4, 32, 15), (26, 9, 5)):
            out = {}
            for precision in (1, 2):
                idx.set_precision(precision)
                o_d = torch.zeros((nb * B, k + 1), dtype=torch.float32, device=dev)
                o_i = torch.zeros((nb * B, k + 1), dtype=torch.int32, device=dev)
                fl = torch.zeros((nb * B,), dtype=torch.int32, device=dev)
                idx.search_dev_multi(qd.data_ptr(), nb, B, k, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(), s)
                torch.cuda.synchronize()
                out[precision] = (o_d.cpu().numpy(), o_i.cpu().numpy(), fl.cpu().numpy())
            assert np.array_equal(out[1][0], out[2][0]), (nb, B, k)
            assert np.array_equal(out[1][1], out[2][1]), (nb, B, k)
            assert np.array_equal(out[1][2], out[2][2]), (nb, B, k)
            assert out[1][1].min() >= 1000
        assert out[1][2].max() <= 1


def test_streaming_scan_overflow_falls_back_on_the_device(gpu_pkg):
    """More rows under a query's bound than the candidate buffers hold (here: tens of thousands of identical rows): the
    binning launch raises the overflow word and the per-batch scan + merge enqueued behind the streaming scan produce
    the result instead -- same outputs as ever, on both data paths, with no host round trip and no special flag."""
    import torch
    base = gpu_pkg.synth_sift(100000, seed=74)
    base[20000:60000] = base[7]           # 40 000 copies of one row
    q = gpu_pkg.synth_sift(5 * 32, seed=75)
    q[0] = base[7]
    q[33] = base[7] + 1
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    oi, od = oracle.search_bf(base, q, 5)
    ex = oracle.exact_int_dists(q, base)
    with gpu_pkg.BruteForceIndex(base) as idx:
        for precision in (1, 2):
            idx.set_precision(precision)
            o_d = torch.zeros((160, 6), dtype=torch.float32, device=dev)
            o_i = torch.zeros((160, 6), dtype=torch.int32, device=dev)
            fl = torch.zeros((160,), dtype=torch.int32, device=dev)
            idx.search_dev_multi(qd.data_ptr(), 5, 32, 5, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            f, gi, gd = fl.cpu().numpy(), o_i.cpu().numpy(), o_d.cpu().numpy()
            assert f.max() <= 1 and f[0] == 1 and f[33] == 1          # ties among the copies, nothing skipped
            assert np.array_equal(gd, np.sort(ex, axis=1)[:, :6].astype(np.float32))
            assert np.array_equal(np.take_along_axis(ex, gi.astype(np.int64), 1).astype(np.float32), gd)
            keep = f == 0
            assert np.array_equal(gi[keep, :5], oi[keep])
            ids, d = idx.search(q, 5)
            assert np.array_equal(ids, oi) and np.array_equal(d, od)


@pytest.mark.parametrize("nb,B", [(7, 32), (4, 20), (9, 1)])
def test_sift1m_streaming_scan_pairs_batches_exact(gpu_pkg, nb, B):
    """The graded path at full size: launches of >= 4 batches on a 1 M-row base go through the streaming fp32 scan, which
    serves two batches per pass over the rows (an odd batch count leaves a half-dead last pass; B < 32 leaves padding
    columns).  k + 1 best by (dist, id) for every query, bit for bit against the oracle."""
    import torch
    base = gpu_pkg.synth_sift(1_000_000, seed=20251205)
    q = gpu_pkg.synth_sift(nb * B, seed=424242 + nb)
    oi, od = oracle.search_bf(base, q, 5)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q).to(dev)
    with gpu_pkg.BruteForceIndex(base) as idx:
        idx.set_precision(1)
        for _ in range(2):
            o_d = torch.zeros((nb * B, 6), dtype=torch.float32, device=dev)
            o_i = torch.full((nb * B, 6), -7, dtype=torch.int32, device=dev)
            fl = torch.full((nb * B,), -7, dtype=torch.int32, device=dev)
            idx.search_dev_multi(qd.data_ptr(), nb, B, 5, o_i.data_ptr(), o_d.data_ptr(), fl.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            f = fl.cpu().numpy()
            assert set(np.unique(f)) <= {0, 1}
            keep = f == 0
            assert keep.mean() > 0.9
            gi, gd = o_i.cpu().numpy(), o_d.cpu().numpy()
            assert np.array_equal(gi[keep, :5], oi[keep]) and np.array_equal(gd[keep, :5], od[keep])
            # flagged queries (equal distances among the k + 1 best): distances still exact, ids a valid tie order
            assert np.array_equal(gd[:, :5], od)
            assert (np.diff(gd, axis=1) >= 0).all() and (gi >= 0).all()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_brute_force_exact_on_virtual_shards(gpu_pkg, world):
    """vs_bf_search_sharded's host path on virtual ranks (vs_bf_search_vshards: `world` row shards on one GPU, the
    exchanges no-ops, everything else the code a collective job runs): ids AND distances must equal the oracle's --
    select_topk's history-dependent tie order (cpu_baseline.cpp:127-153) across shard boundaries included.  Inputs: the
    heavy-tie set of test_heavy_ties_and_duplicates (duplicates spread over shards), all-equal rows, a descending ramp
    (worst case for the dense-prefix bound), and the candidate-overflow fallback (masses of duplicates)."""
    rng = np.random.default_rng(7)
    base = rng.integers(0, 3, size=(5000, 128)).astype(np.float32)
    base[1000:1200] = base[0:200]
    base[3000:3050] = base[0]
    q = np.concatenate([base[[0, 5, 1000, 3001]], rng.integers(0, 3, size=(36, 128)).astype(np.float32)])
    ramp = np.zeros((9000, 128), dtype=np.float32)
    ramp[:, 0] = np.arange(9000, 0, -1) % 251
    big = np.tile(rng.integers(0, 2, size=(25, 128)).astype(np.float32), (1600, 1))  # 40 000 rows, 25 distinct: > 8192 candidates per query
    cases = [(base, q, 5), (base, q[:7], 10), (np.tile(base[:1], (300, 1)), q[:3], 5), (ramp, np.zeros((3, 128), dtype=np.float32), 5),
             (big, np.concatenate([big[[0, 3]], rng.integers(0, 2, size=(3, 128)).astype(np.float32)]), 5)]
    for data, qq, k in cases:
        oi, od = oracle.search_bf(data, qq, k)
        bounds = gpu_pkg.row_shard_bounds(len(data), world) if len(data) >= 16 * world else np.linspace(0, len(data), world + 1).astype(int)
        shards = [gpu_pkg.BruteForceIndex(data[bounds[g]:bounds[g + 1]], id_offset=int(bounds[g])) for g in range(world) if bounds[g + 1] > bounds[g]]
        try:
            for precision in (1, 0):
                for sh in shards:
                    sh.set_precision(precision)
                tm = gpu_pkg.Timing()
                ids, d = gpu_pkg.BruteForceIndex.search_vshards(shards, qq, k, tm)
                assert np.array_equal(d, od), f"dists differ (N={len(data)}, world={world}, k={k}, precision={precision})"
                assert np.array_equal(ids, oi), f"ids differ (N={len(data)}, world={world}, k={k}, precision={precision})"
                assert tm.tie_queries > 0
        finally:
            for sh in shards:
                sh.close()


def test_sharded_shard_that_skips_a_batch_is_rerun_not_dropped(gpu_pkg):
    """One byte-valued shard (int8 scan) and one that is not (fp32 scan), a batch with a non-integer query: the byte shard's
    int8 scan skips the batch; the merged flag must say so and the call must rerun the chunk in fp32 on every shard -- never
    hand back a result that silently lacks the skipping shard's rows."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 200, size=(4000, 128)).astype(np.float32)           # byte valued: int8 copy exists
    b = rng.integers(0, 200, size=(4000, 128)).astype(np.float32) + 0.5     # not byte valued: fp32 rows only
    base = np.concatenate([a, b])
    q = rng.integers(0, 200, size=(40, 128)).astype(np.float32)
    q[3, 10] += 0.25                                                        # non-integer query in the first batch
    q[0] = a[17]                                                            # whose neighbours are in the byte shard
    od = ((q.astype(np.float64)[:, None, :] - base.astype(np.float64)[None]) ** 2).sum(-1)
    want = np.argsort(od, axis=1, kind="stable")[:, :5]
    shards = [gpu_pkg.BruteForceIndex(a, id_offset=0), gpu_pkg.BruteForceIndex(b, id_offset=4000)]
    try:
        ids, d = gpu_pkg.BruteForceIndex.search_vshards(shards, q, 5)
        assert np.array_equal(ids, want.astype(np.int32))
        assert np.allclose(d, np.take_along_axis(od, want, 1), rtol=1e-6)
    finally:
        for sh in shards:
            sh.close()


_TOGGLE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge, oracle
pkg = ge.load_package()
base = pkg.synth_sift(300000, seed=11)
q = pkg.synth_sift(5 * 32, seed=12)
q[3] = base[123456]
oi, od = oracle.search_bf(base, q, 5)
with pkg.BruteForceIndex(base) as idx:
    for precision in (1, 0):
        idx.set_precision(precision)
        ids, d = idx.search(q, 5)
        assert np.array_equal(ids, oi) and np.array_equal(d, od), precision
print("TOGGLE_OK")
"""


@pytest.mark.parametrize("env", [{"VSEARCH_STREAM": "0"}, {"VSEARCH_F32_PAIR": "0"}, {"VSEARCH_F32_PAIR": "2"}, {"VSEARCH_I8_WIDE": "4"},
                                 {"VSEARCH_I8_WIDE": "0"}, {"VSEARCH_SEED_MIN": "0"}, {"VSEARCH_SEED_I8": "0"}, {"VSEARCH_XCHG_IT": "-1"},
                                 {"VSEARCH_GRID_CUS": "64"}, {"VSEARCH_LANES": "2"}])
def test_tuning_toggles_keep_results(gpu_pkg, env):
    """Every code path a VSEARCH_* knob of the brute-force scans selects (read when the library is loaded,
    hence one subprocess each) must reproduce the oracle bit for bit: a 300 000-row base (seeded launches), 5 batches, both
    data paths.  A knob nobody tests is a shipped kernel nobody has verified."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _TOGGLE_SCRIPT, root], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TOGGLE_OK" in r.stdout, (env, r.stdout[-400:], r.stderr[-1200:])
