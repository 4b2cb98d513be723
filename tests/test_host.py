"""CPU-side checks of the product library: it loads, exports the whole C ABI, its host logic
(file formats, result grammar, synthetic data, slot emulation) matches the oracle, and every
compute entry point fails loudly when there is no GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import sys
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "vsearch.h")).read()
    declared = set(re.findall(r"VS_API\s+[\w\s\*]+?\b(vs_\w+)\s*\(", hdr))
    assert len(declared) >= 25
    L = pkg.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vsearch.h but not exported"
    assert declared == set(pkg.exported_symbols())
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported
    assert b"gfx950" in pkg.lib().vs_version()


def test_no_cpu_fallback_without_gpu(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    base = np.zeros((64, 128), dtype=np.float32)
    with pytest.raises(pkg.VSearchError) as e:
        pkg.BruteForceIndex(base)
    assert e.value.status == -3  # VS_ERR_DEVICE
    with pytest.raises(pkg.VSearchError) as e:
        pkg.IVFIndex(vectors_reordered=base, centroids=base[:4], cluster_offsets=[0, 16, 32, 48, 64])
    assert e.value.status == -3


def test_fvecs_ivecs_roundtrip_and_errors(pkg, tmp_path):
    x = np.arange(5 * 128, dtype=np.float32).reshape(5, 128)
    p = str(tmp_path / "a.fvecs")
    pkg.write_fvecs(p, x)
    assert os.path.getsize(p) == 5 * 516  # 4 + 4*128 per record
    assert np.array_equal(pkg.read_fvecs(p), x)
    assert np.array_equal(oracle.read_fvecs(p), x)
    g = np.arange(300, dtype=np.int32).reshape(3, 100)
    pi = str(tmp_path / "g.ivecs")
    pkg.write_ivecs(pi, g)
    assert np.array_equal(pkg.read_ivecs(pi), g)
    with open(p, "ab") as f:
        f.write(b"\x80\x00\x00\x00\x00")  # truncated record (cpu_baseline.cpp:53-56)
    with pytest.raises(pkg.VSearchError) as e:
        pkg.read_fvecs(p)
    assert e.value.status == -2 and "truncated" in str(e.value)
    with pytest.raises(pkg.VSearchError) as e:
        pkg.read_fvecs(str(tmp_path / "missing.fvecs"))
    assert e.value.status == -2 and "Cannot open" in str(e.value)
    # inconsistent dimension (cpu_baseline.cpp:43-46)
    bad = np.concatenate([np.array([2], np.int32).view(np.float32), np.zeros(2, np.float32),
                          np.array([3], np.int32).view(np.float32), np.zeros(2, np.float32)])
    pb = str(tmp_path / "bad.fvecs")
    bad.tofile(pb)
    with pytest.raises(pkg.VSearchError):
        pkg.read_fvecs(pb)
    # empty file: zero rows, like the reference's while-loop never running
    pe = str(tmp_path / "empty.fvecs")
    open(pe, "wb").close()
    assert pkg.read_fvecs(pe).shape[0] == 0


def test_results_grammar_matches_reference_writer(pkg, tmp_path, golden_dir):
    ids, d = oracle.parse_results_txt(os.path.join(golden_dir, "ref_synth10k_results.txt"))
    ids = np.array(ids, dtype=np.int32)
    d = np.array(d, dtype=np.float32)
    p = str(tmp_path / "r.txt")
    pkg.write_results(p, ids, d)
    assert open(p).read() == open(os.path.join(golden_dir, "ref_synth10k_results.txt")).read()
    # default ostream formatting switches to exponent above 6 digits, like "%g"
    big = np.array([[1234567.0, 0.5, 100000.0]], dtype=np.float32)
    pkg.write_results(p, np.array([[1, 2, -1]], dtype=np.int32), big)
    assert open(p).read() == "Query 0: (1, 1.23457e+06) (2, 0.5)\n"
    po = str(tmp_path / "o.txt")
    oracle.write_results(po, np.array([[1, 2, -1]], dtype=np.int32), big)
    assert open(po).read() == open(p).read()
    pkg.write_results(p, np.array([[7]], dtype=np.int32), np.array([[2.5]], dtype=np.float32), style=1)
    assert open(p).read() == "Query 0: (7, 2.5000)\n"  # main_ivf.cpp:181


def test_synth_sift_is_deterministic_and_sift_shaped(pkg):
    a = pkg.synth_sift(3000, seed=20251205)
    b = pkg.synth_sift(3000, seed=20251205)
    assert np.array_equal(a, b)
    assert np.array_equal(a[1000:1100], pkg.synth_sift(100, seed=20251205, row_begin=1000))  # shardable
    assert not np.array_equal(a[:100], pkg.synth_sift(100, seed=20251206))
    assert a.dtype == np.float32 and a.min() >= 0 and a.max() <= 218 and np.all(a == np.rint(a))
    assert 10 < a.mean() < 60
    # clustered: nearest neighbour much closer than a random pair
    ex = oracle.exact_int_dists(a[:50], a[50:])
    assert np.median(ex.min(1)) < 0.6 * np.median(ex)


def test_slot_emulation_equals_oracle(pkg):
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(1, 300))
        k = int(rng.integers(1, 8))
        d = rng.integers(0, 10, size=n).astype(np.float32)
        rows = np.arange(n, dtype=np.int32) * 3 + 1  # arbitrary increasing row numbers
        pi, pd = pkg.select_topk_slots(rows, d, k)
        oi, od = oracle.select_topk_sparse(rows, d, k)
        assert np.array_equal(pi, oi) and np.array_equal(pd, od)


def _write_index(tmp_path, n=200, nlist=4, reordered=True):
    rng = np.random.default_rng(4)
    v = rng.integers(0, 200, size=(n, 128)).astype(np.float32)
    off = np.linspace(0, n, nlist + 1).astype(np.int32)
    d = tmp_path / "idx"
    d.mkdir(exist_ok=True)
    np.save(d / "vectors_reordered.npy", v)
    np.save(d / "reorder_to_original.npy", rng.permutation(n).astype(np.int32))
    np.save(d / "cluster_offsets.npy", off)
    np.save(d / "centroids.npy", rng.random((nlist, 128)).astype(np.float32))
    import json
    json.dump({"n_vectors": n, "n_clusters": nlist, "dim": 128, "batch_size": 32,
               "avg_cluster_size": n / nlist, "reordered": reordered}, open(d / "ivf_config.json", "w"), indent=2)
    return d


def test_ivf_load_parses_reference_directory_format(pkg, tmp_path):
    d = _write_index(tmp_path)
    h = C.c_void_p(None)
    rc = pkg.lib().vs_ivf_load(str(d).encode(), 0, 0, 1, C.byref(h))
    if pkg.device_count() == 0:
        # parsing succeeded (numpy-written v1 .npy + json), then the device step failed loudly
        assert rc == -3, pkg.lib().vs_last_error()
    else:
        assert rc == 0
        pkg.lib().vs_destroy(h)
    # missing key -> the reference's "Missing n_clusters in config" (IVFIndex.cpp:193-195)
    cfg = (d / "ivf_config.json").read_text().replace("n_clusters", "n_klusters")
    (d / "ivf_config.json").write_text(cfg)
    rc = pkg.lib().vs_ivf_load(str(d).encode(), 0, 0, 1, C.byref(h))
    assert rc == -2 and b"Missing n_clusters" in pkg.lib().vs_last_error()
    # wrong dtype is rejected instead of being reinterpreted (the reference ignores descr)
    d2 = _write_index(tmp_path)
    np.save(d2 / "cluster_offsets.npy", np.array([0, 50, 100, 150, 200], dtype=np.int64))
    rc = pkg.lib().vs_ivf_load(str(d2).encode(), 0, 0, 1, C.byref(h))
    assert rc == -2 and b"dtype" in pkg.lib().vs_last_error()
    rc = pkg.lib().vs_ivf_load(str(tmp_path / "nope").encode(), 0, 0, 1, C.byref(h))
    assert rc == -2 and b"Cannot open config file" in pkg.lib().vs_last_error()


def test_argument_validation(pkg):
    L = pkg.lib()
    h = C.c_void_p(None)
    base = np.zeros((8, 64), dtype=np.float32)
    assert L.vs_bf_create(base.ctypes.data_as(C.c_void_p), 8, 64, 0, 0, 0, C.byref(h)) == -5  # dim != 128
    assert L.vs_bf_create(None, 8, 128, 0, 0, 0, C.byref(h)) == -1
    assert L.vs_set_batch(None, 4) == -1
    assert pkg.clamp_nlist(10000, 1024) == 100  # create_ivf_model_reordered.py:92-94
    assert pkg.clamp_nlist(1000000, 1024) == 1024
    v = np.arange(12, dtype=np.float32).reshape(6, 2)
    vr, off, r2o = pkg.ivf_layout_from_assignment(v, np.array([2, 0, 1, 0, 2, 2]), 3)
    assert off.tolist() == [0, 2, 3, 6] and r2o.tolist() == [1, 3, 2, 0, 4, 5]
    assert np.array_equal(vr, v[r2o])


def test_q8_argument_validation(pkg):
    L = pkg.lib()
    h = C.c_void_p(None)
    base = np.zeros((8, 128), dtype=np.float32)
    bp = base.ctypes.data_as(C.c_void_p)
    assert L.vs_q8_create(None, 8, 128, None, 0, 0, C.byref(h)) == -1
    assert L.vs_q8_create(bp, 8, 64, None, 0, 0, C.byref(h)) == -5  # dim != 128
    if pkg.device_count() == 0:
        assert L.vs_q8_create(bp, 8, 128, None, 0, 0, C.byref(h)) == -3  # no device: no runner, no CPU fallback
        with pytest.raises(pkg.VSearchError) as e:
            pkg.Q8Runner(base, 1.0, 1.0, 0, 1.0)
        assert e.value.status == -3
    else:
        bad = pkg.Q8Encodings(1.0, 1.0, 3, 1.0)  # weight_offset > 0
        assert L.vs_q8_create(bp, 8, 128, C.byref(bad), 0, 0, C.byref(h)) == -1
    assert L.vs_q8_execute(None, bp, 1, bp) == -1 and L.vs_q8_search(None, bp, 1, 5, bp, bp) == -1
    assert L.vs_q8_num_docs(None) == 0 and L.vs_q8_batch(None) == 0


def test_sweep_metrics_parser_reads_the_cli_layout():
    """scripts/sweep_ivf.py parses the metrics.txt layout that cli_ivf.cpp writes into the CSV
    columns of the reference's sweep (qidk_ivf/scripts/run_all_ivf.sh:62)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sweep_ivf", os.path.join(root, "scripts", "sweep_ivf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = ("=== IVF Search Performance Metrics ===\n\nQuery Statistics:\n  Number of queries: 100\n"
            "  Avg candidates searched: 1234.500000\n  Candidate reduction: 8.100000x\n\n"
            "Accuracy:\n  Recall@5: 97.250000%\n\n"
            "Latency:\n  Avg per query (amortized): 0.012000 ms\n  Batch P50: 0.350000 ms\n  Batch P95: 0.400000 ms\n"
            "  Batch P99: 0.500000 ms\n\nThroughput:\n  Total time: 0.001 s\n  QPS: 83333.300000\n")
    m = mod.parse_metrics(text)
    assert m == {"recall": "97.250000", "qps": "83333.300000", "avg_latency_ms": "0.012000", "p50_latency_ms": "0.350000",
                 "p95_latency_ms": "0.400000", "p99_latency_ms": "0.500000", "avg_candidates": "1234.500000",
                 "candidate_reduction": "8.100000"}
    assert mod.COLUMNS[:3] == ["dataset", "nprobe", "top_k"] and len(mod.COLUMNS) == 11


def test_index_layout_is_stable_sort_by_cluster(pkg):
    """vs_ivf_layout / vs_ivf_clamp_nlist (host side of create_ivf_model_reordered.py:92-94, :108-128) against numpy."""
    rng = np.random.default_rng(5)
    for n, nlist in ((1, 1), (50, 7), (10000, 64), (4097, 1024)):
        a = rng.integers(0, nlist, size=n).astype(np.int32)
        x = rng.standard_normal((n, 4)).astype(np.float32)
        vr, off, order = pkg.ivf_layout_from_assignment(x, a, nlist)
        ref = np.argsort(a, kind="stable")
        assert np.array_equal(order, ref) and np.array_equal(vr, x[ref])
        assert np.array_equal(off[1:], np.cumsum(np.bincount(a, minlength=nlist)))
    with pytest.raises(pkg.VSearchError):
        pkg.ivf_layout_from_assignment(np.zeros((3, 4), np.float32), np.array([0, 5, 1], dtype=np.int32), 4)
    assert pkg.clamp_nlist(1_000_000, 1024) == 1024 and pkg.clamp_nlist(10_000, 1024) == 100 and pkg.clamp_nlist(500, 256) == 16


def test_corrupt_npy_header_is_an_error_not_an_exception(pkg, tmp_path):
    """A hostile shape in a .npy header must come back as VS_ERR_IO through the C ABI (no C++ exception, no giant allocation)."""
    d = _write_index(tmp_path)
    p = os.path.join(d, "centroids.npy")
    raw = open(p, "rb").read()
    hl = int.from_bytes(raw[8:10], "little")
    hdr = raw[10:10 + hl].decode()
    bad = hdr.replace("(4, 128)", "(400000000000, 128)")
    bad = bad[:hl - 1].ljust(hl - 1) + "\n"
    open(p, "wb").write(raw[:10] + bad.encode() + raw[10 + hl:])
    with pytest.raises(pkg.VSearchError) as e:
        pkg.IVFIndex(d)
    assert e.value.status in (-2, -4)


def test_comm_argument_validation(pkg):
    L = pkg.lib()
    import ctypes as C
    out = C.c_void_p(None)
    assert L.vs_comm_create(None, 0, 1, 0, C.byref(out)) == -1
    buf = C.create_string_buffer(128)
    assert L.vs_comm_create(buf, 2, 2, 0, C.byref(out)) == -1  # rank out of range
    if pkg.device_count() == 0:
        assert L.vs_comm_create(buf, 0, 1, 0, C.byref(out)) == -3  # no device: no communicator, no fallback


def test_bf_sweep_metrics_parser_reads_the_cli_layout():
    """scripts/sweep_bf.py greps the same lines out of metrics.txt as run_all.sh:87-91 does."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import sweep_bf
    text = ("Overall Performance:\n  Total execution time: 0.010000 s\n  Throughput: 123456.500000 queries/sec\n\n"
            "GPU Execution (per batch):\n  Avg upload time: 0.001 ms\n  Avg graph execute time: 0.085000 ms\n"
            "  P50 graph exec time: 0.084 ms\n  P95 graph exec time: 0.090000 ms\n  P99 graph exec time: 0.095000 ms\n\n"
            "GPU Performance (per batch):\n  Avg GFLOPS: 96000.250000\n")
    m = sweep_bf.parse_metrics(text)
    assert m == {"throughput_qps": "123456.500000", "gflops": "96000.250000", "avg_latency_ms": "0.085000",
                 "p95_latency_ms": "0.090000", "p99_latency_ms": "0.095000"}
    assert sweep_bf.COLUMNS == ["dataset", "batch_size", "throughput_qps", "gflops", "avg_latency_ms", "p95_latency_ms", "p99_latency_ms"]
