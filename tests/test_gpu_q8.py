"""GPU parity of the UFIXED_POINT_8 score path (second half) against the oracle's restatement of
QnnRunner.cpp:13-55 / 490-521 / 608-645 and main.cpp:30-57.  Parity unpinned (the NPU graph between the quantiser and
the top-k is not in the reference, and the reference holds no recorded scores): bit-exact against oracle.q8_*."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _calibrated(base, queries):
    """Min-max encodings the way a converter run with calibration inputs would choose them."""
    in_scale = float(queries.max()) / 255.0
    w_scale = float(base.max()) / 255.0
    ip_max = float((queries[:8].astype(np.float64) @ base[:: max(1, len(base) // 4096)].astype(np.float64).T).max())
    return in_scale, w_scale, 1.05 * ip_max / 255.0


@pytest.mark.parametrize("n,B", [(20000, 32), (20037, 17), (16384, 16), (1, 1), (63, 5), (65, 32), (50001, 31)])
def test_score_matrix_bit_exact(gpu_pkg, n, B):
    base = gpu_pkg.synth_sift(n, seed=11)
    q = gpu_pkg.synth_sift(B, seed=12)
    i_s, w_s, o_s = _calibrated(base, q)
    with gpu_pkg.Q8Runner(base, i_s, w_s, 0, o_s) as r:
        assert (r.getNumDocs(), r.getDim(), r.getBatchSize()) == (n, 128, 32)
        assert r.getOutputScale() == np.float32(o_s)
        got = r.executeBatchRaw(q)
    want = oracle.q8_scores(base, q, i_s, w_s, 0, o_s)
    assert got.dtype == np.uint8 and got.shape == (B, n)
    assert np.array_equal(got, want)
    if n >= 10000:
        assert want.max() > 128 and len(np.unique(want)) > 16  # the encodings use the range: not a trivial comparison


def test_reference_constants_and_weight_offset(gpu_pkg):
    rng = np.random.default_rng(3)
    # unit-norm non-negative vectors scaled into the runner's hard-coded input range (0.6627451 * 255 = 169)
    base = np.abs(rng.standard_normal((30000, 128))).astype(np.float32)
    base *= 120.0 / np.linalg.norm(base, axis=1, keepdims=True)
    q = np.abs(rng.standard_normal((32, 128))).astype(np.float32) * 60.0
    q[0, :4] = [-5.0, np.nan, 1e9, 0.33]  # saturation / NaN lanes of the quantiser (QnnRunner.cpp:50-54)
    with gpu_pkg.Q8Runner(base) as r:  # enc = NULL: QnnRunner.cpp:490-521 scales, weights min-max
        e = r.encodings()
        assert e.input_scale == oracle.Q8_INPUT_SCALE and e.output_scale == oracle.Q8_OUTPUT_SCALE
        assert e.weight_scale == np.float32(base.max()) / np.float32(255.0) and e.weight_offset == 0
        got = r.executeBatchRaw(q)
    want = oracle.q8_scores(base, q, e.input_scale, e.weight_scale, 0, e.output_scale)
    assert np.array_equal(got, want)
    # asymmetric weights: real = scale * (q + offset), offset < 0
    shifted = base - 3.0
    w_s = float(shifted.max() - shifted.min()) / 255.0
    off = int(round(float(shifted.min()) / w_s))
    assert off < 0
    with gpu_pkg.Q8Runner(shifted, 0.5, w_s, off, 400.0) as r:
        got = r.executeBatchRaw(q[:9])
    want = oracle.q8_scores(shifted, q[:9], 0.5, w_s, off, 400.0)
    assert np.array_equal(got, want)
    assert len(np.unique(want)) > 8


@pytest.mark.parametrize("k", [1, 5, 16])
def test_topk_over_uint8_scores(gpu_pkg, k):
    base = gpu_pkg.synth_sift(70000, seed=21)  # 5 top-k chunks, the last one short
    q = gpu_pkg.synth_sift(75, seed=22)        # 2 full batches + one of 11
    i_s, w_s, o_s = _calibrated(base, q)
    with gpu_pkg.Q8Runner(base, i_s, w_s, 0, o_s, id_offset=1000) as r:
        ids, top = r.search(q, k)
        raw = np.concatenate([r.executeBatchRaw(q[i:i + 32]) for i in range(0, 75, 32)])
    oid, otop = oracle.q8_topk(raw, k)
    assert np.array_equal(top, otop)
    assert np.array_equal(ids, oid + 1000)
    # 256 score levels over 70 000 rows: equal scores at the cut are the rule, and they come out in ascending id order
    assert any(len(np.unique(t)) < k for t in top) or k == 1


def test_topk_saturated_ties_and_tiny_database(gpu_pkg):
    base = gpu_pkg.synth_sift(40000, seed=31)
    q = gpu_pkg.synth_sift(3, seed=32)
    with gpu_pkg.Q8Runner(base, 1.0, 1.0, 0, 50.0) as r:  # every score saturates at 255
        ids, top = r.search(q, 5)
    assert np.all(top == 255) and np.array_equal(ids, np.tile(np.arange(5, dtype=np.int32), (3, 1)))
    with gpu_pkg.Q8Runner(base, 1.0, 1.0, 0, 1e9) as r:   # every score is 0
        ids, top = r.search(q, 5)
    assert np.all(top == 0) and np.array_equal(ids, np.tile(np.arange(5, dtype=np.int32), (3, 1)))
    small = base[:3]
    with gpu_pkg.Q8Runner(small, 1.0, 1.0, 0, 12000.0) as r:  # fewer rows than k: the tail is (-1, 0)
        ids, top = r.search(q, 5)
        raw = r.executeBatchRaw(q)
    oid, otop = oracle.q8_topk(raw, 5)
    assert np.array_equal(ids, oid) and np.array_equal(top, otop) and np.all(ids[:, 3:] == -1)
    with pytest.raises(gpu_pkg.VSearchError) as e:
        gpu_pkg.Q8Runner(base, 1.0, 1.0, 0, 1.0).search(q, 17)
    assert e.value.status == -5


def test_device_pointer_calls_and_unaligned_leading_dimension(gpu_pkg):
    import torch

    base = gpu_pkg.synth_sift(10007, seed=41)
    q = gpu_pkg.synth_sift(64, seed=42)
    i_s, w_s, o_s = _calibrated(base, q)
    want = oracle.q8_scores(base, q, i_s, w_s, 0, o_s)
    qd = torch.from_numpy(q).cuda()
    st = torch.cuda.current_stream().cuda_stream
    with gpu_pkg.Q8Runner(base, i_s, w_s, 0, o_s) as r:
        for ld in (10007, 10016, 10240):  # byte-store path, 16-byte path with a ragged tail, padded rows
            out = torch.full((32 * ld + 16,), 7, dtype=torch.uint8, device="cuda")
            r.execute_dev(qd.data_ptr(), 32, out.data_ptr(), ld, st)
            torch.cuda.synchronize()
            got = out[: 32 * ld].view(32, ld).cpu().numpy()
            assert np.array_equal(got[:, :10007], want[:32])
            assert np.all(got[:, 10007:] == 7) and np.all(out[32 * ld:].cpu().numpy() == 7)  # nothing past a row's end
        ids = torch.empty((64, 5), dtype=torch.int32, device="cuda")
        top = torch.empty((64, 5), dtype=torch.uint8, device="cuda")
        r.search_dev(qd.data_ptr(), 2, 32, 5, ids.data_ptr(), top.data_ptr(), st)
        torch.cuda.synchronize()
    oid, otop = oracle.q8_topk(want, 5)
    assert np.array_equal(ids.cpu().numpy(), oid) and np.array_equal(top.cpu().numpy(), otop)


def test_sift1m_score_matrix_and_topk(gpu_pkg):
    """BASELINE's full size: 1 M rows x 32 queries, bit for bit against the oracle (4 G integer MACs on the CPU)."""
    base = gpu_pkg.synth_sift(1_000_000, seed=20251205)
    q = gpu_pkg.synth_sift(32, seed=20251206)
    i_s, w_s, o_s = _calibrated(base, q)
    with gpu_pkg.Q8Runner(base, i_s, w_s, 0, o_s) as r:
        got = r.executeBatchRaw(q)
        ids, top = r.search(q, 5)
    want = oracle.q8_scores(base, q, i_s, w_s, 0, o_s)
    assert np.array_equal(got, want)
    oid, otop = oracle.q8_topk(want, 5)
    assert np.array_equal(ids, oid) and np.array_equal(top, otop)


def test_cli_q8_mode_writes_the_qidk_harness_files(gpu_pkg, tmp_path):
    """`vsearch_bf <ctx> <queries> <results_dir> <backend> <documents> <k> [batch] --q8=...`: the qidk_bruteforce harness
    (main.cpp:196-251) through the quantised runner -- results.txt holds (id, score8 * output_scale) with 4 decimals
    (main.cpp:244-246), equal to the oracle's restatement."""
    import os
    import subprocess

    base = gpu_pkg.synth_sift(30000, seed=51)
    q = gpu_pkg.synth_sift(70, seed=52)
    i_s, w_s, o_s = (np.float32(x) for x in _calibrated(base, q))
    gpu_pkg.write_fvecs(str(tmp_path / "docs.fvecs"), base)
    gpu_pkg.write_fvecs(str(tmp_path / "queries.fvecs"), q)
    exe = os.path.join(os.path.dirname(gpu_pkg.LIB_PATH), "vsearch_bf")
    assert os.path.exists(exe), "vsearch_bf not built (make -C hai-25-rag-on-edge_amd/csrc all)"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = gpu_pkg.hip_runtime_dir() + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    enc = f"--q8={float(i_s)!r},{float(w_s)!r},0,{float(o_s)!r}"
    r = subprocess.run([exe, "ctx.bin", "queries.fvecs", "out", "libQnnHtp.so", "docs.fvecs", "5", "32", enc], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = oracle.q8_scores(base, q, i_s, w_s, 0, o_s)
    oid, otop = oracle.q8_topk(raw, 5)
    ids, scores = oracle.parse_results_txt(str(tmp_path / "out" / "results.txt"))
    assert np.array_equal(np.array(ids), oid)
    want = np.array([[float(f"{np.float32(t) * o_s:.4f}") for t in row] for row in otop])
    assert np.array_equal(np.array(scores), want)
    m = open(tmp_path / "out" / "metrics.txt").read()
    for needle in ("UFIXED_POINT_8", "Number of queries: 70", "Number of documents: 30000", "Batch size: 32", "Number of batches: 3",
                   "Top-K: 5", "Output scale:", "Operational Intensity:", "Throughput:", "queries/sec"):
        assert needle in m, needle
    # unparsable encodings are refused before any work
    r = subprocess.run([exe, "ctx.bin", "queries.fvecs", "out", "libQnnHtp.so", "docs.fvecs", "5", "--q8=1,2"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "--q8=" in r.stderr
