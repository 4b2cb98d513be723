"""The CPU oracle against (a) the reference outputs recorded by the survey run of the unmodified
cpu_baseline.cpp (tests/golden/PROVENANCE.md) and (b) an independent int64 recomputation."""
import os

import numpy as np
import pytest

import oracle


def _load_ref(golden_dir, tag):
    b = oracle.read_fvecs(os.path.join(golden_dir, f"ref_ties_{tag}_base.fvecs"))
    q = oracle.read_fvecs(os.path.join(golden_dir, f"ref_ties_{tag}_query.fvecs"))
    ids, d = oracle.parse_results_txt(os.path.join(golden_dir, f"ref_ties_{tag}_results.txt"))
    return b, q, np.array(ids, dtype=np.int32), np.array(d, dtype=np.float32)


@pytest.mark.parametrize("tag", ["fwd", "rev"])
def test_tie_probe_matches_reference(golden_dir, tag):
    # select_topk's history-dependent tie order
    b, q, rid, rd = _load_ref(golden_dir, tag)
    ids, d = oracle.search_bf(b, q, 5)
    assert np.array_equal(ids, rid)
    assert np.array_equal(d, rd)
    if tag == "fwd":
        assert ids[0].tolist() == [5, 7, 2, 3, 4]
    else:
        assert ids[0].tolist() == [1, 3, 0, 2, 4]


def test_synth10k_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_synth10k_inputs.npz"))
    base, query = z["base"].astype(np.float32), z["query"].astype(np.float32)
    rid, rd = oracle.parse_results_txt(os.path.join(golden_dir, "ref_synth10k_results.txt"))
    ids, d = oracle.search_bf(base, query, 5)
    assert np.array_equal(ids, np.array(rid))
    assert np.array_equal(d, np.array(rd, dtype=np.float32))  # all < 1e6, so "%g" printed them exactly
    # independent pin: exact integer arithmetic
    ex = oracle.exact_int_dists(query, base)
    assert np.array_equal(np.sort(ex, axis=1)[:, :5].astype(np.float32), d)
    assert np.array_equal(ex[np.arange(len(ids))[:, None], ids].astype(np.float32), d)


def test_norms_and_distances_exact_on_integer_data():
    rng = np.random.default_rng(0)
    base = rng.integers(0, 219, size=(777, 128)).astype(np.float32)
    q = rng.integers(0, 219, size=(9, 128)).astype(np.float32)
    assert np.array_equal(oracle.compute_norms(base), (base.astype(np.int64) ** 2).sum(1).astype(np.float32))
    ex = oracle.exact_int_dists(q, base)
    for i in range(len(q)):
        assert np.array_equal(oracle.l2_row(q[i], base), ex[i].astype(np.float32))


def test_select_topk_slot_semantics():
    # crafted: first slot holding the max is replaced; equal distance never replaces (strict <)
    d = np.array([100, 100, 100, 100, 100, 9, 100, 9, 100], dtype=np.float32)
    idx, dd = oracle.select_topk(d, 5)
    assert idx.tolist() == [5, 7, 2, 3, 4]
    assert dd.tolist() == [9, 9, 100, 100, 100]
    # descending ids among ties are possible: slot 1 is replaced first, slot 0 later
    d = np.array([50, 60, 1, 2, 3, 7, 7], dtype=np.float32)
    idx, dd = oracle.select_topk(d, 5)
    # slots: [50,60,1,2,3] -> max slot1 <- (7,id5) -> [50,7,1,2,3] -> max slot0 <- (7,id6)
    assert idx.tolist() == [2, 3, 4, 6, 5]
    # k > N clamps
    idx, dd = oracle.select_topk(np.array([3, 1], dtype=np.float32), 4)
    assert idx.tolist() == [1, 0, -1, -1] and np.isinf(dd[2:]).all()


def test_sparse_slots_equal_dense():
    rng = np.random.default_rng(1)
    for trial in range(50):
        n = int(rng.integers(6, 400))
        d = rng.integers(0, 12, size=n).astype(np.float32)  # many ties
        k = int(rng.integers(1, 6))
        di, dd = oracle.select_topk(d, k)
        # candidate superset: the first k rows plus every row below the running k-th value of its prefix
        keep = list(range(min(k, n)))
        for j in range(k, n):
            if d[j] < np.sort(d[:j])[k - 1]:
                keep.append(j)
        keep = np.array(keep, dtype=np.int32)
        si, sd = oracle.select_topk_sparse(keep, d[keep], k)
        assert np.array_equal(di, si) and np.array_equal(dd, sd)


def test_read_fvecs_errors(tmp_path):
    p = tmp_path / "t.fvecs"
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    rec = np.concatenate([np.full((3, 1), 4, dtype=np.int32).view(np.float32), x], axis=1)
    rec.tofile(p)
    assert np.array_equal(oracle.read_fvecs(str(p)), x)
    with open(p, "ab") as f:
        f.write(b"\x04\x00")  # truncated trailing record (cpu_baseline.cpp:53-56)
    with pytest.raises(IOError):
        oracle.read_fvecs(str(p))
    with pytest.raises(IOError):
        oracle.read_fvecs(str(tmp_path / "missing.fvecs"))


def test_ivf_oracle_full_probe_equals_exact():
    rng = np.random.default_rng(2)
    base = rng.integers(0, 219, size=(3000, 128)).astype(np.float32)
    q = rng.integers(0, 219, size=(20, 128)).astype(np.float32)
    nlist = 24
    cents = base[rng.choice(len(base), nlist, replace=False)] + 0.25
    d = (base ** 2).sum(1)[:, None] - 2 * base @ cents.T + (cents ** 2).sum(1)[None]
    assign = d.argmin(1)
    order = np.argsort(assign, kind="stable").astype(np.int32)
    off = np.zeros(nlist + 1, dtype=np.int32)
    off[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    ids, dd, total = oracle.ivf_search(base[order], off, order, cents, q, 5, nlist)
    ex = oracle.exact_int_dists(q, base)
    assert total == len(q) * len(base)
    assert np.array_equal(dd, np.sort(ex, 1)[:, :5].astype(np.float32))
    assert np.array_equal(ex[np.arange(len(q))[:, None], ids].astype(np.float32), dd)
    # fewer probes: recall definition of main_ivf.cpp:52-59
    ids8, _, total8 = oracle.ivf_search(base[order], off, order, cents, q, 5, 8)
    assert total8 < total
    r = oracle.recall(ids8, ids, 5)
    assert 0.0 <= r <= 1.0
    assert oracle.recall(ids, ids, 5) == 1.0


def test_q8_quantiser_known_answers_and_numpy_recomputation():
    """quantize_buffer_neon (QnnRunner.cpp:13-55) by hand: x / 0.6627451 + 0.5, towards zero, saturated to [0, 255];
    then the whole uint8 score path against a numpy recomputation with separately rounded fp32 products and sums."""
    x = np.array([0.0, 0.33, 0.34, 1.0, -3.0, 200.0, 1e9, np.nan, 0.6627451 * 2.5, 168.8, 169.1], dtype=np.float32)
    assert oracle.q8_quantize(x, oracle.Q8_INPUT_SCALE).tolist() == [0, 0, 1, 2, 0, 255, 255, 0, 3, 255, 255]
    assert oracle.q8_quantize_weights(np.array([-2.0, 0.0, 2.0, 300.0], dtype=np.float32), 1.0, -3).tolist() == [3, 3, 5, 255]
    rng = np.random.default_rng(5)
    base = (rng.random((700, 128)) * 90).astype(np.float32)
    q = (rng.random((9, 128)) * 170).astype(np.float32)
    i_s, w_s, off, o_s = 0.6627451, 0.4, -7, 3000.0
    got = oracle.q8_scores(base, q, i_s, w_s, off, o_s)
    inv_i, inv_w = np.float32(1) / np.float32(i_s), np.float32(1) / np.float32(w_s)
    q8 = np.clip(np.trunc(q * inv_i + np.float32(0.5)), 0, 255).astype(np.int64)
    w8 = np.clip(np.clip(np.trunc(base * inv_w + np.float32(0.5)), 0, 255).astype(np.int64) - off, 0, 255)
    ip = q8 @ (w8 + off).T
    mult = (np.float32(i_s) * np.float32(w_s)) / np.float32(o_s)
    want = np.clip(np.trunc(ip.astype(np.float32) * mult + np.float32(0.5)), 0, 255).astype(np.uint8)
    assert np.array_equal(got, want) and len(np.unique(want)) > 10
    ids, top = oracle.q8_topk(got, 6)
    for b in range(len(q)):
        order = np.lexsort((np.arange(got.shape[1]), -got[b].astype(np.int64)))[:6]
        assert ids[b].tolist() == order.tolist() and top[b].tolist() == got[b][order].tolist()
    ids, top = oracle.q8_topk(got[:, :4], 6)  # fewer rows than k (IVFIndex.cpp:457 clamps likewise): (-1, 0) tail
    assert np.all(ids[:, 4:] == -1) and np.all(top[:, 4:] == 0)
