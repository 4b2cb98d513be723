"""CPU oracle (test infrastructure only).

ctypes binding of oracle/libvs_oracle.so (built from oracle/vs_oracle.c by
``make -C oracle``) plus an independent exact-integer recomputation in numpy.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package; the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvs_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "libvs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        # a GPU box may expose far more cores than its CPU share: do not oversubscribe OpenMP
        if "OMP_NUM_THREADS" not in os.environ:
            try:
                n = len(os.sched_getaffinity(0))
            except AttributeError:
                n = os.cpu_count() or 1
            os.environ["OMP_NUM_THREADS"] = str(max(1, min(n, 16)))
        build()
        L = C.CDLL(_LIB_PATH)
        L.vo_read_fvecs.restype = C.c_int
        L.vo_read_fvecs.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.vo_compute_norms.restype = None
        L.vo_compute_norms.argtypes = [_f32p, C.c_int64, C.c_int, _f32p]
        L.vo_l2_row.restype = None
        L.vo_l2_row.argtypes = [_f32p, C.c_float, _f32p, _f32p, C.c_int64, C.c_int, _f32p]
        L.vo_select_topk.restype = None
        L.vo_select_topk.argtypes = [_f32p, C.c_int64, C.c_int, _i32p, _f32p]
        L.vo_select_topk_sparse.restype = None
        L.vo_select_topk_sparse.argtypes = [_i32p, _f32p, C.c_int64, C.c_int, _i32p, _f32p]
        L.vo_search_bf.restype = C.c_int
        L.vo_search_bf.argtypes = [_f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_int, _i32p, _f32p,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.vo_write_results.restype = C.c_int
        L.vo_write_results.argtypes = [C.c_char_p, _i32p, _f32p, C.c_int64, C.c_int]
        L.vo_ivf_search.restype = C.c_int64
        L.vo_ivf_search.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, _i32p, C.c_void_p,
                                    _f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p, C.c_void_p]
        L.vo_ivf_search_metric.restype = C.c_int64
        L.vo_ivf_search_metric.argtypes = [_f32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, _i32p, C.c_void_p,
                                           _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _f32p, C.c_void_p]
        L.vo_recall.restype = C.c_double
        L.vo_recall.argtypes = [_i32p, C.c_int, _i32p, C.c_int, C.c_int]
        L.vo_num_threads.restype = C.c_int
        L.vo_q8_quantize.restype = None
        L.vo_q8_quantize.argtypes = [_f32p, C.c_void_p, C.c_int64, C.c_float]
        L.vo_q8_quantize_weights.restype = None
        L.vo_q8_quantize_weights.argtypes = [_f32p, C.c_void_p, C.c_int64, C.c_float, C.c_int]
        L.vo_q8_scores.restype = None
        L.vo_q8_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.vo_q8_topk.restype = C.c_int
        L.vo_q8_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int, _i32p, C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def read_fvecs(path: str) -> np.ndarray:
    """cpu_baseline.cpp:31-58 via the C restatement (raises like the reference fails)."""
    L = lib()
    rows, dim = C.c_int64(0), C.c_int(0)
    rc = L.vo_read_fvecs(path.encode(), None, 0, C.byref(rows), C.byref(dim))
    if rc != 0:
        raise IOError(f"vo_read_fvecs({path}) failed rc={rc}")
    out = np.empty((rows.value, dim.value), dtype=np.float32)
    rc = L.vo_read_fvecs(path.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(rows), C.byref(dim))
    if rc != 0:
        raise IOError(f"vo_read_fvecs({path}) failed rc={rc}")
    return out


def compute_norms(x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().vo_compute_norms(x, x.shape[0], x.shape[1], out)
    return out


def l2_row(q: np.ndarray, base: np.ndarray, bn: np.ndarray | None = None) -> np.ndarray:
    q = _f32(q).reshape(1, -1)
    base = _f32(base)
    if bn is None:
        bn = compute_norms(base)
    qn = compute_norms(q)[0]
    out = np.empty(base.shape[0], dtype=np.float32)
    lib().vo_l2_row(q[0], float(qn), base, _f32(bn), base.shape[0], base.shape[1], out)
    return out


def select_topk(dist: np.ndarray, k: int):
    dist = _f32(dist)
    idx = np.empty(k, dtype=np.int32)
    dd = np.empty(k, dtype=np.float32)
    lib().vo_select_topk(dist, dist.shape[0], k, idx, dd)
    return idx, dd


def select_topk_sparse(rows: np.ndarray, dist: np.ndarray, k: int):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    dist = _f32(dist)
    idx = np.empty(k, dtype=np.int32)
    dd = np.empty(k, dtype=np.float32)
    lib().vo_select_topk_sparse(rows, dist, rows.shape[0], k, idx, dd)
    return idx, dd


def search_bf(base: np.ndarray, queries: np.ndarray, k: int, timing: dict | None = None):
    """cpu_baseline.cpp run_benchmark loop: returns (ids[nq,k], dists[nq,k])."""
    base, queries = _f32(base), _f32(queries)
    nq = queries.shape[0]
    idx = np.empty((nq, k), dtype=np.int32)
    dd = np.empty((nq, k), dtype=np.float32)
    td, tk = C.c_double(0), C.c_double(0)
    rc = lib().vo_search_bf(base, base.shape[0], base.shape[1], queries, nq, k, idx, dd, C.byref(td), C.byref(tk))
    if rc != 0:
        raise MemoryError("vo_search_bf")
    if timing is not None:
        timing["dist_s"], timing["topk_s"] = td.value, tk.value
    return idx, dd


def write_results(path: str, idx: np.ndarray, dist: np.ndarray) -> None:
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    rc = lib().vo_write_results(path.encode(), idx, _f32(dist), idx.shape[0], idx.shape[1])
    if rc != 0:
        raise IOError(path)


def ivf_search(vectors_reordered, offsets, reorder_to_original, centroids, queries, k, nprobe, return_probes=False, metric=0):
    """IVFIndex::searchBatch (reordered mode) restated with L2 (metric 0) or the reference's own inner product (metric 1:
    dists = -q.v, smallest first).  Returns ids, dists, total_candidates[, probes]."""
    v = _f32(vectors_reordered)
    cen = _f32(centroids)
    q = _f32(queries)
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    vn = compute_norms(v)
    nq = q.shape[0]
    nlist = cen.shape[0]
    npb = min(nprobe, nlist)
    idx = np.empty((nq, k), dtype=np.int32)
    dd = np.empty((nq, k), dtype=np.float32)
    r2o = None
    r2o_p = None
    if reorder_to_original is not None:
        r2o = np.ascontiguousarray(reorder_to_original, dtype=np.int32)
        r2o_p = r2o.ctypes.data_as(C.c_void_p)
    probes = np.empty((nq, npb), dtype=np.int32) if return_probes else None
    total = lib().vo_ivf_search_metric(v, vn, v.shape[0], v.shape[1], cen, nlist, off, r2o_p, q, nq, k, nprobe, int(metric), idx, dd,
                                       probes.ctypes.data_as(C.c_void_p) if probes is not None else None)
    if return_probes:
        return idx, dd, int(total), probes
    return idx, dd, int(total)


def recall(pred: np.ndarray, gt: np.ndarray, k: int) -> float:
    """Mean of main_ivf.cpp:52-59 compute_recall over queries."""
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    gt = np.ascontiguousarray(gt, dtype=np.int32)
    L = lib()
    return float(np.mean([L.vo_recall(pred[i], pred.shape[1], gt[i], gt.shape[1], k) for i in range(pred.shape[0])]))


def num_threads() -> int:
    return int(lib().vo_num_threads())


# ---------------------------------------------------------------------------
# UFIXED_POINT_8 score path (QnnRunner.cpp:13-55, 490-521, 608-645; main.cpp:30-57).
# Parity unpinned: the NPU graph between quantiser and top-k is not in the reference.
# ---------------------------------------------------------------------------
Q8_INPUT_SCALE = np.float32(0.6627451181411743)      # QnnRunner.cpp:490
Q8_OUTPUT_SCALE = np.float32(1013.4312133789062500)  # QnnRunner.cpp:507


def q8_quantize(x: np.ndarray, scale: float) -> np.ndarray:
    """quantize_buffer_neon (QnnRunner.cpp:13-55) with inv_scale = 1.0f / scale (:619)."""
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint8)
    inv = np.float32(1.0) / np.float32(scale)
    lib().vo_q8_quantize(x.reshape(-1), out.ctypes.data, x.size, inv)
    return out


def q8_quantize_weights(x: np.ndarray, scale: float, offset: int = 0) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint8)
    inv = np.float32(1.0) / np.float32(scale)
    lib().vo_q8_quantize_weights(x.reshape(-1), out.ctypes.data, x.size, inv, int(offset))
    return out


def q8_mult(input_scale, weight_scale, output_scale) -> np.float32:
    return (np.float32(input_scale) * np.float32(weight_scale)) / np.float32(output_scale)


def q8_scores(base: np.ndarray, queries: np.ndarray, input_scale, weight_scale, weight_offset, output_scale) -> np.ndarray:
    """uint8 [B x N] score matrix of executeBatchRaw (QnnRunner.cpp:608-645) as restated in vs_oracle.c."""
    w8 = q8_quantize_weights(base, weight_scale, weight_offset)
    q8 = q8_quantize(queries, input_scale)
    B, d = q8.shape
    out = np.empty((B, w8.shape[0]), dtype=np.uint8)
    lib().vo_q8_scores(q8.ctypes.data, w8.ctypes.data, B, w8.shape[0], d, int(weight_offset),
                       q8_mult(input_scale, weight_scale, output_scale), out.ctypes.data)
    return out


def q8_topk(scores: np.ndarray, k: int):
    """find_top_k_int8 (main.cpp:36-57) per row of a uint8 score matrix; equal scores in ascending id order."""
    scores = np.ascontiguousarray(scores, dtype=np.uint8)
    B, n = scores.shape
    ids = np.empty((B, k), dtype=np.int32)
    top = np.empty((B, k), dtype=np.uint8)
    L = lib()
    for b in range(B):
        L.vo_q8_topk(scores[b].ctypes.data, n, k, ids[b], top[b].ctypes.data)
    return ids, top


# ---------------------------------------------------------------------------
# Independent exact recomputation (numpy int64).  On integer-valued data the
# squared L2 distance is an integer < 2^24 and therefore the unique value any
# correct fp32 evaluation of ||q||^2 + ||b||^2 - 2 q.b must produce
# (SURVEY.md 0.1-4).  Used to pin the C restatement's distances independently
# of its own summation order.
# ---------------------------------------------------------------------------
def exact_int_dists(queries: np.ndarray, base: np.ndarray) -> np.ndarray:
    q = np.asarray(queries)
    b = np.asarray(base)
    if not (np.all(q == np.rint(q)) and np.all(b == np.rint(b))):
        raise ValueError("exact_int_dists needs integer-valued inputs")
    qi = q.astype(np.int64)
    bi = b.astype(np.int64)
    qn = (qi * qi).sum(1)
    bn = (bi * bi).sum(1)
    d = qn[:, None] + bn[None, :] - 2 * (qi @ bi.T)
    if d.max() >= 1 << 24:
        raise ValueError("distance exceeds 2^24: fp32 no longer exact")
    return d


def parse_results_txt(path: str):
    """Parse the reference's results grammar (cpu_baseline.cpp:167-173)."""
    import re
    ids, dists = [], []
    pat = re.compile(r"\((-?\d+), ([^)]+)\)")
    with open(path) as f:
        for line in f:
            if not line.startswith("Query"):
                continue
            pairs = pat.findall(line)
            ids.append([int(a) for a, _ in pairs])
            dists.append([float(b) for _, b in pairs])
    return ids, dists
