"""hai-25-rag-on-edge_amd -- MI355X (gfx950) backend for the distance + top-k hot path.

The product is ``libvsearch_hip.so`` (HIP kernels + C ABI, sources in ``csrc/``,
interface in ``include/vsearch.h``).  This Python layer is plumbing: a ctypes
binding of that ABI and thin classes that mirror the reference's C++ operator
interfaces so that tests read like the reference's own call sites:

* :class:`BruteForceIndex`  <- run_benchmark's query loop (cpu/cpu_baseline.cpp:209-254)
  and QnnRunner::executeBatchRaw (qidk_*/android/app/main/jni/QnnRunner.h:28-39)
* :class:`IVFIndex`         <- class IVFIndex (qidk_ivf/android/app/main/jni/IVFIndex.h:14-97)

There is no CPU fallback: every search call goes through the HIP library and
raises :class:`VSearchError` if it is missing or no GPU is present.

The directory name contains '-', so import it through ``__graft_entry__.load_package()``
(or ``importlib``) under the module name ``hai_25_rag_on_edge_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvsearch_hip.so")
CSRC = os.path.join(_HERE, "csrc")

VS_OK = 0
METRIC_L2 = 0
METRIC_IP = 1
MAX_BATCH = 32


class VSearchError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"vsearch error {status}: {msg}")
        self.status = status


class Timing(C.Structure):
    """vs_timing (include/vsearch.h) == IVFIndex::SearchTiming (IVFIndex.h:31-36) + extras."""
    _fields_ = [
        ("centroid_search_ms", C.c_double),
        ("gather_ms", C.c_double),
        ("fine_search_ms", C.c_double),
        ("total_ms", C.c_double),
        ("h2d_ms", C.c_double),
        ("d2h_ms", C.c_double),
        ("tie_resolve_ms", C.c_double),
        ("tie_queries", C.c_int64),
    ]


class Q8Encodings(C.Structure):
    """vs_q8_encodings: QNN scale-offset encodings, real = scale * (q + offset) (QnnRunner.cpp:490-508)."""
    _fields_ = [
        ("input_scale", C.c_float),
        ("weight_scale", C.c_float),
        ("weight_offset", C.c_int32),
        ("output_scale", C.c_float),
    ]


def hip_runtime_dir() -> str:
    """Directory of the ROCm HIP runtime the stand-alone CLIs load (tests run them as child processes)."""
    return os.environ.get("ROCM_PATH", "/opt/rocm") + "/lib"


def build(verbose: bool = False, targets=("lib",)) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, *targets]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
    process cannot both own the GPU (the second one reports "No HIP GPUs"), so when torch is
    installed its runtime is loaded first and libvsearch_hip.so binds to it through the shared
    SONAME (libamdhip64.so.7).  Stand-alone users (the C++ CLIs) use /opt/rocm's runtime.
    Set VSEARCH_SYSTEM_HIP=1 to skip this."""
    if os.environ.get("VSEARCH_SYSTEM_HIP") == "1":
        return None
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)
        return p
    return None


def lib():
    """Load libvsearch_hip.so (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VSearchError(-3, f"{LIB_PATH} not built; run __graft_entry__.build() / make -C {CSRC}")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    sig = {
        "vs_version": (C.c_char_p, []),
        "vs_last_error": (C.c_char_p, []),
        "vs_device_count": (i32, []),
        "vs_fvecs_shape": (i32, [C.c_char_p, C.POINTER(i64), C.POINTER(i32)]),
        "vs_fvecs_read": (i32, [C.c_char_p, vp, i64, C.POINTER(i64), C.POINTER(i32)]),
        "vs_ivecs_read": (i32, [C.c_char_p, vp, i64, C.POINTER(i64), C.POINTER(i32)]),
        "vs_fvecs_write": (i32, [C.c_char_p, vp, i64, i32]),
        "vs_ivecs_write": (i32, [C.c_char_p, vp, i64, i32]),
        "vs_results_write": (i32, [C.c_char_p, vp, vp, i64, i32, i32]),
        "vs_synth_sift": (i32, [vp, i64, i64, i32, C.c_uint64]),
        "vs_synth_mixture": (i32, [vp, i64, i64, i32, C.c_uint64, i32, C.c_double, C.c_double]),
        "vs_select_topk_slots": (i32, [vp, vp, i64, i32, vp, vp]),
        "vs_bf_create": (i32, [vp, i64, i32, i32, i32, i64, C.POINTER(vp)]),
        "vs_set_batch": (i32, [vp, i32]),
        "vs_set_precision": (i32, [vp, i32]),
        "vs_ivf_set_metric": (i32, [vp, i32]),
        "vs_bf_search": (i32, [vp, vp, i64, i32, vp, vp, C.POINTER(Timing)]),
        "vs_bf_search_dev": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
        "vs_bf_search_dev_multi": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
        "vs_bf_scores_dev": (i32, [vp, vp, i32, vp, i64, vp]),
        "vs_ivf_load": (i32, [C.c_char_p, i32, i32, i32, C.POINTER(vp)]),
        "vs_ivf_create": (i32, [vp, i64, i32, vp, i32, vp, vp, i32, i32, i32, C.POINTER(vp)]),
        "vs_ivf_build": (i32, [vp, i64, i32, i32, i32, C.c_double, C.c_uint64, i32, vp, vp, C.POINTER(i32)]),
        "vs_ivf_clamp_nlist": (i32, [i64, i32]),
        "vs_ivf_layout": (i32, [vp, i64, i32, vp, vp]),
        "vs_ivf_build_index": (i32, [vp, i64, i32, i32, i32, C.c_double, C.c_uint64, i32, C.POINTER(vp), C.POINTER(i32)]),
        "vs_ivf_save": (i32, [vp, C.c_char_p]),
        "vs_ivf_search": (i32, [vp, vp, i64, i32, i32, vp, vp, C.POINTER(i64), C.POINTER(Timing)]),
        "vs_ivf_search_dev": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "vs_ivf_search_dev_multi": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "vs_ivf_list_owners": (i32, [vp, i32, i32, vp]),
        "vs_topk_merge_dev": (i32, [vp, vp, i32, i32, i32, i64, i32, vp, vp, vp, vp]),
        "vs_comm_unique_id": (i32, [vp]),
        "vs_comm_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "vs_comm_rank": (i32, [vp]),
        "vs_comm_world": (i32, [vp]),
        "vs_comm_destroy": (None, [vp]),
        "vs_bf_search_dev_sharded": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
        "vs_bf_search_sharded": (i32, [vp, vp, vp, i64, i32, vp, vp, C.POINTER(Timing)]),
        "vs_ivf_search_sharded": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, C.POINTER(i64), C.POINTER(Timing)]),
        "vs_ivf_search_dev_sharded": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "vs_bf_search_vshards": (i32, [C.POINTER(vp), i32, vp, i64, i32, vp, vp, C.POINTER(Timing)]),
        "vs_ivf_shard_group": (i32, [i32]),
        "vs_ivf_shard_slice": (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "vs_ivf_shard_block_words": (i64, [i32, i32]),
        "vs_ivf_search_dev_vshards": (i32, [C.POINTER(vp), i32, vp, i32, i32, i32, i32, vp, vp, C.POINTER(C.c_double), vp]),
        "vs_q8_create": (i32, [vp, i64, i32, C.POINTER(Q8Encodings), i32, i64, C.POINTER(vp)]),
        "vs_q8_destroy": (None, [vp]),
        "vs_q8_num_docs": (i64, [vp]),
        "vs_q8_dim": (i32, [vp]),
        "vs_q8_batch": (i32, [vp]),
        "vs_q8_output_scale": (C.c_float, [vp]),
        "vs_q8_get_encodings": (i32, [vp, C.POINTER(Q8Encodings)]),
        "vs_q8_execute_dev": (i32, [vp, vp, i32, vp, i64, vp]),
        "vs_q8_execute": (i32, [vp, vp, i32, vp]),
        "vs_q8_search_dev": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "vs_q8_search": (i32, [vp, vp, i64, i32, vp, vp]),
        "vs_prof_enable": (i32, [vp, i32]),
        "vs_prof_read": (i32, [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]),
        "vs_prof_read_launches": (i32, [vp, i32, vp, i64, C.POINTER(i64)]),
        "vs_index_rows": (i64, [vp]),
        "vs_index_dim": (i32, [vp]),
        "vs_index_nlist": (i32, [vp]),
        "vs_destroy": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._vs_signatures = sig
    _lib = L
    return L


def exported_symbols():
    """Names declared in include/vsearch.h that this binding expects."""
    return sorted(lib()._vs_signatures.keys())


def _check(rc: int):
    if rc != VS_OK:
        raise VSearchError(rc, lib().vs_last_error().decode(errors="replace"))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    return int(lib().vs_device_count())


# ------------------------------------------------------------------ file formats
def read_fvecs(path: str) -> np.ndarray:
    """read_fvecs (cpu_baseline.cpp:31-58) through the C ABI."""
    L = lib()
    rows, dim = C.c_int64(0), C.c_int(0)
    _check(L.vs_fvecs_shape(path.encode(), C.byref(rows), C.byref(dim)))
    out = np.empty((rows.value, max(dim.value, 0)), dtype=np.float32)
    if out.size:
        _check(L.vs_fvecs_read(path.encode(), _p(out), out.size, C.byref(rows), C.byref(dim)))
    return out


def read_ivecs(path: str) -> np.ndarray:
    """load_ivecs (main_ivf.cpp:35-50)."""
    L = lib()
    rows, dim = C.c_int64(0), C.c_int(0)
    _check(L.vs_fvecs_shape(path.encode(), C.byref(rows), C.byref(dim)))
    out = np.empty((rows.value, max(dim.value, 0)), dtype=np.int32)
    if out.size:
        _check(L.vs_ivecs_read(path.encode(), _p(out), out.size, C.byref(rows), C.byref(dim)))
    return out


def write_fvecs(path: str, x) -> None:
    x = _f32c(x)
    _check(lib().vs_fvecs_write(path.encode(), _p(x), x.shape[0], x.shape[1]))


def write_ivecs(path: str, x) -> None:
    x = np.ascontiguousarray(x, dtype=np.int32)
    _check(lib().vs_ivecs_write(path.encode(), _p(x), x.shape[0], x.shape[1]))


def write_results(path: str, ids, dists, style: int = 0) -> None:
    """write_results (cpu_baseline.cpp:155-175) / results.txt of main_ivf.cpp:179-183 (style=1)."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    dists = _f32c(dists)
    _check(lib().vs_results_write(path.encode(), _p(ids), _p(dists), ids.shape[0], ids.shape[1], style))


def synth_sift(rows: int, seed: int, dim: int = 128, row_begin: int = 0) -> np.ndarray:
    """Deterministic SIFT-shaped synthetic rows; integer valued f32 in [0, 218]."""
    out = np.empty((rows, dim), dtype=np.float32)
    _check(lib().vs_synth_sift(_p(out), row_begin, rows, dim, seed))
    return out


def synth_mixture(rows: int, seed: int, n_centers: int, center_sigma: float, row_sigma: float, dim: int = 128, row_begin: int = 0) -> np.ndarray:
    """vs_synth_mixture: synth_sift's generator with the mixture as parameters (weakly clustered data for IVF recall tests)."""
    out = np.empty((rows, dim), dtype=np.float32)
    _check(lib().vs_synth_mixture(_p(out), row_begin, rows, dim, seed, n_centers, center_sigma, row_sigma))
    return out


# the second synthetic distribution (SIFT-range integers, weakly clustered): IVF recall well below 1 at small nprobe
WEAK_MIXTURE = dict(n_centers=65536, center_sigma=40.0, row_sigma=12.0)


def select_topk_slots(rows, dists, k: int):
    """select_topk (cpu_baseline.cpp:127-153) over a row-ordered candidate list (tie resolver)."""
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    dists = _f32c(dists)
    oi = np.empty(k, dtype=np.int32)
    od = np.empty(k, dtype=np.float32)
    _check(lib().vs_select_topk_slots(_p(rows), _p(dists), rows.shape[0], k, _p(oi), _p(od)))
    return oi, od


# ------------------------------------------------------------------ index classes
class _Index:
    def __init__(self):
        self._h = C.c_void_p(None)

    def close(self):
        if self._h:
            lib().vs_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_batch(self, batch: int):
        _check(lib().vs_set_batch(self._h, batch))

    precision_used = 0  # what set_precision() was last called with (0 = auto)

    def set_precision(self, precision: int):
        """0 = auto (int8 scan when the base is integer valued in [0, 255]), 1 = fp32, 2 = require int8.  Brute force and
        IVF alike (an IVF index then scans the fp32 rows, IVFIndex.cpp:270-358's arithmetic)."""
        _check(lib().vs_set_precision(self._h, precision))
        self.precision_used = precision

    # QnnRunner-style getters (QnnRunner.h:37-39)
    def getNumDocs(self) -> int:
        return int(lib().vs_index_rows(self._h))

    def getDim(self) -> int:
        return int(lib().vs_index_dim(self._h))

    def prof_enable(self, on: bool = True):
        _check(lib().vs_prof_enable(self._h, 1 if on else 0))

    def prof_read(self, which: int = 0):
        ms, n = C.c_double(0), C.c_int64(0)
        _check(lib().vs_prof_read(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class BruteForceIndex(_Index):
    """Exact L2 (or IP) search over a resident base: the GPU side of run_benchmark
    (cpu_baseline.cpp:177-254).  ``search`` returns exactly what the reference writes
    to results.txt: ids and squared-L2 distances, ascending, reference tie order."""

    def __init__(self, base, metric: int = METRIC_L2, device: int = 0, id_offset: int = 0):
        super().__init__()
        base = _f32c(base)
        if base.ndim != 2:
            raise ValueError("base must be [N, d]")
        self.n, self.d = base.shape
        _check(lib().vs_bf_create(_p(base), self.n, self.d, metric, device, id_offset, C.byref(self._h)))

    def search(self, queries, k: int, timing: Timing | None = None):
        q = _f32c(queries).reshape(-1, self.d)
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dists = np.empty((nq, k), dtype=np.float32)
        tm = timing if timing is not None else Timing()
        _check(lib().vs_bf_search(self._h, _p(q), nq, k, _p(ids), _p(dists), C.byref(tm)))
        return ids, dists

    def search_sharded(self, comm: "Comm", queries, k: int, timing: Timing | None = None):
        """Collective host-buffer search over row shards (vs_bf_search_sharded); ties in (dist, id) order."""
        q = _f32c(queries).reshape(-1, self.d)
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dists = np.empty((nq, k), dtype=np.float32)
        tm = timing if timing is not None else Timing()
        _check(lib().vs_bf_search_sharded(self._h, comm._c, _p(q), nq, k, _p(ids), _p(dists), C.byref(tm)))
        return ids, dists

    def search_dev(self, q_ptr: int, B: int, k: int, ids_ptr: int, dists_ptr: int, flags_ptr: int, stream: int):
        """Asynchronous device-pointer call (vs_bf_search_dev): outputs are [B, k+1]."""
        _check(lib().vs_bf_search_dev(self._h, q_ptr, B, k, ids_ptr, dists_ptr, flags_ptr, stream))

    def search_dev_multi(self, q_ptr: int, n_batches: int, B: int, k: int, ids_ptr: int, dists_ptr: int,
                         flags_ptr: int, stream: int):
        """n_batches consecutive batches of B queries, pipelined over internal streams (vs_bf_search_dev_multi)."""
        _check(lib().vs_bf_search_dev_multi(self._h, q_ptr, n_batches, B, k, ids_ptr, dists_ptr, flags_ptr, stream))

    @staticmethod
    def search_vshards(shards, queries: np.ndarray, k: int, timing: "Timing | None" = None):
        """vs_bf_search_vshards: row shards of one base on ONE device (shards[g] = BruteForceIndex(rows of g, id_offset=first row))
        -> the reference's answer, tie order and fp32 rerun included, as vs_bf_search gives on the unsharded base."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dists = np.empty((nq, k), dtype=np.float32)
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        _check(lib().vs_bf_search_vshards(arr, len(shards), q.ctypes.data, nq, k, ids.ctypes.data, dists.ctypes.data,
                                          C.byref(timing) if timing is not None else None))
        return ids, dists

    def search_dev_sharded(self, comm: "Comm", q_ptr: int, n_batches: int, B: int, k: int, ids_ptr: int, dists_ptr: int,
                           flags_ptr: int, stream: int):
        """Collective: this rank's row shard + ONE RCCL all-gather of top-(k+1) lists per launch group + device merge."""
        _check(lib().vs_bf_search_dev_sharded(self._h, comm._c, q_ptr, n_batches, B, k, ids_ptr, dists_ptr, flags_ptr, stream))

    def scores_dev(self, q_ptr: int, B: int, scores_ptr: int, ld: int, stream: int):
        """QnnRunner::executeBatchRaw analogue: raw [B, ld] score matrix on the device."""
        _check(lib().vs_bf_scores_dev(self._h, q_ptr, B, scores_ptr, ld, stream))


class Q8Runner:
    """The reference's device runner with its UFIXED_POINT_8 I/O (QnnRunner.h:18-55): the database is baked in at
    construction, ``executeBatchRaw`` returns the raw uint8 [B, N] inner-product scores, ``search`` adds
    find_top_k_int8 (main.cpp:30-57).  Method names follow QnnRunner."""

    def __init__(self, base, input_scale=None, weight_scale=None, weight_offset: int = 0, output_scale=None,
                 device: int = 0, id_offset: int = 0):
        self._h = C.c_void_p()
        base = _f32c(base)
        if base.ndim != 2:
            raise ValueError("base must be [N, d]")
        enc = None
        if input_scale is not None or weight_scale is not None or output_scale is not None:
            if input_scale is None or weight_scale is None or output_scale is None:
                raise ValueError("give all three scales or none (None = the runner's hard-coded encodings)")
            enc = C.byref(Q8Encodings(input_scale, weight_scale, weight_offset, output_scale))
        _check(lib().vs_q8_create(_p(base), base.shape[0], base.shape[1], enc, device, id_offset, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().vs_q8_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def getNumDocs(self) -> int:
        return int(lib().vs_q8_num_docs(self._h))

    def getDim(self) -> int:
        return int(lib().vs_q8_dim(self._h))

    def getBatchSize(self) -> int:
        return int(lib().vs_q8_batch(self._h))

    def getOutputScale(self) -> float:
        return float(lib().vs_q8_output_scale(self._h))

    def encodings(self) -> Q8Encodings:
        e = Q8Encodings()
        _check(lib().vs_q8_get_encodings(self._h, C.byref(e)))
        return e

    def executeBatchRaw(self, batch_queries) -> np.ndarray:
        """executeBatchRaw + getRawOutputBuffer (QnnRunner.cpp:608-645): uint8 [B, N]."""
        q = _f32c(batch_queries).reshape(-1, self.getDim())
        out = np.empty((q.shape[0], self.getNumDocs()), dtype=np.uint8)
        _check(lib().vs_q8_execute(self._h, _p(q), q.shape[0], _p(out)))
        return out

    def execute_dev(self, q_ptr: int, B: int, scores_ptr: int, ld: int, stream: int):
        _check(lib().vs_q8_execute_dev(self._h, q_ptr, B, scores_ptr, ld, stream))

    def search(self, queries, k: int):
        """The harness loop of main.cpp:201-251: ids [nq, k], uint8 scores [nq, k] (score * getOutputScale() = float)."""
        q = _f32c(queries).reshape(-1, self.getDim())
        ids = np.empty((q.shape[0], k), dtype=np.int32)
        top = np.empty((q.shape[0], k), dtype=np.uint8)
        _check(lib().vs_q8_search(self._h, _p(q), q.shape[0], k, _p(ids), _p(top)))
        return ids, top

    def search_dev(self, q_ptr: int, n_batches: int, B: int, k: int, ids_ptr: int, scores_ptr: int, stream: int):
        _check(lib().vs_q8_search_dev(self._h, q_ptr, n_batches, B, k, ids_ptr, scores_ptr, stream))


class IVFIndex(_Index):
    """class IVFIndex (IVFIndex.h:14-97) on the GPU, reordered (contiguous list) layout, L2."""

    def __init__(self, index_dir: str | None = None, device: int = 0, rank: int = 0, world: int = 1, *,
                 vectors_reordered=None, centroids=None, cluster_offsets=None, reorder_to_original=None):
        super().__init__()
        if index_dir is not None:
            _check(lib().vs_ivf_load(os.fspath(index_dir).encode(), device, rank, world, C.byref(self._h)))
        else:
            v = _f32c(vectors_reordered)
            c = _f32c(centroids)
            off = np.ascontiguousarray(cluster_offsets, dtype=np.int32)
            r2o = None if reorder_to_original is None else np.ascontiguousarray(reorder_to_original, dtype=np.int32)
            _check(lib().vs_ivf_create(_p(v), v.shape[0], v.shape[1], _p(c), c.shape[0], _p(off),
                                       _p(r2o) if r2o is not None else None, device, rank, world, C.byref(self._h)))
        self.d = self.getDim()

    @classmethod
    def build(cls, base, n_clusters: int, max_iter: int = 100, tol: float = 1e-4, seed: int = 42, device: int = 0):
        """build_ivf_index_reordered (create_ivf_model_reordered.py:82-177) entirely inside the library
        (vs_ivf_build_index); returns (index, n_iter).  `index.save(dir)` writes the reference's directory."""
        base = _f32c(base)
        self = cls.__new__(cls)
        _Index.__init__(self)
        it = C.c_int(0)
        _check(lib().vs_ivf_build_index(_p(base), base.shape[0], base.shape[1], n_clusters, max_iter, tol, seed, device,
                                        C.byref(self._h), C.byref(it)))
        self.d = self.getDim()
        return self, int(it.value)

    def getNumVectors(self) -> int:
        return self.getNumDocs()

    def getNumClusters(self) -> int:
        return int(lib().vs_index_nlist(self._h))

    def save(self, index_dir: str):
        os.makedirs(index_dir, exist_ok=True)
        _check(lib().vs_ivf_save(self._h, index_dir.encode()))

    def searchBatch(self, queries, batchSize: int, k: int, nprobe: int, timing: Timing | None = None):
        """IVFIndex::searchBatch (IVFIndex.h:45-48): returns (allIndices, allScores, totalCandidates)."""
        q = _f32c(queries).reshape(-1, self.d)[:batchSize]
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dists = np.empty((nq, k), dtype=np.float32)
        total = C.c_int64(0)
        tm = timing if timing is not None else Timing()
        _check(lib().vs_ivf_search(self._h, _p(q), nq, k, nprobe, _p(ids), _p(dists), C.byref(total), C.byref(tm)))
        return ids, dists, int(total.value)

    def searchBatch_sharded(self, comm: "Comm", queries, k: int, nprobe: int):
        """Collective host-buffer search over list shards (vs_ivf_search_sharded)."""
        q = _f32c(queries).reshape(-1, self.d)
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dists = np.empty((nq, k), dtype=np.float32)
        total = C.c_int64(0)
        tm = Timing()
        _check(lib().vs_ivf_search_sharded(self._h, comm._c, _p(q), nq, k, nprobe, _p(ids), _p(dists), C.byref(total), C.byref(tm)))
        return ids, dists, int(total.value)

    def search(self, query, k: int, nprobe: int):
        """IVFIndex::search (IVFIndex.h:25-26) for one query."""
        ids, dists, total = self.searchBatch(np.asarray(query).reshape(1, -1), 1, k, nprobe)
        return ids[0], dists[0], total

    def search_dev(self, q_ptr: int, B: int, k: int, nprobe: int, ids_ptr: int, dists_ptr: int, stream: int):
        _check(lib().vs_ivf_search_dev(self._h, q_ptr, B, k, nprobe, ids_ptr, dists_ptr, stream))

    def search_dev_multi(self, q_ptr: int, n_batches: int, B: int, k: int, nprobe: int, ids_ptr: int, dists_ptr: int,
                         stream: int):
        """n_batches independent batches [n_batches][B][128] -> [n_batches][B][k]; asynchronous on `stream`."""
        _check(lib().vs_ivf_search_dev_multi(self._h, q_ptr, n_batches, B, k, nprobe, ids_ptr, dists_ptr, stream))

    def search_dev_sharded(self, comm: "Comm", q_ptr: int, n_batches: int, B: int, k: int, nprobe: int, ids_ptr: int,
                           dists_ptr: int, stream: int):
        """Collective: this rank's lists + ONE RCCL all-gather of top-k lists per launch group + device merge."""
        _check(lib().vs_ivf_search_dev_sharded(self._h, comm._c, q_ptr, n_batches, B, k, nprobe, ids_ptr, dists_ptr, stream))

    def set_metric(self, metric: int):
        """0 = squared L2 (the north-star's), 1 = inner product (the reference's own ranking, IVFIndex.cpp:449-496)."""
        _check(lib().vs_ivf_set_metric(self._h, metric))

    @staticmethod
    def search_dev_vshards(shards, q_ptr: int, n_batches: int, B: int, k: int, nprobe: int, ids_ptr: int, dists_ptr: int,
                           stream: int, timed: bool = False):
        """Virtual ranks (vs_ivf_search_dev_vshards): `shards[r]` = IVFIndex(..., rank=r, world=len(shards)) on one device.
        Returns the per-rank device time in ms (front + back halves) when `timed`."""
        G = len(shards)
        arr = (C.c_void_p * G)(*[s._h for s in shards])
        ms = (C.c_double * G)() if timed else None
        _check(lib().vs_ivf_search_dev_vshards(arr, G, q_ptr, n_batches, B, k, nprobe, ids_ptr, dists_ptr, ms, stream))
        return list(ms) if timed else None


class Comm:
    """vs_comm: an RCCL communicator owned by the library (one process per GPU).  rank 0: ``uid = Comm.unique_id()``,
    ship the bytes to the other ranks, then every rank: ``Comm(uid, rank, world, device)``."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(lib().vs_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        self._c = C.c_void_p(None)
        if len(unique_id) != Comm.ID_BYTES:
            raise ValueError("unique_id must be VS_COMM_ID_BYTES bytes")
        _check(lib().vs_comm_create(C.create_string_buffer(unique_id, Comm.ID_BYTES), rank, world, device, C.byref(self._c)))
        self.rank, self.world = rank, world

    def close(self):
        if self._c:
            lib().vs_comm_destroy(self._c)
            self._c = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def topk_merge_dev(dists_ptr: int, ids_ptr: int, G: int, B: int, kin: int, kout: int, out_d_ptr: int,
                   out_i_ptr: int, flags_ptr: int, stream: int, stride_g: int = 0):
    """Merge G per-shard sorted lists (e.g. an RCCL all-gather receive buffer) on the device."""
    _check(lib().vs_topk_merge_dev(dists_ptr, ids_ptr, G, B, kin, stride_g, kout, out_d_ptr, out_i_ptr,
                                   flags_ptr, stream))


# ------------------------------------------------------------------ IVF index building (host logic)
def ivf_layout_from_assignment(vectors: np.ndarray, cluster_ids: np.ndarray, n_clusters: int):
    """Reordered index layout of create_ivf_model_reordered.py:108-128 through the library (vs_ivf_layout: stable sort
    by cluster, offsets = running sum of sizes).  Returns (vectors_reordered, cluster_offsets, reorder_to_original)."""
    a = np.ascontiguousarray(cluster_ids, dtype=np.int32)
    offsets = np.empty(n_clusters + 1, dtype=np.int32)
    order = np.empty(a.shape[0], dtype=np.int32)
    _check(lib().vs_ivf_layout(_p(a), a.shape[0], n_clusters, _p(offsets), _p(order)))
    return np.ascontiguousarray(vectors[order], dtype=np.float32), offsets, order


def ivf_build(base, n_clusters: int, max_iter: int = 100, tol: float = 1e-4, seed: int = 42, device: int = 0):
    """build_ivf_index_reordered (create_ivf_model_reordered.py:82-177) on the GPU: k-means (vs_ivf_build),
    then the reordered layout.  Returns (vectors_reordered, cluster_offsets, reorder_to_original, centroids, n_iter)."""
    base = _f32c(base)
    n, d = base.shape
    n_clusters = clamp_nlist(n, n_clusters)
    cents = np.empty((n_clusters, d), dtype=np.float32)
    assign = np.empty(n, dtype=np.int32)
    it = C.c_int(0)
    _check(lib().vs_ivf_build(_p(base), n, d, n_clusters, max_iter, tol, seed, device, _p(cents), _p(assign), C.byref(it)))
    vr, off, r2o = ivf_layout_from_assignment(base, assign, n_clusters)
    return vr, off, r2o, cents, int(it.value)


def clamp_nlist(n_vectors: int, n_clusters: int) -> int:
    """nlist clamp of create_ivf_model_reordered.py:92-94 (vs_ivf_clamp_nlist)."""
    return int(lib().vs_ivf_clamp_nlist(n_vectors, n_clusters))


# ------------------------------------------------------------------ multi-GPU host logic (no GPU needed)
def row_shard_bounds(n_rows: int, world: int) -> np.ndarray:
    """Contiguous row shards for brute force; interior bounds are multiples of 16 (one MFMA tile)."""
    b = np.linspace(0, n_rows, world + 1).astype(np.int64)
    b[1:-1] = (b[1:-1] // 16) * 16
    return b


def ivf_list_owners(cluster_offsets, world: int) -> np.ndarray:
    """Rank owning each inverted list (longest-first round robin) -- the library's own assignment."""
    off = np.ascontiguousarray(cluster_offsets, dtype=np.int32)
    out = np.empty(len(off) - 1, dtype=np.int32)
    _check(lib().vs_ivf_list_owners(_p(off), len(off) - 1, world, _p(out)))
    return out


class ProbeBlockLayout:
    """The block a rank contributes to the exchange between the two halves of the cluster-sharded IVF pipeline
    (vs_ivf_search_dev_sharded): for the `sbb * 32` query slots of its slice (batches padded to 32 queries),
    probes [slots][nprobe] int32 | bounds [slots] f32 bits | slow marks [slots] int32."""

    def __init__(self, n_batches: int, world: int, nprobe: int):
        self.world, self.nprobe = world, nprobe
        sbb, b0, nbs = C.c_int(), C.c_int(), C.c_int()
        _check(lib().vs_ivf_shard_slice(n_batches, world, 0, C.byref(sbb), C.byref(b0), C.byref(nbs)))
        self.sbb = sbb.value
        self.slots = self.sbb * 32
        self.words = int(lib().vs_ivf_shard_block_words(self.sbb, nprobe))
        self.tau_offset = self.slots * nprobe
        self.slow_offset = self.tau_offset + self.slots
        self.n_batches = n_batches

    def slice_of(self, rank: int):
        """(first batch, batches) of `rank`'s own slice."""
        sbb, b0, nbs = C.c_int(), C.c_int(), C.c_int()
        _check(lib().vs_ivf_shard_slice(self.n_batches, self.world, rank, C.byref(sbb), C.byref(b0), C.byref(nbs)))
        return b0.value, nbs.value

    def slot(self, batch: int, b: int):
        """(slice, slot in the slice's block) of query b of batch `batch` of the group."""
        return batch // self.sbb, (batch % self.sbb) * 32 + b


class GatherLayout:
    """The per-rank buffer that goes through the all-gather: [2][S][B][K] 32-bit words, plane 0 = distances
    (f32 bits), plane 1 = ids.  After an all-gather into [world][2][S][B][K] the merge kernel reads shard g's
    list of query (s, b) at g*stride_g + (s*B + b)*K (+ ids_offset words for the ids)."""

    def __init__(self, S: int, B: int, K: int):
        self.S, self.B, self.K = S, B, K
        self.words = 2 * S * B * K
        self.stride_g = 2 * S * B * K
        self.ids_offset = S * B * K

    def dist_offset(self, s: int) -> int:
        return s * self.B * self.K

    def id_offset(self, s: int) -> int:
        return self.ids_offset + s * self.B * self.K

    def merge_reference(self, gathered: np.ndarray, kout: int):
        """numpy restatement of vs_topk_merge_dev on a gathered int32 buffer [world, words]."""
        world = gathered.shape[0]
        g = gathered.reshape(world, 2, self.S * self.B, self.K)
        d = g[:, 0].view(np.float32)
        i = g[:, 1]
        nq = self.S * self.B
        out_d = np.full((nq, kout), np.inf, dtype=np.float32)
        out_i = np.full((nq, kout), -1, dtype=np.int32)
        for q in range(nq):
            cand = [(float(d[w, q, j]), int(i[w, q, j])) for w in range(world) for j in range(self.K)
                    if i[w, q, j] >= 0 and np.isfinite(d[w, q, j])]
            cand.sort()
            for t, (dd, ii) in enumerate(cand[:kout]):
                out_d[q, t], out_i[q, t] = dd, ii
        return out_d, out_i
