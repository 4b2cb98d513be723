#!/usr/bin/env python3
"""bench.py -- QPS of the distance + top-k hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one pass of the hot path over one batch of 32 synthetic queries that are already
resident in HBM: exact brute force, k = 5, against the synthetic SIFT-1M base (1 000 000 x 128 fp32,
generator of SURVEY.md 8d, seeds 20251205 / 20251206).  `value` = queries / second of the whole job.

N = 1  : config "SIFT-1M brute-force batch=32 on 1xMI355X" (BASELINE.json configs[2]).
N > 1  : the same base row-sharded over the ranks (strong scaling); every rank scans its shard for
         the same 32 queries, the per-shard top-(k+1) lists are exchanged with ONE RCCL all-gather per
         `--coll-every` steps and merged on the device (the only data-path collective the path has).
Extra (same JSON line, key "ivf"): IVF nlist=1024 nprobe=32 QPS + recall@1 on the same data, list
shards dealt over the ranks when N > 1 (BASELINE.json configs[3]/[4]).

The JSON line also carries `roofline` (dominant kernel vs the HBM roof, timed with HIP events on the
stream it runs on) and `cpu_baseline` (the oracle's restatement of cpu_baseline.cpp on this box's
host cores, bounded sample, rank 0 at N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BASE = 1_000_000
DIM = 128
BATCH = 32
K = 5
NLIST = 1024
NPROBE = 32
SEED_BASE, SEED_QUERY = 20251205, 20251206
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def kmeans_assign(x, cents, chunk=131072):
    """argmin_c ||x - c||^2 on the GPU with torch (index BUILD only -- not the graded path)."""
    import torch
    cn = (cents * cents).sum(1)
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        d = cn[None, :] - 2.0 * (xs @ cents.T)
        out[s:s + chunk] = d.argmin(1)
    return out


def build_ivf_torch(base_dev, nlist, iters, seed):
    """Lloyd k-means (create_ivf_model_reordered.py:96-105 uses sklearn KMeans; rebuilt here with
    torch because the reference builder does not parse and sklearn is not the point of the bench)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = base_dev.shape[0]
    cents = base_dev[torch.randperm(n, generator=g)[:nlist].to(base_dev.device)].clone()
    for _ in range(iters):
        a = kmeans_assign(base_dev, cents)
        sums = torch.zeros_like(cents).index_add_(0, a, base_dev)
        cnt = torch.bincount(a, minlength=nlist).to(cents.dtype)
        nz = cnt > 0
        cents[nz] = sums[nz] / cnt[nz, None]
    a = kmeans_assign(base_dev, cents)
    return cents, a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--coll-every", type=int, default=32,
                    help="batches per multi-batch call (and per all-gather + merge when N > 1)")
    ap.add_argument("--rows", type=int, default=N_BASE, help="base rows (default SIFT-1M)")
    ap.add_argument("--no-ivf", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not time the scan kernel with HIP events")
    ap.add_argument("--no-int8", action="store_true")
    ap.add_argument("--kmeans-iters", type=int, default=20)
    ap.add_argument("--torch-kmeans", action="store_true", help="build the index with torch instead of vs_ivf_build")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N > 1 path on ONE GPU (all ranks share cuda:0, gathers staged through host)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if pkg.device_count() < 1:
        raise SystemExit("no HIP device: the bench never runs a CPU fallback")

    n_rows = args.rows
    steps, warmup, S = args.steps, args.warmup, max(1, args.coll_every)
    K1 = K + 1

    # ---------------------------------------------------------------- data (synthetic, deterministic)
    t0 = time.time()
    bounds = pkg.row_shard_bounds(n_rows, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    shard = pkg.synth_sift(r1 - r0, seed=SEED_BASE, row_begin=r0)
    n_queries = 4096
    queries = pkg.synth_sift(n_queries, seed=SEED_QUERY)
    log(f"generated shard rows [{r0},{r1}) + {n_queries} queries in {time.time() - t0:.1f}s")

    bf = pkg.BruteForceIndex(shard, device=local_rank, id_offset=r0)
    bf.set_precision(1)  # the graded config is fp32 rows + fp32 MFMA (cblas_sgemm arithmetic); int8 is reported as an extra
    q_dev = torch.from_numpy(queries).to(dev)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    # per-rank result buffer: [2][S][B][K1] words (dists as float bits | ids) -> one all-gather per S steps
    lay = pkg.GatherLayout(S, BATCH, K1)
    loc = torch.zeros((lay.words,), dtype=torch.int32, device=dev)
    loc_d_ptr = loc.data_ptr()
    loc_i_ptr = loc.data_ptr() + lay.ids_offset * 4
    gath = torch.zeros((world * lay.words,), dtype=torch.int32, device=dev) if world > 1 else None
    out_d = torch.zeros((S * BATCH, K1), dtype=torch.float32, device=dev)
    out_i = torch.zeros((S * BATCH, K1), dtype=torch.int32, device=dev)
    flags = torch.zeros((S * BATCH,), dtype=torch.int32, device=dev)
    n_qbatches = n_queries // BATCH

    def bf_step(i, n):
        # steps are issued in groups of S batches: one multi-batch call (the harness loop of main.cpp:201-251);
        # the last group of a run of n steps may be shorter, so that exactly n steps are processed
        s = i % S
        if s != S - 1 and i != n - 1:
            return
        gs = s + 1  # batches in this group
        g0 = i - s  # first batch of the group; the query set is cycled
        qb = g0 % n_qbatches
        if qb + gs > n_qbatches:
            qb = 0
        qp = q_dev.data_ptr() + qb * BATCH * DIM * 4
        if world == 1:
            bf.search_dev_multi(qp, gs, BATCH, K, out_i.data_ptr(), out_d.data_ptr(), flags.data_ptr(), sptr)
        else:
            bf.search_dev_multi(qp, gs, BATCH, K, loc_i_ptr, loc_d_ptr, 0, sptr)
            all_gather(gath, loc)
            pkg.topk_merge_dev(gath.data_ptr(), gath.data_ptr() + lay.ids_offset * 4, world, S * BATCH, K1, K1,
                               out_d.data_ptr(), out_i.data_ptr(), flags.data_ptr(), sptr, stride_g=lay.stride_g)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def all_gather(dst, src):
        """ONE collective for the per-shard top-k lists (RCCL over xGMI; gloo only in rehearsals)."""
        if args.backend == "gloo":
            torch.cuda.current_stream().synchronize()
            hd = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_gather_into_tensor(hd, src.cpu())
            dst.copy_(hd)
        else:
            dist.all_gather_into_tensor(dst, src)

    def timed(step_fn, nsteps, nwarm):
        for i in range(nwarm):
            step_fn(i, nwarm)
        barrier()
        t = time.perf_counter()
        for i in range(nsteps):
            step_fn(i, nsteps)
        barrier()
        el = time.perf_counter() - t
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    steps = max(1, steps)
    warmup = max(1, warmup)

    # correctness spot check on the first batch before timing (rank 0, against the oracle at small scale
    # happens in tests; here: self-consistency of the sharded path vs. properties)
    bf.prof_enable(not args.no_prof)
    elapsed = timed(bf_step, steps, warmup)
    kern_ms, kern_n = bf.prof_read(0)
    bf.prof_enable(False)
    qps = steps * BATCH / elapsed
    ms_per_step = elapsed / steps * 1e3
    # the prof window covers the warmup + timed launches = warmup + steps batches (the last launch of a run may
    # hold fewer than S batches): scale to a launch of S batches
    kern_avg_s = (kern_ms * 1e-3) / (warmup + steps) * S if kern_n else 0.0
    rows_local = r1 - r0
    algo_bytes = (4 * rows_local * DIM + 4 * rows_local + 4 * BATCH * DIM + 8 * BATCH * K) * S  # SURVEY.md 8(d) x S batches per launch
    achieved = algo_bytes / kern_avg_s / 1e9 if kern_avg_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_bf_scan.json")
    if os.path.exists(tpath) and world == 1 and n_rows == N_BASE:
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_batch") * S
        except Exception:
            traffic = None
    log(f"brute force: {qps:.0f} QPS, {ms_per_step * 1e3:.1f} us/step, scan kernel {kern_avg_s * 1e6:.1f} us "
        f"({achieved:.0f} GB/s algorithmic)")

    # exactness guard on what was just timed: ascending, finite, ids in range
    torch.cuda.synchronize()
    od = out_d.cpu().numpy()
    oi = out_i.cpu().numpy()
    assert np.all(np.diff(od, axis=1) >= 0) and np.all((oi >= 0) & (oi < n_rows)), "bench output is not a valid top-k"

    # ---------------------------------------------------------------- int8 extra
    int8_info = None
    if not args.no_int8:
        try:
            bf.set_precision(2)
        except pkg.VSearchError:
            bf.set_precision(1)
        else:
            fp32_d, fp32_i = od.copy(), oi.copy()
            bf.prof_enable(not args.no_prof)
            el8 = timed(bf_step, steps, warmup)
            k8_ms, k8_n = bf.prof_read(0)
            bf.prof_enable(False)
            torch.cuda.synchronize()
            assert np.array_equal(out_d.cpu().numpy(), fp32_d) and np.array_equal(out_i.cpu().numpy(), fp32_i), \
                "int8 path differs from the fp32 path"
            k8_s = (k8_ms * 1e-3) / (warmup + steps) * S if k8_n else 0.0
            b8 = (rows_local * DIM + 4 * rows_local + 4 * BATCH * DIM + 8 * BATCH * K) * S  # u8 rows + i32 row terms
            int8_info = {"metric": "QPS, same workload, rows stored as u8 + int8 MFMA (bit-identical results)",
                         "value": round(steps * BATCH / el8, 1), "ms_per_step": round(el8 / steps * 1e3, 5),
                         "kernel_us_per_launch": round(k8_s * 1e6, 1), "batches_per_launch": S,
                         "roofline": {"bound": "hbm", "achieved": round(b8 / k8_s / 1e9, 1) if k8_s > 0 else None,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(b8 / k8_s / 1e9 / HBM_PEAK_GBS, 4) if k8_s > 0 else None,
                                      "traffic": None, "algorithmic_bytes_per_launch": b8,
                                      "note": "128 MB of rows fit the 256 MB Infinity Cache, so the rate can exceed HBM's"}}
            log(f"int8 path: {int8_info['value']:.0f} QPS, {int8_info['ms_per_step'] * 1e3:.1f} us/step")
            bf.set_precision(1)

    # ---------------------------------------------------------------- IVF extra
    ivf_info = None
    if not args.no_ivf:
        t0 = time.time()
        # every rank builds the same index deterministically from the full base (index build is
        # outside the timed region and outside the graded path)
        full = shard if world == 1 else pkg.synth_sift(n_rows, seed=SEED_BASE)
        nlist = pkg.clamp_nlist(n_rows, NLIST)
        if args.torch_kmeans:
            full_dev = torch.from_numpy(full).to(dev)
            cents, assign = build_ivf_torch(full_dev, nlist, args.kmeans_iters, seed=42)
            vr, off, r2o = pkg.ivf_layout_from_assignment(full, assign.cpu().numpy(), nlist)
            cents_h = cents.cpu().numpy()
            del full_dev, assign
            torch.cuda.empty_cache()
        else:
            # native builder: k-means on the library's own MFMA scan kernel, deterministic
            vr, off, r2o, cents_h, n_it = pkg.ivf_build(full, nlist, max_iter=args.kmeans_iters, seed=42, device=local_rank)
            log(f"vs_ivf_build: {n_it} Lloyd iterations")
        ivf = pkg.IVFIndex(vectors_reordered=vr, centroids=cents_h, cluster_offsets=off, reorder_to_original=r2o,
                           device=local_rank, rank=rank, world=world)
        sizes = np.diff(off)
        log(f"IVF index: nlist={nlist}, list sizes min/avg/max = {sizes.min()}/{sizes.mean():.0f}/{sizes.max()}, "
            f"built in {time.time() - t0:.1f}s")
        ilay = pkg.GatherLayout(S, BATCH, K)
        iloc = torch.zeros((ilay.words,), dtype=torch.int32, device=dev)
        igath = torch.zeros((world * ilay.words,), dtype=torch.int32, device=dev) if world > 1 else None
        iout_d = torch.zeros((S * BATCH, K), dtype=torch.float32, device=dev)
        iout_i = torch.zeros((S * BATCH, K), dtype=torch.int32, device=dev)

        def ivf_step(i, n):
            # like bf_step: groups of (up to) S independent batches per call (four streams inside the library)
            s = i % S
            if s != S - 1 and i != n - 1:
                return
            gs = s + 1
            qb = (i - s) % n_qbatches
            if qb + gs > n_qbatches:
                qb = 0
            qp = q_dev.data_ptr() + qb * BATCH * DIM * 4
            if world == 1:
                ivf.search_dev_multi(qp, gs, BATCH, K, NPROBE, iout_i.data_ptr(), iout_d.data_ptr(), sptr)
            else:
                ivf.search_dev_multi(qp, gs, BATCH, K, NPROBE, iloc.data_ptr() + ilay.id_offset(0) * 4,
                                     iloc.data_ptr() + ilay.dist_offset(0) * 4, sptr)
                all_gather(igath, iloc)
                pkg.topk_merge_dev(igath.data_ptr(), igath.data_ptr() + ilay.ids_offset * 4, world, S * BATCH, K, K,
                                   iout_d.data_ptr(), iout_i.data_ptr(), 0, sptr, stride_g=ilay.stride_g)

        ivf.prof_enable(True)
        iel = timed(ivf_step, steps, warmup)
        okern_ms, okern_n = ivf.prof_read(1)  # scan kernel while other batches' kernels share the GPU
        ivf.prof_enable(False)
        # the scan kernel alone (one stream, batch by batch): the figure its roofline is computed from
        ivf.prof_enable(True)
        for i in range(64):
            ivf.search_dev(q_dev.data_ptr() + (i % n_qbatches) * BATCH * DIM * 4, BATCH, K, NPROBE, iout_i.data_ptr(),
                           iout_d.data_ptr(), sptr)
        torch.cuda.synchronize()
        ikern_ms, ikern_n = ivf.prof_read(1)
        ivf.prof_enable(False)
        ivf_qps = steps * BATCH / iel
        # recall@1 / recall@5 against exact ground truth from the brute-force path (N = 1 only: the
        # sharded variants are covered by tests)
        rec1 = rec5 = None
        avg_cand = None
        if world == 1:
            nrec = 1024
            gt_ids, _ = bf.search(queries[:nrec], K)
            ids, _, total = ivf.searchBatch(queries[:nrec], nrec, K, NPROBE)
            rec1 = float(np.mean(ids[:, 0] == gt_ids[:, 0]))
            rec5 = float(np.mean([len(set(ids[i]) & set(gt_ids[i])) / K for i in range(nrec)]))
            avg_cand = total / nrec
        ivf_info = {"metric": "ivf_qps", "value": round(ivf_qps, 1), "ms_per_step": round(iel / steps * 1e3, 4),
                    "nlist": nlist, "nprobe": NPROBE, "batch": BATCH, "recall_at_1": rec1, "recall_at_5": rec5,
                    "avg_candidates": avg_cand,
                    "scan_kernel_us": round(ikern_ms / max(ikern_n, 1) * 1e3, 2),
                    "scan_kernel_us_overlapped": round(okern_ms / max(okern_n, 1) * 1e3, 2),
                    "note": "value and roofline: vs_ivf_search_dev_multi, every kernel launched once per group of S batches "
                            "(scan_kernel_us_overlapped = that launch); scan_kernel_us: the scan of ONE batch alone"}
        if avg_cand:
            # The list-major scan reads every probed list ONCE per batch, so its algorithmic bytes are
            # (4d + 4) * rows of the distinct lists probed by the batch (+ 4 B per (query, row) score written),
            # not SURVEY 8(d)'s per-query (4d + 8) * S_q, which assumes one pass per query.
            cn = (cents_h.astype(np.float64) ** 2).sum(1)
            uniq_rows, nb_s = 0, 8
            for b0 in range(nb_s):
                qs = queries[b0 * BATCH:(b0 + 1) * BATCH].astype(np.float64)
                pr = np.argsort(cn[None, :] - 2.0 * qs @ cents_h.astype(np.float64).T, axis=1)[:, :NPROBE]
                uniq_rows += int(sizes[np.unique(pr)].sum())
            uniq_rows /= nb_s
            # rows: the exact int8 copy (d + 4 bytes per row) when base and queries are byte valued, else fp32 rows
            i8_rows = bool(np.all(full == np.floor(full)) and full.min() >= 0 and full.max() <= 255)
            row_bytes = (DIM + 4) if i8_rows else (4 * DIM + 4)
            ib = row_bytes * uniq_rows + 4 * avg_cand * BATCH
            # the timed region launches the scan once per group of S batches (blockIdx.y = batch)
            ks = okern_ms / max(okern_n, 1) * 1e-3
            itraffic = None
            ipath = os.path.join(ROOT, "profiles", "traffic_ivf_list_scan.json")
            if os.path.exists(ipath) and n_rows == N_BASE and i8_rows:
                itraffic = json.load(open(ipath)).get("hbm_bytes_per_launch_32_batches")
            ivf_info["roofline"] = {"bound": "hbm", "achieved": round(ib * S / ks / 1e9, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(ib * S / ks / 1e9 / HBM_PEAK_GBS, 4), "traffic": itraffic,
                                    "kernel": "vs::ivf_unit_scan_kernel", "kernel_us": round(ks * 1e6, 2), "batches_per_launch": S,
                                    "algorithmic_bytes_per_launch": int(ib * S), "row_bytes": row_bytes,
                                    "distinct_rows_per_batch": int(uniq_rows),
                                    "per_query_pass_bytes": int((4 * DIM + 8) * avg_cand * BATCH)}
        log(f"IVF: {ivf_qps:.0f} QPS, recall@1={rec1}, recall@5={rec5}, avg candidates={avg_cand}")
        ivf.close()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1)
    cpu_info = None
    if world == 1 and rank == 0 and not args.no_cpu:
        import oracle
        oracle.search_bf(shard, queries[:2], K)  # thread pool + page-in warm-up
        # bounded sample: chunks of 128 queries until ~15 s of CPU work (or 2048 queries) are done
        tm = {"dist_s": 0.0, "topk_s": 0.0}
        cids, cds, nq_cpu, cel = [], [], 0, 0.0
        while nq_cpu < 2048 and cel < 15.0:
            t1 = {}
            tq = time.perf_counter()
            ci_, cd_ = oracle.search_bf(shard, queries[nq_cpu:nq_cpu + 128], K, t1)
            cel += time.perf_counter() - tq
            tm["dist_s"] += t1["dist_s"]
            tm["topk_s"] += t1["topk_s"]
            cids.append(ci_)
            cds.append(cd_)
            nq_cpu += 128
        cid, cd = np.concatenate(cids), np.concatenate(cds)
        gid, gd = bf.search(queries[:nq_cpu], K)
        assert np.array_equal(gid, cid) and np.array_equal(gd, cd), "GPU result differs from the CPU oracle"
        cpu_info = {"value": round(nq_cpu / cel, 2), "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
                    "sample": f"{nq_cpu} queries x {n_rows} base rows, k={K}, oracle/vs_oracle.c (restates cpu_baseline.cpp: "
                              f"GEMV + L2 epilogue + select_topk, serial over queries, OpenMP inside); "
                              f"dist {tm['dist_s']:.2f}s topk {tm['topk_s']:.2f}s; ids+dists equal to the GPU's"}
        log(f"cpu baseline: {cpu_info['value']} QPS on {cpu_info['cores']} threads")

    if rank == 0:
        line = {
            "metric": "QPS, SIFT-1M brute-force batch=32 k=5 (exact L2)",
            "value": round(qps, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"SIFT-1M-shaped synthetic {n_rows}x{DIM} fp32 base, brute force, batch={BATCH}, k={K}",
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "collective_every_steps": S if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "vs::scan_kernel<2,8,0>", "kernel_us": round(kern_avg_s * 1e6, 2),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "batches_per_launch": S,
                         "mfma_tflops": round(2.0 * BATCH * rows_local * DIM * S / max(kern_avg_s, 1e-12) / 1e12, 2)},
            "cpu_baseline": cpu_info,
            "ivf": ivf_info,
            "bf_int8": int8_info,
        }
        print(json.dumps(line), flush=True)
    bf.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
