#!/usr/bin/env python3
"""bench.py -- QPS of the distance + top-k hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a launcher: this script starts N child ranks itself, before anything touches the GPU;
   under `python -m torch.distributed.run --nproc-per-node N ...` it is one of the ranks.)

A *step* is one pass of the hot path over one batch of 32 synthetic queries that are already
resident in HBM: exact brute force, k = 5, against the synthetic SIFT-1M base (1 000 000 x 128 fp32,
generator of SURVEY.md 8d, seeds 20251205 / 20251206).  `value` = queries / second of the whole job.
The timed region of exactly K steps is repeated `--repeats` times (barrier + synchronize on both
sides, max over ranks each time); `value` / `ms_per_step` are the median region.

N = 1  : config "SIFT-1M brute-force batch=32 on 1xMI355X" (BASELINE.json configs[2]).
N > 1  : the same base row-sharded over the ranks (strong scaling); every rank scans its shard for
         the same queries, the per-shard top-(k+1) lists are exchanged with ONE RCCL all-gather per
         group of `--coll-every` steps and merged on the device (the only collective the path has).
Extras on the same JSON line: "ivf" (nlist=1024 nprobe=32, configs[3]/[4]), "ivf_nprobe8", "bf_int8", "q8_runner",
"host_api" (vs_bf_search / vs_ivf_search on host buffers: upload, download and reference tie order
inside the timed region -- SURVEY 8(d)'s QPS definition), "siftsmall_b1", "sift1m_b1" (configs[1]).

The line also carries `roofline` (dominant kernel vs the HBM roof, timed with HIP events on the
stream it runs on) and `cpu_baseline` (the oracle's restatement of cpu_baseline.cpp on this box's
host cores, bounded sample, rank 0 at N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BASE = 1_000_000
N_SMALL = 10_000
DIM = 128
BATCH = 32
K = 5
NLIST = 1024
NPROBE = 32
SEED_BASE, SEED_QUERY = 20251205, 20251206
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: dense fp32 MFMA (256 CUs x 256 FLOP/clk x 2.4 GHz)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes.  The parent
    has not imported torch or touched HIP at this point and never does."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region of K steps is repeated R times and the median is reported; 0 = as many as fit "
                         "a quarter of a second (at least 5, at most 200): a short region (K = 20 is under 2 ms) is at the mercy of "
                         "clock ramps and launch jitter, many of them back to back are not")
    ap.add_argument("--coll-every", type=int, default=32,
                    help="batches per multi-batch call (and per all-gather + merge when N > 1)")
    ap.add_argument("--rows", type=int, default=N_BASE, help="base rows (default SIFT-1M)")
    ap.add_argument("--no-ivf", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not time the scan kernel with HIP events")
    ap.add_argument("--no-int8", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host_api / B=1 / nprobe=8 legs")
    ap.add_argument("--only-b1", action="store_true", help="of the extras run only the B = 1 legs (quick check)")
    ap.add_argument("--kmeans-iters", type=int, default=20)
    ap.add_argument("--collective", default="auto", choices=["auto", "library", "torch"],
                    help="N > 1: RCCL all-gather inside libvsearch_hip.so (vs_comm_*) or torch.distributed's")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N > 1 path on ONE GPU (all ranks share cuda:0, gathers staged through host)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # no launcher: become one (nothing has initialised the GPU in this process)
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    steps, warmup, S = max(1, args.steps), max(1, args.warmup), max(1, min(32, args.coll_every))
    R = args.repeats  # 0 = chosen from the first region's length (the same on every rank: the time is the max over ranks)
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
    n_qbatches = n_queries // BATCH

    # ---------------------------------------------------------------- the one collective (N > 1)
    comm = None
    coll_kind = None
    if world > 1:
        coll_kind = "torch-rccl" if args.backend == "nccl" else "gloo-rehearsal"
        if args.backend == "nccl" and args.collective in ("auto", "library") and hasattr(pkg, "Comm"):
            try:
                uid = [pkg.Comm.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = pkg.Comm(uid[0], rank, world, local_rank)
                coll_kind = "library-rccl"
            except Exception as e:  # the torch collective below is the same all-gather driven from here
                if args.collective == "library":
                    raise
                log(f"rank {rank}: library RCCL communicator unavailable ({e}); using torch.distributed's all-gather")
                comm = None
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
            coll_kind = "torch-rccl"

    loc = torch.zeros((2 * S * BATCH * K1,), dtype=torch.int32, device=dev)
    gath = torch.zeros((world * loc.numel(),), dtype=torch.int32, device=dev) if world > 1 else None
    out_d = torch.zeros((S * BATCH, K1), dtype=torch.float32, device=dev)
    out_i = torch.zeros((S * BATCH, K1), dtype=torch.int32, device=dev)
    flags = torch.zeros((S * BATCH,), dtype=torch.int32, device=dev)

    def all_gather(dst, src):
        """ONE collective for the per-shard top-k lists (RCCL over xGMI; gloo only in rehearsals)."""
        if args.backend == "gloo":
            torch.cuda.current_stream().synchronize()
            hd = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_gather_into_tensor(hd, src.cpu())
            dst.copy_(hd)
        else:
            dist.all_gather_into_tensor(dst, src)

    def group_of(i, n):
        """steps are issued in groups of S batches (one multi-batch call = the harness loop of main.cpp:201-251);
        the last group of a run of n steps may be shorter, so that exactly n steps are processed"""
        s = i % S
        if s != S - 1 and i != n - 1:
            return None
        gs = s + 1
        qb = (i - s) % n_qbatches
        if qb + gs > n_qbatches:
            qb = 0
        return gs, q_dev.data_ptr() + qb * BATCH * DIM * 4

    def sharded_call(search_local, search_sharded, gs, qp, kk, o_i, o_d, fl):
        """N > 1: local top-kk per shard -> one all-gather of gs * BATCH lists per rank -> device merge."""
        if comm is not None:
            search_sharded(comm, qp, gs, BATCH, o_i.data_ptr(), o_d.data_ptr(), fl, sptr)
            return
        lay = pkg.GatherLayout(gs, BATCH, kk)
        lv, gv = loc[:lay.words], gath[:world * lay.words]
        search_local(qp, gs, lv.data_ptr() + lay.ids_offset * 4, lv.data_ptr())
        all_gather(gv, lv)
        pkg.topk_merge_dev(gv.data_ptr(), gv.data_ptr() + lay.ids_offset * 4, world, gs * BATCH, kk, kk,
                           o_d.data_ptr(), o_i.data_ptr(), fl, sptr, stride_g=lay.stride_g)

    def bf_step(i, n):
        g = group_of(i, n)
        if g is None:
            return
        gs, qp = g
        if world == 1:
            bf.search_dev_multi(qp, gs, BATCH, K, out_i.data_ptr(), out_d.data_ptr(), flags.data_ptr(), sptr)
        else:
            sharded_call(lambda q, nb, ip, dp: bf.search_dev_multi(q, nb, BATCH, K, ip, dp, 0, sptr),
                         lambda c, q, nb, B, ip, dp, fl, st: bf.search_dev_sharded(c, q, nb, B, K, ip, dp, fl, st),
                         gs, qp, K1, out_i, out_d, flags.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, nsteps, nwarm, repeats=None):
        """W warm-up steps, then R regions of exactly `nsteps` steps, each bracketed by barrier + synchronize, max over
        ranks per region; returns the list of region times."""
        for i in range(nwarm):
            step_fn(i, nwarm)
        out = []
        want = repeats if repeats is not None else R
        while True:
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
            out.append(el)
            if want <= 0:
                want = int(min(200, max(5, 0.25 / max(el, 1e-6))))
            if len(out) >= want:
                return out

    # ---------------------------------------------------------------- brute force fp32 (the headline)
    bf.prof_enable(not args.no_prof)
    regions = timed(bf_step, steps, warmup)
    kern_ms, kern_n = bf.prof_read(0)
    bf.prof_enable(False)
    elapsed = median(regions)
    qps = steps * BATCH / elapsed
    ms_per_step = elapsed / steps * 1e3
    # the prof window covers every launch of the warm-up and of the R regions = warmup + R * steps batches (the last
    # launch of a run may hold fewer than S batches): time per batch, scaled to a launch of S batches
    R_used = len(regions)
    n_batches_prof = warmup + R_used * steps
    kern_per_batch_s = (kern_ms * 1e-3) / n_batches_prof if kern_n else 0.0
    kern_avg_s = kern_per_batch_s * S
    rows_local = r1 - r0
    batch_bytes = 4 * rows_local * DIM + 4 * rows_local + 4 * BATCH * DIM + 8 * BATCH * K
    algo_bytes = batch_bytes * S
    achieved = algo_bytes / kern_avg_s / 1e9 if kern_avg_s > 0 else 0.0
    # launches of at least two batches share a pass over the rows between two batches (VSEARCH_F32_PAIR, default on): a
    # tile is then 128 MFMAs for 8 KB and the kernel is bound by the fp32 MFMA pipe, not by HBM
    # (the library's rule, vs_api.hip bf_launch: shards with at least 96 tiles of 16 rows per workgroup and pass)
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    pair_env = int(os.environ.get("VSEARCH_F32_PAIR", "1") or 1)
    pair = S >= 2 and min(steps, S) >= 2 and (pair_env > 1 or (pair_env == 1 and ((rows_local + 15) // 16) // max(min(n_cus, 256), 1) >= 96))
    mfma_tflops = 2.0 * BATCH * rows_local * DIM / max(kern_per_batch_s, 1e-12) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_bf_scan.json")
    if os.path.exists(tpath) and world == 1 and n_rows == N_BASE:
        try:
            tj = json.load(open(tpath))
            if int(tj.get("batches_per_pass", 1)) == (2 if pair else 1):
                traffic = int(tj.get("hbm_bytes_per_batch") * S)
        except Exception:
            traffic = None
    log(f"brute force: {qps:.0f} QPS, {ms_per_step * 1e3:.1f} us/step (median of {R_used} regions, min "
        f"{min(regions) / steps * 1e6:.1f} max {max(regions) / steps * 1e6:.1f}), scan kernel {kern_per_batch_s * 1e6:.1f} us/batch "
        f"({mfma_tflops:.1f} TFLOP/s fp32 MFMA, {achieved:.0f} GB/s of SURVEY 8(d) bytes, {2 if pair else 1} batch(es) per pass)")

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
            # an extra, timed in its own steady state: regions of at least one full launch (S batches), whatever --steps
            # says for the headline (20 batches are 5 passes of 4: the 8 waves of a workgroup cannot share them evenly)
            steps8 = max(steps, S) if world == 1 else steps
            bf.prof_enable(not args.no_prof)
            reg8 = timed(bf_step, steps8, warmup)
            k8_ms, k8_n = bf.prof_read(0)
            bf.prof_enable(False)
            torch.cuda.synchronize()
            i8_d, i8_i = out_d.cpu().numpy().copy(), out_i.cpu().numpy().copy()
            bf.set_precision(1)  # the same last call once more through the fp32 rows: the results must be the same bits
            for i in range(steps8 - ((steps8 - 1) % S + 1), steps8):
                bf_step(i, steps8)
            torch.cuda.synchronize()
            assert np.array_equal(out_d.cpu().numpy(), i8_d) and np.array_equal(out_i.cpu().numpy(), i8_i), \
                "int8 path differs from the fp32 path"
            el8 = median(reg8)
            k8_s = (k8_ms * 1e-3) / (warmup + len(reg8) * steps8) * S if k8_n else 0.0
            # the wide scan serves VSEARCH_I8_WIDE / 2 batches of 32 queries per pass over the rows (default 4): its
            # algorithmic bytes per launch are one pass of u8 rows + i32 row terms per GROUP of batches
            bpp = max(1, int(os.environ.get("VSEARCH_I8_WIDE", "8")) // 2)
            passes = (S + bpp - 1) // bpp
            b8 = (rows_local * DIM + 4 * rows_local) * passes + (BATCH * DIM + 8 * BATCH * K) * S
            b8_per_batch_unit = (rows_local * DIM + 4 * rows_local + 4 * BATCH * DIM + 8 * BATCH * K) * S
            int8_info = {"metric": "QPS, same workload, rows stored as u8 + int8 MFMA (bit-identical results); "
                                   f"{bpp} batches share one pass over the rows",
                         "value": round(steps8 * BATCH / el8, 1), "ms_per_step": round(el8 / steps8 * 1e3, 5), "steps": steps8,
                         "kernel_us_per_launch": round(k8_s * 1e6, 1), "batches_per_launch": S, "batches_per_row_pass": bpp,
                         "roofline": {"bound": "hbm", "achieved": round(b8 / k8_s / 1e9, 1) if k8_s > 0 else None,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(b8 / k8_s / 1e9 / HBM_PEAK_GBS, 4) if k8_s > 0 else None,
                                      "traffic": None, "algorithmic_bytes_per_launch": b8,
                                      "bytes_if_every_batch_streamed_the_rows": b8_per_batch_unit,
                                      "note": "the 128 MB of rows fit the 256 MB Infinity Cache; with several batches per pass the "
                                              "scan is bound by VALU/MFMA issue and LDS-DMA delivery per CU, not by HBM; HBM peak is "
                                              "the reference roof"}}
            log(f"int8 path: {int8_info['value']:.0f} QPS, {int8_info['ms_per_step'] * 1e3:.1f} us/step")
            bf.set_precision(1)

    # ---------------------------------------------------------------- host-buffer API (SURVEY 8(d)'s QPS definition)
    host_info = None
    if world == 1 and not args.no_extras and not args.only_b1:
        nq_h = n_queries
        tm = pkg.Timing()
        bf.search(queries[:nq_h], K)  # pinned staging + tie scratch allocated here, not in the timed calls
        th = []
        for _ in range(5):
            t = time.perf_counter()
            hid, hdd = bf.search(queries[:nq_h], K, tm)
            th.append(time.perf_counter() - t)
        host_info = {"bf": {"metric": "QPS of vs_bf_search on host buffers (query upload, result download and reference tie "
                                      "order inside the timed call)", "value": round(nq_h / median(th), 1), "queries_per_call": nq_h,
                            "tie_queries": int(tm.tie_queries), "tie_resolve_ms": round(tm.tie_resolve_ms, 3),
                            "vs_device_api": round(nq_h / median(th) / qps, 4)}}
        log(f"host API brute force: {host_info['bf']['value']:.0f} QPS ({nq_h} queries per call, {tm.tie_queries} tie queries, "
            f"{tm.tie_resolve_ms:.2f} ms resolving them)")

    # ---------------------------------------------------------------- B = 1 legs (BASELINE.json configs[1])
    b1_info = {}
    if world == 1 and not args.no_extras:
        o1_d = torch.zeros((32, K1), dtype=torch.float32, device=dev)
        o1_i = torch.zeros((32, K1), dtype=torch.int32, device=dev)
        f1 = torch.zeros((32,), dtype=torch.int32, device=dev)

        def b1_leg(index, rows, tag):
            # (a) what a caller issuing one query at a time gets (cpu_baseline.cpp:222-254: one query per iteration):
            #     ONE single-query call, synchronised -- call + launch + scan + in-kernel merge + sync on the host clock --
            #     and the same launch's device time alone (HIP events around 64 back-to-back single-query calls)
            lat = []
            for i in range(80):
                torch.cuda.synchronize()
                t = time.perf_counter()
                index.search_dev(q_dev.data_ptr() + i * DIM * 4, 1, K, o1_i.data_ptr(), o1_d.data_ptr(), f1.data_ptr(), sptr)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t)
            lat_us = median(lat[16:]) * 1e6
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dev_us = []
            for rep in range(3):
                e0.record(stream)
                for i in range(64):
                    index.search_dev(q_dev.data_ptr() + i * DIM * 4, 1, K, o1_i.data_ptr(), o1_d.data_ptr(), f1.data_ptr(), sptr)
                e1.record(stream)
                torch.cuda.synchronize()
                dev_us.append(e0.elapsed_time(e1) * 1e3 / 64)
            call_us = median(dev_us)
            # (b) many single-query batches handed over in one call (32 per launch; the launch shares passes over the rows
            #     between batches, so this is NOT "each query streams the base once" and carries no HBM fraction)
            def step(i, n):
                if i % 32 != 31 and i != n - 1:
                    return
                gs = i % 32 + 1
                index.search_dev_multi(q_dev.data_ptr() + ((i - gs + 1) % 1024) * DIM * 4, gs, 1, K, o1_i.data_ptr(),
                                       o1_d.data_ptr(), f1.data_ptr(), sptr)
            nst = 640
            reg = timed(step, nst, 64, repeats=3)
            us_q = median(reg) / nst * 1e6
            bytes_q = 4 * rows * DIM + 4 * rows + 4 * DIM + 8 * K
            info = {"latency_us_single_call": round(lat_us, 1), "us_per_call_back_to_back": round(call_us, 2),
                    "qps_single_calls_back_to_back": round(1e6 / call_us, 1),
                    "rows": rows, "batch": 1, "algorithmic_bytes_per_query": bytes_q,
                    "hbm_frac": round(bytes_q / (call_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "hbm_frac_of_single_call_latency": round(bytes_q / (lat_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "us_per_query_32_per_launch": round(us_q, 2),
                    "note": "hbm_frac = SURVEY 8(d) bytes of one query / device time of one single-query call (one launch: "
                            "vs::scan_one_kernel) / 8 TB/s; latency_us_single_call adds the host's launch + synchronise"}
            assert info["hbm_frac"] <= 1.0, info
            log(f"{tag}: single call {lat_us:.1f} us on the host clock, {call_us:.2f} us per call back to back "
                f"({info['hbm_frac']:.3f} of the HBM roof), {us_q:.2f} us/query at 32 single-query batches per launch")
            return info

        b1_info["sift1m_b1"] = b1_leg(bf, rows_local, "SIFT-1M B=1")
        small = pkg.synth_sift(N_SMALL, seed=SEED_BASE)
        bfs = pkg.BruteForceIndex(small, device=local_rank)
        bfs.set_precision(1)
        b1_info["siftsmall_b1"] = b1_leg(bfs, N_SMALL, "SIFT-small B=1")
        b1_info["siftsmall_b1"]["note"] = "5 MB base: cache resident, launch-latency bound (SURVEY 8(d)): absolute us are the figure"
        bfs.close()

    # ---------------------------------------------------------------- UFIXED_POINT_8 score path (second half)
    q8_info = None
    if world == 1 and not args.no_extras and not args.only_b1:
        try:
            in_s = float(queries.max()) / 255.0
            w_s = float(shard.max()) / 255.0
            samp = shard[:: max(1, rows_local // 4096)].astype(np.float64)
            o_s = 1.05 * float((queries[:64].astype(np.float64) @ samp.T).max()) / 255.0  # min-max calibration on a sample
            qr = pkg.Q8Runner(shard, in_s, w_s, 0, o_s, device=local_rank)
            n_pad = (rows_local + 63) // 64 * 64
            sc = torch.empty((BATCH * n_pad,), dtype=torch.uint8, device=dev)
            q8_i = torch.empty((S * BATCH, K), dtype=torch.int32, device=dev)
            q8_t = torch.empty((S * BATCH, K), dtype=torch.uint8, device=dev)

            def q8_exec(i, n):
                qr.execute_dev(q_dev.data_ptr() + (i % n_qbatches) * BATCH * DIM * 4, BATCH, sc.data_ptr(), n_pad, sptr)

            def q8_search(i, n):
                if i % S != S - 1 and i != n - 1:
                    return
                gs = i % S + 1
                qr.search_dev(q_dev.data_ptr() + ((i - gs + 1) % n_qbatches) * BATCH * DIM * 4, gs, BATCH, K, q8_i.data_ptr(),
                              q8_t.data_ptr(), sptr)

            # kernel time by HIP events on the stream the launches go to (torch's current stream)
            for i in range(8):
                q8_exec(i, 8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrun = 64
            e0.record(stream)
            for i in range(nrun):
                q8_exec(i, nrun)
            e1.record(stream)
            torch.cuda.synchronize()
            ex_us = e0.elapsed_time(e1) * 1e3 / nrun
            reg = timed(q8_search, steps, warmup, repeats=5)
            el = median(reg)
            b_q8 = rows_local * DIM + 4 * rows_local + BATCH * rows_local + 4 * BATCH * DIM  # u8 rows + row terms in, u8 scores out
            # the check of what was just timed: the last batch's ids must be the top-k of its own score matrix
            q8_exec((steps - 1), steps)
            torch.cuda.synchronize()
            raw = sc.view(BATCH, n_pad)[:, :rows_local].cpu().numpy()
            want_top = -np.sort(-raw.astype(np.int16), axis=1)[:, :K]
            got_i = q8_i.cpu().numpy()[(steps - 1) % S * BATCH:][:BATCH] - r0
            assert np.array_equal(q8_t.cpu().numpy()[(steps - 1) % S * BATCH:][:BATCH].astype(np.int16), want_top), "q8 top-k scores"
            assert np.array_equal(np.take_along_axis(raw, got_i.astype(np.int64), 1).astype(np.int16), want_top), "q8 top-k ids"
            q8_info = {"metric": "QPS, same workload through the UFIXED_POINT_8 runner (QnnRunner::executeBatchRaw + find_top_k_int8): "
                                 "uint8 queries x uint8 database, int8 MFMA, raw uint8 [B x N] score matrix written, top-k over it",
                       "value": round(steps * BATCH / el, 1), "ms_per_step": round(el / steps * 1e3, 5),
                       "score_matrix_us_per_batch": round(ex_us, 2),
                       "encodings": {"input_scale": in_s, "weight_scale": w_s, "weight_offset": 0, "output_scale": o_s},
                       "distinct_scores_in_last_batch": int(len(np.unique(raw))),
                       "roofline": {"bound": "hbm", "achieved": round(b_q8 / (ex_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(b_q8 / (ex_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                    "kernel": "q8_scores_kernel<2>", "kernel_us": round(ex_us, 2),
                                    "algorithmic_bytes_per_launch": b_q8,
                                    "note": "kernel_us = quantiser launch + score kernel per batch (HIP events over 64 batches); "
                                            "the 128 MB of uint8 rows stay in the 256 MB Infinity Cache between batches, the "
                                            "32 MB score matrix is written per batch: HBM peak is the reference roof"},
                       "parity": "unpinned (the NPU graph is not in the reference); bit-exact against oracle.q8_* in tests/test_gpu_q8.py"}
            log(f"q8 runner: {q8_info['value']:.0f} QPS with top-k, score matrix {ex_us:.1f} us per batch "
                f"({q8_info['roofline']['achieved']:.0f} GB/s of {b_q8 / 1e6:.0f} MB)")
            qr.close()
            del sc
        except Exception as e:  # an extra: it must not cost the headline line
            log(f"q8 leg failed: {type(e).__name__}: {e}")
            q8_info = {"error": f"{type(e).__name__}: {e}"}

    # ---------------------------------------------------------------- IVF extras
    ivf_info = None
    ivf8_info = None
    if not args.no_ivf:
        # (an extra: whatever goes wrong in it must not cost the headline line below)
        try:
            t0 = time.time()
            # every rank builds the same index deterministically from the full base (index build is
            # outside the timed region and outside the graded path)
            full = shard if world == 1 else pkg.synth_sift(n_rows, seed=SEED_BASE)
            nlist = pkg.clamp_nlist(n_rows, NLIST)
            # native builder: k-means on the library's own MFMA scan kernel, deterministic
            vr, off, r2o, cents_h, n_it = pkg.ivf_build(full, nlist, max_iter=args.kmeans_iters, seed=42, device=local_rank)
            log(f"vs_ivf_build: {n_it} Lloyd iterations")
            ivf = pkg.IVFIndex(vectors_reordered=vr, centroids=cents_h, cluster_offsets=off, reorder_to_original=r2o,
                               device=local_rank, rank=rank, world=world)
            sizes = np.diff(off)
            log(f"IVF index: nlist={nlist}, list sizes min/avg/max = {sizes.min()}/{sizes.mean():.0f}/{sizes.max()}, "
                f"built in {time.time() - t0:.1f}s")
            # one vs_ivf_search_dev_multi call takes SI batches = ONE launch group: every kernel of the pipeline is launched
            # once for all of them, the list scan makes one pass per super-batch of 32 batches (N = 1: 256 batches; N > 1:
            # 32 per rank, the cluster-sharded pipeline's group: slice r's per-query stages run on rank r only)
            SI = 256 if world == 1 else min(256, 32 * world)
            q_ivf = torch.from_numpy(np.tile(queries, ((SI * BATCH + n_queries - 1) // n_queries, 1))[:SI * BATCH].copy()).to(dev)
            iout_d = torch.zeros((SI * BATCH, K), dtype=torch.float32, device=dev)
            iout_i = torch.zeros((SI * BATCH, K), dtype=torch.int32, device=dev)
            iloc = torch.zeros((2 * SI * BATCH * K,), dtype=torch.int32, device=dev) if world > 1 else None
            igath = torch.zeros((world * 2 * SI * BATCH * K,), dtype=torch.int32, device=dev) if world > 1 else None
            i8_rows = bool(np.all(full == np.floor(full)) and full.min() >= 0 and full.max() <= 255)
            row_bytes = (DIM + 4) if i8_rows else (4 * DIM + 4)
            cn = (cents_h.astype(np.float64) ** 2).sum(1)
            gt_ids = None
            if world == 1:
                gt_ids, _ = bf.search(queries[:1024], K)

            def ivf_leg(nprobe, index=None, tag="IVF"):
                index = index or ivf

                def ivf_step(i, n):
                    si = i % SI
                    if si != SI - 1 and i != n - 1:
                        return
                    gs = si + 1
                    qp = q_ivf.data_ptr()
                    if world == 1:
                        index.search_dev_multi(qp, gs, BATCH, K, nprobe, iout_i.data_ptr(), iout_d.data_ptr(), sptr)
                    elif comm is not None:
                        index.search_dev_sharded(comm, qp, gs, BATCH, K, nprobe, iout_i.data_ptr(), iout_d.data_ptr(), sptr)
                    else:  # (no library communicator: every rank runs the whole pipeline on its lists, torch's all-gather)
                        lay = pkg.GatherLayout(gs, BATCH, K)
                        lv, gv = iloc[:lay.words], igath[:world * lay.words]
                        index.search_dev_multi(qp, gs, BATCH, K, nprobe, lv.data_ptr() + lay.ids_offset * 4, lv.data_ptr(), sptr)
                        all_gather(gv, lv)
                        pkg.topk_merge_dev(gv.data_ptr(), gv.data_ptr() + lay.ids_offset * 4, world, gs * BATCH, K, K,
                                           iout_d.data_ptr(), iout_i.data_ptr(), 0, sptr, stride_g=lay.stride_g)

                # an extra, timed in its own steady state: regions of at least one full call (SI batches = one launch group),
                # whatever --steps says for the headline (the driver's 20 steps would be one 20-batch group: a caller issuing
                # 640 queries at a time gets about half the rate)
                isteps = max(steps, SI)
                index.prof_enable(True)
                ireg = timed(ivf_step, isteps, max(warmup, SI))
                okern_ms, okern_n = index.prof_read(1)
                index.prof_enable(False)
                iel = median(ireg)
                info = {"metric": "ivf_qps", "value": round(isteps * BATCH / iel, 1), "ms_per_step": round(iel / isteps * 1e3, 4),
                        "nlist": nlist, "nprobe": nprobe, "batch": BATCH, "batches_per_call": min(SI, isteps), "steps": isteps,
                        "kmeans_iterations": int(n_it)}
                if world == 1:
                    nrec = 1024
                    ids, _, total = index.searchBatch(queries[:nrec], nrec, K, nprobe)
                    info["recall_at_1"] = float(np.mean(ids[:, 0] == gt_ids[:, 0]))          # main_ivf.cpp:52-59 with k = 1
                    info["recall_at_5"] = float(np.mean([len(set(ids[i]) & set(gt_ids[i])) / K for i in range(nrec)]))
                    info["avg_candidates"] = total / nrec
                    # The list-major scan reads every list probed by the launch group (S batches = S * BATCH queries share ONE
                    # pass) once, so its algorithmic bytes per launch are row_bytes * rows of the distinct lists the group
                    # probes + 4 B per (query, probe) slot-table entry + the group's queries, not SURVEY 8(d)'s per-query
                    # (4d + 8) * S_q, which assumes one pass per query.
                    cents64 = cents_h.astype(np.float64)

                    rb = row_bytes if index.precision_used == 0 else (4 * DIM + 4)

                    def launch_bytes(gs):  # one scan launch = a group of gs batches = super-batches of 32 batches, one pass each
                        tot, rows_tot = 0, 0
                        gq_all = q_ivf[:gs * BATCH].cpu().numpy().astype(np.float64)
                        for b0 in range(0, gs, 32):
                            gq = gq_all[b0 * BATCH:(b0 + 32) * BATCH]
                            pr = np.argsort(cn[None, :] - 2.0 * gq @ cents64.T, axis=1)[:, :nprobe]
                            rows = int(sizes[np.unique(pr)].sum())
                            tot += rb * rows + 4 * pr.size + len(gq) * (DIM + 16 if rb < 512 else 4 * DIM + 16)
                            rows_tot += rows
                        return tot, rows_tot

                    def launches_of(n):  # sizes of the launch groups of a run of n steps: calls of <= SI batches = one group each
                        return [SI] * (n // SI) + ([n % SI] if n % SI else [])

                    window = launches_of(max(warmup, SI)) + launches_of(isteps) * len(ireg)   # every launch of the prof window
                    per_size = {gs: launch_bytes(gs) for gs in set(window)}
                    ib = sum(per_size[gs][0] for gs in window) / max(len(window), 1)   # mean algorithmic bytes per launch
                    uniq_rows = per_size[max(per_size)][1]
                    ks_launch = (okern_ms * 1e-3) / okern_n if okern_n else 0.0       # mean kernel time per launch
                    ach = ib / ks_launch / 1e9 if ks_launch > 0 else None
                    itraffic = None
                    ipath = os.path.join(ROOT, "profiles", "traffic_ivf_list_scan.json")
                    tkey = "hbm_bytes_per_launch_fp32_rows" if rb > 512 else "hbm_bytes_per_launch"
                    if os.path.exists(ipath) and n_rows == N_BASE and nprobe == NPROBE and set(window) == {SI}:
                        itraffic = json.load(open(ipath)).get(tkey)   # (PMC figure of a launch of SI batches, profiles/README.md)
                    # fp32 rows (466 MB per super-batch: no cache holds them): HBM binds, `achieved` = algorithmic bytes / time and
                    # a fraction above 1 would mean the byte model is wrong.  Byte rows (132 MB: resident in the 256 MB
                    # Infinity Cache from launch to launch): the algorithmic bytes are DELIVERED to the CUs, mostly not by HBM --
                    # `delivered_gbs` (it can exceed the HBM peak) beside `achieved` / `frac` from the bytes HBM really moved
                    # (the PMC figure of a launch of this shape, when profiles/ holds one; else null).
                    delivered = round(ach, 1) if ach else None
                    if rb > 512:
                        frac = round(ach / HBM_PEAK_GBS, 4) if ach else None
                        assert frac is None or frac <= 1.0, ("the byte model does not describe the kernel", frac)
                        achieved = delivered
                    else:
                        achieved = round(itraffic / ks_launch / 1e9, 1) if (itraffic and ks_launch > 0) else None
                        frac = round(achieved / HBM_PEAK_GBS, 4) if achieved else None
                        assert frac is None or frac <= 1.0, ("HBM traffic above the HBM peak: the counters or the clock are wrong", frac)
                    # fp32 rows: the MFMA pipe is the nearer roof (2 * 128 flop per (query, row) pair of the launch's candidates)
                    mfma = None
                    if rb > 512 and ks_launch > 0:
                        pairs = info["avg_candidates"] * BATCH * (sum(window) / max(len(window), 1))
                        tf = 2.0 * DIM * pairs / ks_launch / 1e12
                        mfma = {"achieved_tflops": round(tf, 1), "peak_tflops": MFMA_F32_PEAK_TFLOPS, "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
                        assert mfma["frac"] <= 1.0, mfma
                    info["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "mfma_f32": mfma,
                                        "frac": frac,
                                        "frac_basis": "algorithmic bytes" if rb > 512 else "measured HBM traffic (PMC); rows are Infinity-Cache resident",
                                        "delivered_gbs": delivered,
                                        "delivered_over_hbm_peak": round(ach / HBM_PEAK_GBS, 4) if ach else None,
                                        "traffic": itraffic,
                                        "traffic_source": "static: profiles/traffic_ivf_list_scan.json (rocprofv3 --pmc passes, not measured in this run)",
                                        "kernel": "vs::ivf_scan_wide_kernel" if rb < 512 else "vs::ivf_scan_wide_f32_kernel",
                                        "kernel_us": round(ks_launch * 1e6, 2),
                                        "launches": int(okern_n), "batches_per_launch": round(sum(window) / max(len(window), 1), 2),
                                        "algorithmic_bytes_per_launch": int(ib), "row_bytes": rb,
                                        "distinct_rows_per_launch": uniq_rows,
                                        "per_query_pass_bytes": int((4 * DIM + 8) * info["avg_candidates"] * BATCH * max(per_size)),
                                        "note": "one list-major pass per super-batch of 32 batches over the tiled exact int8 copy (128 B of "
                                                "row + 4 B of row term per row); the 132 MB of rows fit the 256 MB Infinity Cache, so "
                                                "repeated launches are served from there and the HBM counters can read below the "
                                                "algorithmic bytes: `frac` is the HBM traffic's, `delivered_over_hbm_peak` the algorithmic bytes' (can pass 1)"
                                                if rb < 512 else "one list-major pass per super-batch of 32 batches over the fp32 rows "
                                                "(IVFIndex.cpp:270-358's arithmetic; 512 B of row + 4 B of norm per row): the 466 MB a "
                                                "super-batch touches do not fit the Infinity Cache, the HBM roof is the binding one"}
                log(f"{tag} nprobe={nprobe}: {info['value']:.0f} QPS, recall@1={info.get('recall_at_1')}, "
                    f"recall@5={info.get('recall_at_5')}, avg candidates={info.get('avg_candidates')}")
                return info

            ivf_info = ivf_leg(NPROBE)
            if not args.no_extras:
                ivf8_info = ivf_leg(8)
            if world == 1 and not args.no_extras:
                # the list scan in the reference's arithmetic: fp32 rows (IVFIndex.cpp:270-358), same index, same pipeline
                ivf.set_precision(1)
                ivf_info["fp32_rows"] = ivf_leg(NPROBE, tag="IVF, fp32 rows,")
                ivf.set_precision(0)
            if world == 1 and not args.no_extras:
                # BASELINE configs[4] on ONE GPU: what a rank of a cluster-sharded job does per launch group.  G shards of
                # this index (vs_ivf_create(..., rank r, world G)) run the sharded call's sliced pipeline as virtual ranks,
                # one after the other, the two collectives replaced by writes into the gathered layout; rank_us = device
                # time (HIP events) of a rank's front + back half for a group of 32 G batches.
                def ev_us(fn, reps=4):
                    fn()
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ts = []
                    for _ in range(3):
                        e0.record(stream)
                        for _ in range(reps):
                            fn()
                        e1.record(stream)
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
                    return sorted(ts)[1]

                def make_ivf(env=None, **kw):
                    env = env or {}
                    old = {k_: os.environ.get(k_) for k_ in env}
                    os.environ.update({k_: str(v_) for k_, v_ in env.items()})
                    try:
                        return pkg.IVFIndex(vectors_reordered=vr, centroids=cents_h, cluster_offsets=off, reorder_to_original=r2o,
                                            device=local_rank, **kw)
                    finally:
                        for k_, v_ in old.items():
                            if v_ is None:
                                os.environ.pop(k_, None)
                            else:
                                os.environ[k_] = v_

                call1 = lambda ix: ix.search_dev_multi(q_ivf.data_ptr(), SI, BATCH, K, NPROBE, iout_i.data_ptr(), iout_d.data_ptr(), sptr)
                us_default = ev_us(lambda: call1(ivf)) / (SI * BATCH / 1024)
                with make_ivf({"VSEARCH_IVF_GROUP": 32, "VSEARCH_IVF_WIDE_LANES": 1}) as ivf32:
                    us_g32 = ev_us(lambda: call1(ivf32)) / (SI * BATCH / 1024)
                want_d = iout_d.clone()
                shard_info = {"metric": "device time per launch group of a rank of a G-way cluster-sharded job, virtual ranks on one GPU "
                                        "(vs_ivf_search_dev_vshards: the collectives replaced by writes into the gathered layout)",
                              "nprobe": NPROBE,
                              "unsharded_us_per_1024_queries": round(us_default, 2),
                              "unsharded_groups_of_32_one_stream_us_per_1024_queries": round(us_g32, 2)}
                for G in (2, 4, 8):
                    shards = [make_ivf(rank=r_, world=G) for r_ in range(G)]
                    nb_g = 32 * G
                    vcall = lambda timed_=False: pkg.IVFIndex.search_dev_vshards(shards, q_ivf.data_ptr(), nb_g, BATCH, K, NPROBE, iout_i.data_ptr(),
                                                                                iout_d.data_ptr(), sptr, timed=timed_)
                    vcall()
                    vcall()
                    torch.cuda.synchronize()
                    assert torch.equal(iout_d[:nb_g * BATCH], want_d[:nb_g * BATCH]), "sliced pipeline differs from the unsharded index"
                    per_rank = np.median(np.array([vcall(True) for _ in range(5)]), axis=0) * 1e3
                    worst = float(per_rank.max())
                    shard_info[f"world{G}"] = {"batches_per_group": nb_g, "rank_us_per_group": [round(float(x), 1) for x in per_rank],
                                               "slowest_rank_us_per_1024_queries": round(worst / G, 2),
                                               "speedup_vs_unsharded": round(us_default / (worst / G), 2),
                                               "speedup_vs_unsharded_groups_of_32_one_stream": round(us_g32 / (worst / G), 2)}
                    log(f"IVF sliced, world {G}: slowest rank {worst:.1f} us per group of {nb_g} batches = {worst / G:.1f} us per 1024 queries "
                        f"({us_default / (worst / G):.2f}x the unsharded index at {us_default:.1f} us, {us_g32 / (worst / G):.2f}x its "
                        f"32-batch one-stream organisation at {us_g32:.1f} us)")
                    for sh_ in shards:
                        sh_.close()
                ivf_info["shard_1ofG"] = shard_info
            if world == 1 and not args.no_extras:
                # the second synthetic distribution (WEAK_MIXTURE: weakly clustered SIFT-range integers): recall well below 1,
                # so that the figure says something; same shapes, same pipeline, its own index and exact ground truth
                wbase = pkg.synth_mixture(n_rows, SEED_BASE, **pkg.WEAK_MIXTURE)
                wq = pkg.synth_mixture(n_queries, SEED_QUERY, **pkg.WEAK_MIXTURE)
                wvr, woff, wr2o, wcents, w_it = pkg.ivf_build(wbase, nlist, max_iter=args.kmeans_iters, seed=42, device=local_rank)
                with pkg.BruteForceIndex(wbase, device=local_rank) as wbf:
                    wgt, _ = wbf.search(wq[:1024], K)
                del wbase
                wsizes = np.diff(woff)
                weak = {"mixture": pkg.WEAK_MIXTURE, "kmeans_iterations": int(w_it), "list_sizes_min_avg_max": [int(wsizes.min()), float(wsizes.mean()), int(wsizes.max())]}
                with pkg.IVFIndex(vectors_reordered=wvr, centroids=wcents, cluster_offsets=woff, reorder_to_original=wr2o, device=local_rank) as wivf:
                    wqd = torch.from_numpy(np.tile(wq, ((SI * BATCH + n_queries - 1) // n_queries, 1))[:SI * BATCH].copy()).to(dev)
                    for npb in (8, NPROBE):
                        wids, _, wtot = wivf.searchBatch(wq[:1024], 1024, K, npb)
                        us = ev_us(lambda: wivf.search_dev_multi(wqd.data_ptr(), SI, BATCH, K, npb, iout_i.data_ptr(), iout_d.data_ptr(), sptr))
                        weak[f"nprobe{npb}"] = {"qps": round(SI * BATCH / us * 1e6, 1),
                                                "recall_at_1": float(np.mean(wids[:, 0] == wgt[:, 0])),
                                                "recall_at_5": float(np.mean([len(set(wids[i]) & set(wgt[i])) / K for i in range(1024)])),
                                                "avg_candidates": wtot / 1024, "nprobe_x_rows_over_nlist": npb * n_rows / nlist}
                        log(f"IVF, weakly clustered set, nprobe={npb}: {weak[f'nprobe{npb}']['qps']:.0f} QPS, recall@1={weak[f'nprobe{npb}']['recall_at_1']:.4f}, "
                            f"recall@5={weak[f'nprobe{npb}']['recall_at_5']:.4f}, avg candidates={wtot / 1024:.0f}")
                ivf_info["weakly_clustered"] = weak
            if world > 1 and not args.no_extras:
                # N > 1, the other way to use N GPUs for an index that fits one of them (1 M rows are 0.2 % of a GPU's HBM):
                # every rank holds the WHOLE index and serves its own share of the batches (queries are independent: no
                # data-path collective; the shares' results are all-gathered at the end of the region).  The cluster-sharded
                # leg above replicates everything but the list scan on every rank, and the scan is a third of a launch group.
                full_ivf = pkg.IVFIndex(vectors_reordered=vr, centroids=cents_h, cluster_offsets=off, reorder_to_original=r2o, device=local_rank)
                share_max = (steps + world - 1) // world
                rep_i = torch.zeros((max(share_max, 1) * BATCH, K), dtype=torch.int32, device=dev)
                rep_d = torch.zeros((max(share_max, 1) * BATCH, K), dtype=torch.float32, device=dev)
                all_i = torch.zeros((world * rep_i.numel(),), dtype=torch.int32, device=dev)
                all_d = torch.zeros((world * rep_d.numel(),), dtype=torch.float32, device=dev)

                def rep_step(i, n):
                    if i != n - 1:
                        return
                    share = len(range(rank, n, world))   # this rank's batches of the n steps
                    done = 0
                    while done < share:
                        nb = min(share - done, n_qbatches)
                        full_ivf.search_dev_multi(q_dev.data_ptr(), nb, BATCH, K, NPROBE, rep_i.data_ptr() + done * BATCH * K * 4,
                                                  rep_d.data_ptr() + done * BATCH * K * 4, sptr)
                        done += nb
                    all_gather(all_i.view(torch.float32), rep_i.view(-1).view(torch.float32))
                    all_gather(all_d, rep_d.view(-1))

                rreg = timed(rep_step, steps, warmup)
                ivf_info["replicas"] = {"metric": "ivf_qps, every rank holds the whole index and serves steps / N batches (results all-gathered)",
                                        "value": round(steps * BATCH / median(rreg), 1), "nprobe": NPROBE, "n_gpus": world}
                log(f"IVF nprobe={NPROBE}, replicas: {ivf_info['replicas']['value']:.0f} QPS on {world} GPUs")
                full_ivf.close()
            if world == 1 and not args.no_extras:
                tm = pkg.Timing()
                ivf.searchBatch(queries, n_queries, K, NPROBE)
                th = []
                for _ in range(5):
                    t = time.perf_counter()
                    ivf.searchBatch(queries, n_queries, K, NPROBE, tm)
                    th.append(time.perf_counter() - t)
                q4 = np.tile(queries, (4, 1))   # main_ivf.cpp runs 10 000 queries per call on SIFT-1M: also a call of 16 384
                ivf.searchBatch(q4, len(q4), K, NPROBE)
                th4 = []
                for _ in range(5):
                    t = time.perf_counter()
                    ivf.searchBatch(q4, len(q4), K, NPROBE)
                    th4.append(time.perf_counter() - t)
                host_info["ivf"] = {"metric": "QPS of vs_ivf_search on host buffers, nprobe=32", "value": round(n_queries / median(th), 1),
                                    "queries_per_call": n_queries, "vs_device_api": round(n_queries / median(th) / ivf_info["value"], 4),
                                    "value_16384_queries_per_call": round(len(q4) / median(th4), 1),
                                    "stage_ms": {"centroid_search": round(tm.centroid_search_ms, 3), "gather": round(tm.gather_ms, 3),
                                                 "fine_search": round(tm.fine_search_ms, 3)}}
                log(f"host API IVF: {host_info['ivf']['value']:.0f} QPS ({n_queries} queries per call), "
                    f"{host_info['ivf']['value_16384_queries_per_call']:.0f} QPS (16384 per call)")
            ivf.close()
        except Exception as e:  # noqa: BLE001
            log(f"IVF legs failed: {e!r}")
            ivf_info = {"error": repr(e), **(ivf_info or {})}

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
                       "collective_every_steps": S if world > 1 else None, "collective": coll_kind},
            "repeats": R_used,
            "ms_per_step_min_max": [round(min(regions) / steps * 1e3, 5), round(max(regions) / steps * 1e3, 5)],
            "roofline": ({"bound": "mfma", "achieved": round(mfma_tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(mfma_tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                          "traffic_source": "static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, summarised in profiles/traffic_bf_scan.json (not measured in this run)",
                          "kernel": "vs::scan_f32s_kernel<2>", "kernel_us": round(kern_avg_s * 1e6, 2),
                          "kernel_us_per_batch": round(kern_per_batch_s * 1e6, 2),
                          "algorithmic_flops_per_launch": int(2 * BATCH * rows_local * DIM * S),
                          "batches_per_launch": S, "batches_per_pass": 2,
                          "algorithmic_bytes_per_launch": algo_bytes, "hbm_algorithmic_gbs": round(achieved, 1),
                          "note": "two batches share one pass over the fp32 rows (same FMA chain per (row, query), same bits): "
                                  "HBM moves half of SURVEY 8(d)'s per-batch bytes (see traffic), the fp32 MFMA pipe binds"}
                         if pair else
                         {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                          "traffic_source": "static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, summarised in profiles/traffic_bf_scan.json (not measured in this run)",
                          "kernel": "vs::scan_f32s_kernel<1>", "kernel_us": round(kern_avg_s * 1e6, 2),
                          "kernel_us_per_batch": round(kern_per_batch_s * 1e6, 2),
                          "algorithmic_bytes_per_launch": algo_bytes,
                          "batches_per_launch": S, "batches_per_pass": 1,
                          "mfma_tflops": round(mfma_tflops, 2)}),
            "cpu_baseline": cpu_info,
            "ivf": ivf_info,
            "ivf_nprobe8": ivf8_info,
            "bf_int8": int8_info,
            "host_api": host_info,
            "q8_runner": q8_info,
        }
        line.update(b1_info)
        print(json.dumps(line), flush=True)
    bf.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
