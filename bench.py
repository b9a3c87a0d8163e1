#!/usr/bin/env python3
"""bench.py — queries/sec @ recall@10 >= 0.95 on synthetic 768-d embeddings (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hnsw10m|hnsw1m|...]

A "step" is one pass of the hot path over one batch of synthetic queries:
    HNSW beam traversal kernel over the rank's index  (+ RCCL all-gather of per-shard top-k and the
    merge kernel when --gpus > 1, mode "shard").
Inputs (corpus rows, graph, queries) are resident in HBM before the timed region.

Without --workload (the driver's command shape) and N = 1: the headline workload runs first (a child process, --headline-only), then
compact legs of every other 1-GPU BASELINE config — configs[1] hnsw1m at ef = 128, configs[2] exhaustive recompute and the
recompute-on graph (throughput batch and batch 64), configs[4] DiskANN R = 32 at 1536-d WITH the hybrid leg — each its own process,
and ONE line comes out: the headline's plus `other_configs` (LEANN_BENCH_BUDGET_S bounds the whole run, default 420 s).
--hybrid (graph workloads): every step is search(fetch_k = 5 k) + BM25 injection + hybrid_rerank on the device (csrc/hybrid.hip).

N > 1: `python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` itself as a child process
(falling back to --mode composite: one process, N devices behind one handle); under a launcher it is one rank per GPU:
  mode shard   (default, north_star): the corpus is partitioned, every rank holds `rows` vectors
               (weak scaling: corpus = N x rows) with its own graph, all ranks search the same query
               batch, per-shard top-k lists are all-gathered over RCCL and merged on every rank — all
               of it inside the library (leann_sharded_attach / leann_sharded_search_batch_device_async,
               csrc/shard.hip); torch only hands rank 0's RCCL id to the other ranks.
               value = batch * K / t: END-TO-END queries/s over the whole N x rows corpus (every query
               is answered once, by all shards together).  Weak scaling therefore means: the corpus
               grows N-fold at (ideally) constant queries/s.  The per-shard search rate N * batch * K / t
               rides along as "shard_searches_per_s", and a second barrier-bracketed region of the same K steps WITHOUT the
               collective — every rank answering different query batches from its own index, i.e. N replicas of a `rows`-row
               corpus ("report both shard and replica curves") — as "replica_mode".
  mode replica every rank holds the whole `rows`-vector index and takes different query batches;
               no data-path collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

torch = None  # imported by main() on the paths that touch the GPU (device memory, streams, events, torch.distributed / RCCL: plumbing only);
              # the launcher-less parents (default run, --gpus N) only start children and need neither torch nor a GPU

WORKLOADS = {
    # name: rows per GPU, dims, M, efc, ef, synthetic params
    # BASELINE metric config.  ef_construction 200: measured on 10M rows (final builder and kernel), recall@10 >= 0.955 needs ef = 56
    # (0.9566, 1.15 M QPS, build 27 s); an efc = 320 graph (--efc 320) still needs ef = 56 (0.9635, 1.11 M QPS, build 41 s)
    "hnsw10m": dict(rows=10_000_000, d=768, M=32, efc=200, ef=128),
    "hnsw1m": dict(rows=1_000_000, d=768, M=32, efc=128, ef=128),     # BASELINE configs[1]
    "hnsw100m_shard8": dict(rows=12_500_000, d=768, M=32, efc=200, ef=128),  # BASELINE configs[3]: 100M x 768 = 8 shards of 12.5M (--gpus 8)
    "hnsw100k": dict(rows=100_000, d=768, M=32, efc=128, ef=128),     # quick check
    "hnsw4m": dict(rows=4_000_000, d=768, M=32, efc=200, ef=128),     # rehearsal size: 8 shards of it fit ONE GPU (--mode composite, LEANN_BENCH_COMPOSITE_DEVICES=0,0,...)
    # configs[4] search leg: DiskANN/Vamana, 1536-d.  R = 64 (R = 32 reaches recall 0.98 at 1M but only 0.60 at 10M with beam 128; 0.81 with a
    # second build pass, LEANN_VAMANA_PASSES=2: profiles/r02_vamana10m1536_r32_bench.json)
    "vamana10m1536": dict(rows=10_000_000, d=1536, M=64, efc=128, ef=128, backend=1),
    "vamana1m1536": dict(rows=1_000_000, d=1536, M=64, efc=128, ef=128, backend=1),
    "vamana10m1536_r32": dict(rows=10_000_000, d=1536, M=32, efc=128, ef=128, backend=1),  # configs[4]'s other degree (max_degree = 32, beam 128)
    # configs[2]: recompute-on (no stored vectors), batch-64 queries: features [rows x 256] bf16 + W [256 x 768] bf16
    "recompute10m": dict(rows=10_000_000, d=768, h=256, kind="recompute", batch=64),
    "recompute1m": dict(rows=1_000_000, d=768, h=256, kind="recompute", batch=64),
    # configs[2] with the LEANN idea proper: HNSW whose distances are recomputed from the bf16 features (no stored vectors)
    "recompute10m_graph": dict(rows=10_000_000, d=768, h=256, M=32, efc=200, ef=128, kind="recompute_graph"),
    "recompute1m_graph": dict(rows=1_000_000, d=768, h=256, M=32, efc=200, ef=128, kind="recompute_graph"),
    # SURVEY 8(d) "brute-force scan": exact search over stored f32 rows (RecomputeSearcher's dot + sort + take over materialised
    # embeddings, recompute.rs:96-109; also every workload's ground truth), batch 64
    "scan10m": dict(rows=10_000_000, d=768, kind="scan", batch=64),
}
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = f32 vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16
SEED, GEN_R, GEN_CLUSTERS, GEN_SIGMA = 0x5EED0001, 64, 4096, 1.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_REAL_STDOUT = None


def quiet_stdout():
    """stdout carries exactly ONE JSON line (the driver's contract).  RCCL prints a version banner to the C-level stdout when a
    communicator is made, so fd 1 is pointed at stderr for the whole run and the line goes out through the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    line = (json.dumps(out) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def host_cores():
    """threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def bench_recompute(args, wl, la, L, chk, dev, local_rank, world, rank, dist, log):
    """configs[2]: brute-force search with on-the-fly embedding recompute (src/index/recompute.rs:52-123).
    One step = one batch of 64 queries against all `rows` passages: encode GEMM (bf16 MFMA) + scoring
    GEMM (f32 MFMA) + top-k.  N > 1: passages are partitioned, per-shard top-k all-gathered and merged."""
    from leann_rs_amd.shard import exchange_topk, _hip_merge as hip_merge
    rows, d, h = wl["rows"], wl["d"], wl["h"]
    B = wl["batch"] if args.batch == 16384 else args.batch
    k = args.k
    ld = (d + 3) // 4 * 4
    steps, warmup = args.steps, args.warmup  # exactly K timed steps after W warm-up steps
    shard = world > 1
    row0 = rank * rows if shard else 0
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    F = torch.empty((rows, h), dtype=torch.int16, device=dev)
    W = torch.empty((h, d), dtype=torch.int16, device=dev)
    Fq = torch.empty((B * 4, h), dtype=torch.int16, device=dev)
    Q = torch.empty((B * 4, ld), dtype=torch.float32, device=dev)
    chk(L.leann_synth_features_device(SEED, h, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, row0, rows, F.data_ptr(), sp))
    chk(L.leann_synth_weights_device(SEED, h, d, W.data_ptr(), sp))
    chk(L.leann_synth_features_device(SEED, h, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 1, 0, B * 4, Fq.data_ptr(), sp))
    stream.synchronize()
    r, rq = C.c_void_p(), C.c_void_p()
    chk(L.leann_recompute_create(F.data_ptr(), rows, h, W.data_ptr(), d, local_rank, row0, C.byref(r)))
    chk(L.leann_recompute_create(Fq.data_ptr(), B * 4, h, W.data_ptr(), d, local_rank, 0, C.byref(rq)))
    chk(L.leann_recompute_encode_device(rq, 0, B * 4, Q.data_ptr(), sp))  # queries = embeddings of query-side features
    stream.synchronize()
    keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    scores = torch.empty((B, k), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    ms = (C.c_float * 3)()
    tm = []
    allow, n_allowed = None, rows
    if args.filter_selectivity > 0:  # side experiment: the early filter of recompute.rs:62-79 (only the allowed passages are embedded)
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x5EED0004 + rank)
        bits = torch.zeros(((rows + 7) // 8) * 8, dtype=torch.uint8, device=dev)
        bits[:rows] = (torch.rand(rows, device=dev, generator=gen) < args.filter_selectivity).to(torch.uint8)
        n_allowed = int(bits.sum(dtype=torch.int64).item())
        wts = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
        allow = (bits.view(-1, 8) * wts).sum(1, dtype=torch.int32).to(torch.uint8).contiguous()
        del bits
        torch.cuda.synchronize()

    def step(i):
        qptr = Q.data_ptr() + (i % 4) * B * ld * 4
        chk(L.leann_recompute_search_batch_device(r, qptr, B, k, allow.data_ptr() if allow is not None else None, keys.data_ptr(),
                                                  scores.data_ptr(), counts.data_ptr(), sp))
        L.leann_recompute_last_timing(r, ms)
        tm.append((ms[0], ms[1], ms[2]))
        if shard:
            with torch.cuda.stream(stream):
                gk, gs, gc = exchange_topk(keys, scores, counts, world)
                return hip_merge(gk, gs, gc, k, True, stream.cuda_stream)[0]
        return keys

    # recall@10 measured in-run, not assumed: the reference's literal order — materialise every embedding (recompute.rs:86-93), then
    # dot + sort + take (:96-109) — as leann_recompute_encode_device + leann_scan_topk_device over the same rows, against the fused search
    recall, recall_note = None, "not measured (sharded / filtered run or too many rows to materialise)"
    if not shard and allow is None and rows * ld * 4 <= 64 << 30:
        E = torch.empty((rows, ld), dtype=torch.float32, device=dev)
        for r0 in range(0, rows, 4 << 20):
            chk(L.leann_recompute_encode_device(r, r0, min(4 << 20, rows - r0), E.data_ptr() + r0 * ld * 4, sp))
        tk_, ts_, tc_ = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
                         torch.empty((B,), dtype=torch.int32, device=dev))
        chk(L.leann_scan_topk_device(E.data_ptr(), rows, d, ld, Q.data_ptr(), B, k, None, 0, tk_.data_ptr(), ts_.data_ptr(), tc_.data_ptr(), sp))
        step(0)
        stream.synchronize()
        a_, b_ = keys.cpu().numpy(), tk_.cpu().numpy()
        recall = float(np.mean([len(set(a_[i].tolist()) & set(b_[i].tolist())) / k for i in range(B)]))
        recall_note = (f"fused recompute search vs exact scan over the materialised embeddings of all {rows} passages, {B} queries "
                       f"(max |score difference| {float((scores - ts_).abs().max()):.2e})")
        del E
        torch.cuda.empty_cache()
        log("recall@%d = %.4f (%s)" % (k, recall, recall_note))
    for w in range(warmup):
        step(w)
    stream.synchronize()
    tm.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        step(s_)
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    enc_ms, sc_ms, tk_ms = [float(np.mean([t[i] for t in tm])) for i in range(3)]
    # fused kernel: encode GEMM F[N x h] W[h x d] + feature-space scoring F (W Q^T) as three bf16 pieces (hi/lo/lo2)
    enc_flops, sc_flops = 2.0 * rows * h * d, 3 * 2.0 * rows * h * ((B + 31) // 32 * 32)  # one encode shared by the batch; a G tile per 32 queries
    fused_tf = (enc_flops + sc_flops) / (enc_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "unit": "TFLOP/s", "traffic": None,
            "kernel": "fused_fstat_kernel<16,true> (bf16 MFMA 32x32x16: encode GEMM + row norms + feature-space scoring + candidate emission, fused)",
            "achieved": fused_tf, "peak": BF16_MFMA_PEAK_TFLOPS, "frac": fused_tf / BF16_MFMA_PEAK_TFLOPS,
            "fused_encode_score_ms": enc_ms, "topk_ms": tk_ms,
            "mfma_flops_per_step": enc_flops + sc_flops,
            "algorithmic_flops_per_step": enc_flops + 2.0 * rows * d * B,  # what the reference's order (embed, then d-dim dots) costs
            "algorithmic_bytes_per_step": rows * h * 2 + h * d * 2,
            "hbm_gbps_features": rows * h * 2 / (enc_ms * 1e-3) / 1e9}
    # shard mode: every query is answered once by all shards together -> end-to-end queries/s = B * K / t
    out = {"metric": "queries/sec @ recall@10>=0.95", "value": B * steps / elapsed, "unit": "queries/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16 encode / f32 score", "data": "synthetic", "recall_at_10": recall, "recall_note": recall_note,
           "config": {"workload": f"{args.workload}: recompute-on (no stored vectors), {rows} passages x {h} bf16 features per GPU, "
                                  f"encoder W[{h}x{d}] bf16, embedding = l2norm(W^T f), batch {B} queries/step, exhaustive scan (exact)",
                      "rows_per_gpu": rows, "dims": d, "feature_dim": h, "batch": B, "top_k": k,
                      "parallelism": "single" if world == 1 else f"shard{world}+rccl_allgather"},
           "roofline": roof}
    if allow is not None:
        out["config"]["filter_selectivity"] = args.filter_selectivity
        out["config"]["filter_note"] = (f"side experiment: early filter (recompute.rs:62-79), {n_allowed} of {rows} passages allowed; "
                                        "the allowed rows are compacted and only they are embedded (fused_fstat_kernel<16,false,true>); "
                                        "roofline flops count the allowed rows")
        roof["allowed_rows"] = n_allowed
        e2, s2 = 2.0 * n_allowed * h * d, 3 * 2.0 * n_allowed * h * ((B + 31) // 32 * 32)
        roof.update({"achieved": (e2 + s2) / (enc_ms * 1e-3) / 1e12, "frac": (e2 + s2) / (enc_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                     "mfma_flops_per_step": e2 + s2, "kernel": "fused_fstat_kernel<16,false,true> (row list)"})
    if shard:
        out["shard_searches_per_s"] = B * steps * world / elapsed
    # ---- CPU baseline: the oracle's literal recompute.rs:86-109 (embed every passage, dot, sort) on a bounded sample -------
    if world == 1 and rank == 0 and not args.no_cpu_baseline and allow is None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            from concurrent.futures import ThreadPoolExecutor
            cores = host_cores()
            ns = min(rows, 10_000_000)  # the reference re-embeds ALL passages for every query: one whole query is timed (~10 s)
            Fh = F[:ns].cpu().numpy().view(np.uint16)
            Wh = W.cpu().numpy().view(np.uint16)
            qh = Q[0, :d].contiguous().cpu().numpy()
            bounds = [(ns * t // cores, ns * (t + 1) // cores) for t in range(cores)]

            def part(b):  # ctypes releases the GIL: one slice per host thread
                E = po.recompute_encode(Fh[b[0]:b[1]], Wh)
                return po.scan_topk(E, qh, k, mode=0)
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(part, bounds[:cores]))  # touch
                t0 = time.perf_counter()
                list(ex.map(part, bounds))
                cpu_s = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": 1.0 / (cpu_s * rows / ns), "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"one query over {ns} of the {rows} passages (oracle/oracle.c: orc_recompute_encode + orc_scan_topk, the literal "
                          f"embed-everything-then-dot order of recompute.rs:86-109), {cores} host threads, {cpu_s:.2f} s; value = extrapolated "
                          f"to all {rows} passages per query",
            }
            log(f"cpu baseline: {ns} passages of one query in {cpu_s:.2f} s on {cores} threads -> {out['cpu_baseline']['value']:.4f} q/s at {rows} passages")
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_scan(args, wl, la, L, chk, dev, local_rank, world, rank, dist, log):
    """Exact scan: one step = one batch of 64 queries against all `rows` stored f32 rows (score_mfma_kernel on the f32 matrix
    cores, k-ordered chains = bit-exact with the oracle's sequential fmaf dot, + segment top-k).  Replicas only for N > 1."""
    rows, d = wl["rows"], wl["d"]
    B = wl["batch"] if args.batch == 16384 else args.batch
    k, ld = args.k, (d + 3) // 4 * 4
    steps, warmup = args.steps, args.warmup
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)
    X = torch.empty((rows, ld), dtype=torch.float32, device=dev)
    Q = torch.empty((B * 4, ld), dtype=torch.float32, device=dev)
    chk(L.leann_synth_rows_device(SEED, d, ld, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, 0, rows, X.data_ptr(), sp))
    chk(L.leann_synth_rows_device(SEED, d, ld, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 1, rank * B * 4, B * 4, Q.data_ptr(), sp))
    stream.synchronize()
    keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    scores = torch.empty((B, k), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)

    def step(i):
        chk(L.leann_scan_topk_device(X.data_ptr(), rows, d, ld, Q.data_ptr() + (i % 4) * B * ld * 4, B, k, None, 0, keys.data_ptr(),
                                     scores.data_ptr(), counts.data_ptr(), sp))
    for w in range(warmup):
        step(w)
    stream.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        step(s_)
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    step_s = elapsed / steps
    nbytes, flops = rows * d * 4.0, 2.0 * rows * d * B
    out = {"metric": "queries/sec @ recall@10>=0.95", "value": B * steps * world / elapsed, "unit": "queries/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "recall_at_10": 1.0, "recall_note": "this workload IS the exact scan (every other workload's ground truth)",
           "config": {"workload": f"{args.workload}: exact scan of {rows} x {d} stored f32 rows per GPU, batch {B} queries/step, top-{k}",
                      "rows_per_gpu": rows, "dims": d, "batch": B, "top_k": k, "parallelism": "single" if world == 1 else f"replica{world}"},
           # bit-exact f32 chains run on v_mfma_f32_32x32x2_f32: at batch 64 the matrix-core floor (flops / 157.3 TFLOP/s) is 1.6x the
           # HBM floor (bytes / 8 TB/s), so the kernel is priced against the f32 MFMA peak; the HBM figures ride along
           "roofline": {"bound": "mfma", "achieved": flops / step_s / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / step_s / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "kernel": "score_mfma_kernel<false,true> (+ fold_candidates_kernel / topk_scores_kernel, whole step)",
                        "algorithmic_flops_per_step": flops, "algorithmic_bytes_per_step": nbytes,
                        "hbm_gbps": nbytes / step_s / 1e9, "hbm_frac": nbytes / step_s / 1e9 / HBM_PEAK_GBS}}
    if (2.0 * B) / 4.0 < F32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):  # small batches: HBM-bound
        r_ = out["roofline"]
        r_.update({"bound": "hbm", "achieved": r_["hbm_gbps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_["hbm_frac"]})
    if world == 1 and rank == 0 and not args.no_cpu_baseline:  # CPU baseline: the oracle's literal dot + stable sort + take on the host cores
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            from concurrent.futures import ThreadPoolExecutor
            cores = host_cores()
            ns = min(rows, 10_000_000)  # ~5 s of CPU work per thread at 10M x 768
            Xh = X[:ns, :d].contiguous().cpu().numpy()
            Qh = Q[:cores, :d].contiguous().cpu().numpy()
            with ThreadPoolExecutor(cores) as ex:  # one query per host thread (ctypes releases the GIL)
                list(ex.map(lambda i: po.scan_topk(Xh[:65536], Qh[i], k, mode=0), range(cores)))
                t0 = time.perf_counter()
                list(ex.map(lambda i: po.scan_topk(Xh, Qh[i], k, mode=0), range(cores)))
                cpu_s = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": cores / (cpu_s * rows / ns), "unit": "queries/s", "cores": cores, "kind": "port",
                                   "sample": f"{cores} queries (one per host thread) over {ns} of the {rows} rows (oracle/oracle.c:orc_scan_topk, the "
                                             f"sequential dot + stable sort + take of recompute.rs:96-109), {cpu_s:.2f} s; value = extrapolated to all rows"}
            log(f"cpu baseline: {out['cpu_baseline']['value']:.2f} q/s")
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_command(n, port, argv):
    """the command line `python bench.py --gpus N` turns itself into: one rank per GPU under torch.distributed.run"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def composite_command(argv):
    """fallback: ONE process driving all N devices through the library's composite handle (leann_sharded_build_device)"""
    out, skip = [], False
    for a in argv:  # drop any --mode the caller gave
        if skip:
            skip = False
        elif a == "--mode":
            skip = True
        elif not a.startswith("--mode="):
            out.append(a)
    return [sys.executable, os.path.abspath(__file__)] + out + ["--mode", "composite"]


def self_launch(args, argv, run=None):
    """Start the N ranks as a child process (never an exec: see the GPU-box rules), relay the one JSON line, return the exit code.
    If the launcher cannot produce a line (torch.distributed.run missing, RCCL bootstrap failure) and the workload is a graph
    search, the one-process composite handle is tried next; `config.parallelism` says which one ran."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env["LEANN_BENCH_SELF_LAUNCHED"] = "1"
    def run_with_deadline(cmd, limit_s):
        # a rank that hangs (e.g. an RCCL bootstrap that never completes) must not eat the whole run: past the deadline the child's own
        # process group — the one started here, nothing else — is ended and the fallback gets its turn
        import signal
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            print(f"[bench] {' '.join(cmd[:4])} ... gave no result within {limit_s:.0f} s: ending its process group", file=sys.stderr, flush=True)
            try:
                os.killpg(p.pid, signal.SIGTERM)
                out, _ = p.communicate(timeout=20)
            except Exception:  # noqa: BLE001
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except Exception:  # noqa: BLE001
                    pass
                out, _ = p.communicate()
            return subprocess.CompletedProcess(cmd, -9, out, None)
        return subprocess.CompletedProcess(cmd, p.returncode, out, None)
    limit_ranks = float(os.environ.get("LEANN_BENCH_LAUNCH_TIMEOUT_S", "360"))
    injected = run is not None
    run = run or (lambda cmd: run_with_deadline(cmd, limit_ranks if "torch.distributed.run" in cmd else 3600.0))

    def relay(proc):
        lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if proc.returncode == 0 and lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
            return True
        return False

    cmd = launch_command(args.gpus, port, argv)
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = run(cmd)
    if relay(proc):
        return 0
    kind = WORKLOADS[args.workload].get("kind")
    if kind in (None, "recompute_graph") and args.mode == "shard" and not os.environ.get("LEANN_BENCH_NO_COMPOSITE_FALLBACK"):
        cmd2 = composite_command(argv)
        print("[bench] the ranks exited with code %d without a result; trying the one-process composite handle: %s"
              % (proc.returncode, " ".join(cmd2)), file=sys.stderr, flush=True)
        proc2 = run(cmd2)
        if relay(proc2):
            return 0
        return proc2.returncode or proc.returncode or 1
    return proc.returncode or 1


# compact legs of the other 1-GPU BASELINE configs: (name, argv, rough seconds on an MI355X box incl. corpus generation and index build)
OTHER_CONFIGS = [
    ("hnsw1m_ef128", ["--workload", "hnsw1m", "--ef", "128", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-latency"], 25),
    ("recompute10m_exhaustive_batch64", ["--workload", "recompute10m", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], 40),
    ("recompute10m_graph", ["--workload", "recompute10m_graph", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-latency",
                            "--small-batch", "64"], 70),
    ("vamana10m1536_r32_hybrid", ["--workload", "vamana10m1536_r32", "--hybrid", "--steps", "10", "--warmup", "2", "--no-latency",
                                  "--cpu-queries", "1024"], 170),
]


def summarise_leg(j):
    """what `other_configs` keeps of a leg's own JSON line"""
    r = j.get("roofline", {})
    o = {"value": j.get("value"), "unit": j.get("unit"), "recall_at_10": j.get("recall_at_10"), "ms_per_step": j.get("ms_per_step"),
         "steps": j.get("steps"), "ef_search": j.get("config", {}).get("ef_search"),
         "index_build_s": j.get("config", {}).get("index_build_s"),
         "roofline": {"bound": r.get("bound"), "frac": r.get("frac"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit")},
         "kernel": (r.get("kernel") or "").split(" (")[0], "kernel_avg_ms": r.get("kernel_avg_ms", r.get("fused_encode_score_ms"))}
    # numbers only: the explanatory strings live in the leg's own line (`command` reproduces it); the one line stays a few KB
    if "small_batch" in j:
        o["small_batch"] = {k: j["small_batch"].get(k) for k in ("batch", "ms_per_call", "value", "hbm_frac", "streams_in_flight", "value_concurrent", "value_concurrent_hipgraph")}
    if "hybrid" in j:
        h = j["hybrid"]
        o["hybrid"] = {k: h.get(k) for k in ("fetch_k", "alpha", "compat_polarity", "rerank_avg_ms", "traversal_avg_ms")}
        o["hybrid"]["rerank_on"] = "device"
        if "rerank_parity" in h:
            o["hybrid"]["parity_sample"] = h["rerank_parity"].get("sample")
            o["hybrid"]["parity_mismatches"] = h["rerank_parity"].get("mismatching_queries")
    if "cpu_baseline" in j:
        c = j["cpu_baseline"]
        o["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "gpu_results_bit_identical_on_sample") if k in c}
    return o


def run_all_configs(args, argv, run=None):
    """Default run: the headline workload (a child with --headline-only and the caller's flags), then compact legs of configs[1], [2]
    and [4] while the time budget lasts (LEANN_BENCH_BUDGET_S, default 420 s), one JSON line in the end: the headline's line plus
    `other_configs`.  Returns the exit code (the headline's)."""
    import subprocess
    me = os.path.abspath(__file__)
    t_start = time.time()
    budget = float(os.environ.get("LEANN_BENCH_BUDGET_S", "420"))
    run = run or (lambda cmd: subprocess.run(cmd, stdout=subprocess.PIPE))

    def leg(cmd):
        p = run(cmd)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        return p.returncode, (json.loads(lines[-1]) if lines else None)

    rc, head = leg([sys.executable, me] + list(argv) + ["--headline-only"])
    if rc != 0 or head is None:
        return rc or 1
    others = {}
    for name, extra, est in OTHER_CONFIGS:
        if os.environ.get("LEANN_BENCH_SKIP_OTHERS"):
            break
        if time.time() - t_start + est > budget:
            others[name] = {"skipped": f"time budget ({budget:.0f} s for the whole default run; this leg needs ~{est} s): run `python bench.py {' '.join(extra)}`"}
            continue
        t0 = time.time()
        print(f"[bench] other config {name}: {' '.join(extra)}", file=sys.stderr, flush=True)
        rc2, j = leg([sys.executable, me] + extra)
        others[name] = summarise_leg(j) if (rc2 == 0 and j) else {"failed": f"exit code {rc2}"}
        others[name]["leg_wall_s"] = round(time.time() - t0, 1)
        others[name]["command"] = "python bench.py " + " ".join(extra)
    head["other_configs"] = others
    head["other_configs_note"] = ("compact legs of the other 1-GPU BASELINE configs, each its own process after the headline (same contract: inputs "
                                  "resident in HBM, barrier-free single-GPU timing of exactly `steps` steps); --headline-only skips them")
    sys.stdout.write(json.dumps(head) + "\n")
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hnsw10m", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="shard", choices=["shard", "replica", "composite"],
                    help="N > 1: shard = one rank per GPU, RCCL all-gather inside the library (default); replica = whole index per GPU; "
                         "composite = ONE process, N devices behind one handle (leann_sharded_build_device: peer copies instead of RCCL)")
    ap.add_argument("--batch", type=int, default=16384, help="queries per step (per rank in replica mode)")
    ap.add_argument("--ef", default="auto", help="beam width: 'auto' (default) = smallest of 40..128 whose measured "
                    "recall@10 is >= 0.955 on the recall queries (the metric's operating point is defined by recall >= 0.95); "
                    "a number = fixed (BASELINE configs[1] names ef=128); 0 = the workload's named value")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--recall-queries", type=int, default=4000, help="queries with exact ground truth: the first half picks --ef auto, the second half reports recall")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-query latency / 64-caller section")
    ap.add_argument("--cpu-queries", type=int, default=16384)
    ap.add_argument("--filter-selectivity", type=float, default=0.0,
                    help="side experiment (not the headline metric): metadata-filtered search with a seeded random allow-bitmap "
                         "of this density evaluated inside the traversal; recall is measured against the exact FILTERED top-k")
    ap.add_argument("--sigma", type=float, default=None,
                    help="per-point noise of the synthetic corpus (default 1.0; SURVEY.md §8d wrote 0.15 — tighter clusters; the value used is in config.workload)")
    ap.add_argument("--clusters", type=int, default=None,
                    help="number of cluster centres of the synthetic corpus (default 4096).  65536 puts (almost) every query of a 16 384-query "
                         "launch into a cluster of its own: bounds what queries sharing a neighbourhood gain from L2 / Infinity Cache")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1, modes shard / composite: STRONG scaling — the workload's rows are the WHOLE corpus, split N ways (rows / N per GPU), instead of "
                         "rows per GPU (weak scaling, the default).  SURVEY.md §8e: every GPU still processes every query, so expect little gain.")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the workload itself.  Without it (and without --workload) the default run measures the headline AND compact legs "
                         "of every other 1-GPU BASELINE config, each in a child process, and reports them under `other_configs` of the one line")
    ap.add_argument("--small-batch", type=int, default=0,
                    help="graph workloads: also time calls of this many queries (BASELINE configs[2] names batch-64) on the same index -> `small_batch`")
    ap.add_argument("--hybrid", action="store_true",
                    help="BASELINE configs[4] as a whole (graph workloads): every step searches with fetch_k = 5 k (searcher.rs:129-133), injects "
                         "the BM25-only hits of a seeded sparse BM25 score vector per query (searcher.rs:154-165) and runs hybrid_rerank "
                         "(bm25.rs:135-170) on the device (csrc/hybrid.hip), inside the timed region; a >= 1000-query sample of the reranked "
                         "(id, score) lists is compared with oracle/searcher_oracle.py")
    ap.add_argument("--hybrid-alpha", type=float, default=0.7, help="weight of the vector term (searcher.rs:47)")
    ap.add_argument("--compat-polarity", default="true", choices=["true", "false"],
                    help="true: the reference's blend of DISTANCES (SURVEY.md N1); false: corrected, 1 - dist")
    ap.add_argument("--efc", type=int, default=0, help="override the workload's ef_construction (graph build quality; 0 = the workload's value)")
    ap.add_argument("--filter-exact", action="store_true",
                    help="with --filter-selectivity: answer the filtered queries exactly (allowed rows compacted + f32 MFMA scan, "
                         "leann_backend_search_filtered_exact_batch_device) instead of walking the graph")
    args = ap.parse_args()

    global GEN_SIGMA, GEN_CLUSTERS
    if args.sigma is not None:
        GEN_SIGMA = float(args.sigma)
    if args.clusters is not None:
        GEN_CLUSTERS = int(args.clusters)
    explicit_workload = any(x == "--workload" or x.startswith("--workload=") for x in sys.argv[1:])
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not explicit_workload and not args.headline_only:
        # the driver's command shape (`python bench.py --gpus 1 --steps K --warmup W`): headline + every other 1-GPU BASELINE config.
        # This parent never touches the GPU; every leg is a child process of its own.
        sys.exit(run_all_configs(args, sys.argv[1:]))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mode != "composite":
        # `python bench.py --gpus N` without a launcher (the driver's command shape): this process has not touched the GPU yet, so
        # it starts the N ranks itself as a CHILD process and relays rank 0's JSON line
        sys.exit(self_launch(args, sys.argv[1:]))
    global torch
    import torch as _torch
    torch = _torch
    quiet_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.mode == "composite":
        world, rank, local_rank = 1, 0, 0  # one process, args.gpus devices behind one handle (csrc/shard.hip)
    elif world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists in the product path)")
    gloo_rehearsal = os.environ.get("LEANN_BENCH_DIST_BACKEND") == "gloo"
    if torch.cuda.device_count() < args.gpus and not gloo_rehearsal and not (args.mode == "composite" and os.environ.get("LEANN_BENCH_COMPOSITE_DEVICES")):
        sys.exit(f"bench.py: {args.gpus} GPUs requested, {torch.cuda.device_count()} visible (rank {rank})")
    if torch.cuda.device_count() < world and gloo_rehearsal:
        local_rank = 0  # rehearsal of the N > 1 path on a one-GPU box: ranks share device 0, exchange over gloo
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_shard = os.environ.get("LEANN_BENCH_FORCE_SHARD") == "1" and "RANK" in os.environ  # rehearsal: the N > 1 code path with one rank
    if world > 1 or force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend_name = os.environ.get("LEANN_BENCH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend_name)

    import leann_rs_amd as la  # after torch: binds to the HIP runtime torch already loaded
    from leann_rs_amd.shard import exchange_topk, _hip_merge as hip_merge
    L, chk = la.lib(), la._native.check

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    wl = dict(WORKLOADS[args.workload])
    if wl.get("kind") == "scan":
        return bench_scan(args, wl, la, L, chk, dev, local_rank, world, rank, dist, log)
    if wl.get("kind") == "recompute":
        return bench_recompute(args, wl, la, L, chk, dev, local_rank, world, rank, dist, log)
    rows, d, M, efc = wl["rows"], wl["d"], wl["M"], (args.efc or wl["efc"])
    strong = bool(args.strong) and args.gpus > 1 and args.mode in ("shard", "composite")
    if strong:
        rows = (rows // args.gpus) & ~63  # the workload's rows are the whole corpus (multiples of 64 per shard: bitmaps slice at bytes)
    ef_auto = str(args.ef).lower() == "auto"
    ef = wl["ef"] if ef_auto else (int(args.ef) or wl["ef"])
    backend = wl.get("backend", 0)
    k, B = args.k, args.batch
    ld = (d + 3) // 4 * 4
    shard = (world > 1 or force_shard) and args.mode == "shard"
    composite = args.mode == "composite" and args.gpus > 1
    G = args.gpus if composite else 1
    n_gpus = args.gpus if composite else world
    row0 = rank * rows if shard else 0
    corpus_total = rows * world if shard else rows * G
    stream = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(stream.cuda_stream)

    # ---- synthetic corpus shard + index, all in HBM ---------------------------------------------
    rgraph = wl.get("kind") == "recompute_graph"
    n_pool = max(1, min(args.steps, 8))
    q_first = 0 if (shard or world == 1) else rank * n_pool * B
    t0 = time.time()
    Xs, keep = [], []
    if composite:
        # ONE process, G devices: shard g's rows (or features) live on device g, one sub-index with its own graph per device, all of them
        # behind one composite leann_backend handle whose searches fan out, peer-copy the per-shard lists to device 0 and merge there
        from leann_rs_amd.backend import ShardedIndex
        subs = []
        # (LEANN_BENCH_COMPOSITE_DEVICES="0,0": rehearsal of this path on a one-GPU box — several shards on one device)
        devs = [int(x) for x in os.environ.get("LEANN_BENCH_COMPOSITE_DEVICES", ",".join(str(g) for g in range(G))).split(",")]
        assert len(devs) == G, "LEANN_BENCH_COMPOSITE_DEVICES must name one device per shard"
        for g in range(G):
            dg = torch.device("cuda", devs[g])
            with torch.cuda.device(dg):
                if rgraph:
                    hfeat = wl["h"]
                    Fg = torch.empty((rows, hfeat), dtype=torch.int16, device=dg)
                    Wg = torch.empty((hfeat, d), dtype=torch.int16, device=dg)
                    chk(L.leann_synth_features_device(SEED, hfeat, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, g * rows, rows, Fg.data_ptr(), None))
                    chk(L.leann_synth_weights_device(SEED, hfeat, d, Wg.data_ptr(), None))
                    torch.cuda.synchronize(dg)
                    rcg = C.c_void_p()
                    chk(L.leann_recompute_create(Fg.data_ptr(), rows, hfeat, Wg.data_ptr(), d, devs[g], g * rows, C.byref(rcg)))
                    keep += [Fg, Wg, rcg]
                    if g == 0:
                        F, Wt, rc_h = Fg, Wg, rcg
                else:
                    Xg = torch.empty((rows, ld), dtype=torch.float32, device=dg)
                    chk(L.leann_synth_rows_device(SEED, d, ld, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, g * rows, rows, Xg.data_ptr(), None))
                    torch.cuda.synchronize(dg)
                    Xs.append(Xg)
        log(f"composite: rows of {G} shards generated in {time.time() - t0:.1f}s")
        t0 = time.time()
        if rgraph:
            for g in range(G):
                hb = C.c_void_p()
                chk(L.leann_recompute_build_index(keep[3 * g + 2], backend, M, efc, C.byref(hb)))
                subs.append(la.BackendSearcher(hb, backend))
            searcher = ShardedIndex.from_searchers(subs, take_ownership=True).as_backend(backend)
        else:
            searcher = ShardedIndex.build_device(backend, [x.data_ptr() for x in Xs], [rows] * G, d, ld, M, efc, devs,
                                                 keep=tuple(Xs)).as_backend(backend)
        X = Xs[0] if Xs else None
        torch.cuda.set_device(0)
    elif rgraph:
        # no stored vectors: bf16 features + encoder weights; the graph is built on transient embeddings
        hfeat = wl["h"]
        F = torch.empty((rows, hfeat), dtype=torch.int16, device=dev)
        Wt = torch.empty((hfeat, d), dtype=torch.int16, device=dev)
        chk(L.leann_synth_features_device(SEED, hfeat, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, row0, rows, F.data_ptr(), sp))
        chk(L.leann_synth_weights_device(SEED, hfeat, d, Wt.data_ptr(), sp))
        stream.synchronize()
        rc_h = C.c_void_p()
        chk(L.leann_recompute_create(F.data_ptr(), rows, hfeat, Wt.data_ptr(), d, local_rank, row0, C.byref(rc_h)))
        log(f"rank {rank}: features [{rows} x {hfeat}] bf16 generated in {time.time() - t0:.1f}s")
        t0 = time.time()
        hb = C.c_void_p()
        chk(L.leann_recompute_build_index(rc_h, backend, M, efc, C.byref(hb)))
        searcher = la.BackendSearcher(hb, backend)
        X = None
    else:
        X = torch.empty((rows, ld), dtype=torch.float32, device=dev)
        with torch.cuda.stream(stream):
            chk(L.leann_synth_rows_device(SEED, d, ld, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 0, row0, rows, X.data_ptr(), sp))
        stream.synchronize()
        log(f"rank {rank}: corpus rows [{row0}, {row0 + rows}) x {d} generated in {time.time() - t0:.1f}s")
        t0 = time.time()
        searcher = la.BackendSearcher.build_device(backend, X.data_ptr(), rows, d, ld, M, efc, device=local_rank,
                                                   key_offset=row0)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    gi = subs[0].graph_info() if (composite and rgraph) else searcher.graph_info()
    if composite:  # the composite handle carries n and dims only: degrees are the build's
        gi = dict(gi, M=M, M0=(M if backend else 2 * M))
    log(f"rank {rank}: index built on GPU in {build_s:.1f}s ({rows * G / build_s:.0f} rows/s), max_level={gi['max_level']}")

    # ---- queries: a pool of distinct batches, same on every rank in shard mode --------------------
    Q = torch.empty((n_pool * B, ld), dtype=torch.float32, device=dev)
    if rgraph:  # queries = embeddings of query-side features (the provider embeds the query text the same way)
        Fq = torch.empty((n_pool * B, hfeat), dtype=torch.int16, device=dev)
        chk(L.leann_synth_features_device(SEED, hfeat, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 1, q_first, n_pool * B, Fq.data_ptr(), sp))
        rq_h = C.c_void_p()
        chk(L.leann_recompute_create(Fq.data_ptr(), n_pool * B, hfeat, Wt.data_ptr(), d, local_rank, 0, C.byref(rq_h)))
        chk(L.leann_recompute_encode_device(rq_h, 0, n_pool * B, Q.data_ptr(), sp))
        stream.synchronize()
    else:
        with torch.cuda.stream(stream):
            chk(L.leann_synth_rows_device(SEED, d, ld, GEN_R, GEN_CLUSTERS, GEN_SIGMA, 1, q_first, n_pool * B, Q.data_ptr(), sp))
    # two sets of result buffers: with N > 1 the library runs the exchange + merge of step i on its own stream while the
    # traversal of step i + 1 fills the other set (the all-gather is latency-bound, so it is overlapped)
    hybrid = bool(args.hybrid)
    kk = 5 * k if hybrid else k  # hybrid: fetch_k = 5 * top_k (searcher.rs:129-133)
    compat = args.compat_polarity == "true"
    BM_P = 64  # BM25 positives per query (a BM25 score vector is zero except for passages sharing a term with the query)
    kbuf = [torch.empty((B, kk), dtype=torch.int64, device=dev) for _ in range(2)]
    dbuf = [torch.empty((B, kk), dtype=torch.float32, device=dev) for _ in range(2)]
    hyb = None
    if hybrid:
        if shard or composite:
            sys.exit("bench.py --hybrid: single-GPU leg (configs[4] names 1 x MI355X)")
        hyb = dict(keys=torch.empty((B, k), dtype=torch.int64, device=dev), scores=torch.empty((B, k), dtype=torch.float32, device=dev),
                   counts=torch.empty((B,), dtype=torch.int32, device=dev), pos=None, sc=None, cnt=None, ms=[])
    cbuf = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(2)]
    keys, dists, counts = kbuf[0], dbuf[0], cbuf[0]
    xstream = torch.cuda.Stream(device=dev)
    stats = torch.zeros((n_pool, B, 4), dtype=torch.int32, device=dev)
    stream.synchronize()
    group, tickets = None, [None, None]
    rccl = shard and dist.get_backend() == "nccl"
    if rccl:
        # one RCCL communicator of the LIBRARY's own: local traversal, ncclAllGather of the packed per-shard block, merge kernel
        # (csrc/shard.hip).  torch.distributed only carries the 128-byte id from rank 0 to the other ranks.
        from leann_rs_amd.shard import rccl_group
        ok = 1
        try:
            group = rccl_group(searcher, corpus_total, world, rank)
        except Exception as e:  # noqa: BLE001  (e.g. librccl.so not resolvable): every rank agrees on the fallback below
            ok, group = 0, None
            print(f"[bench] rank {rank}: library RCCL group unavailable: {e}", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if group is not None:
                group.close()
            group, rccl = None, False
            log("falling back to the torch.distributed all-gather + HIP merge kernel for the exchange step")
        else:
            log(f"rank {rank}: attached to the library's RCCL group ({group.n_shards()} shards, {group.len()} rows)")
    ev_search = [torch.cuda.Event() for _ in range(2)]
    ev_xdone = [None, None]

    allow = None
    if args.filter_selectivity > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x5EED0004 + rank)
        bits = torch.zeros(((rows + 7) // 8) * 8, dtype=torch.uint8, device=dev)
        bits[:rows] = (torch.rand(rows, device=dev, generator=gen) < args.filter_selectivity).to(torch.uint8)
        wts = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
        allow = (bits.view(-1, 8) * wts).sum(1, dtype=torch.int32).to(torch.uint8).contiguous()  # bit i&7 of byte i>>3
        del bits
        torch.cuda.synchronize()

    def search(step, timed_events=None):
        """one step of the hot path on `stream` (N > 1: + exchange and merge on the library's stream, overlapping the next step's traversal)"""
        qb = step % n_pool
        b = step & 1 if shard else 0
        keys, dists, counts = kbuf[b], dbuf[b], cbuf[b]
        qptr = Q.data_ptr() + qb * B * ld * 4
        if rccl and tickets[b] is not None:
            group.wait(tickets[b], sp)  # the exchange that wrote this buffer set two steps ago (at most two tickets may be outstanding)
            tickets[b] = None
        if shard and not rccl and ev_xdone[b] is not None:
            stream.wait_event(ev_xdone[b])
        if timed_events is not None:
            timed_events[0].record(stream)
        if rccl:
            tickets[b] = group.search_batch_device_async(qptr, B, k, ef, keys.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                                         stats.data_ptr() + qb * B * 16, sp)
        elif allow is not None and args.filter_exact:
            searcher.search_filtered_exact_batch_device(qptr, B, k, allow.data_ptr(), 0, keys.data_ptr(), dists.data_ptr(), counts.data_ptr(), sp)
        elif allow is not None:
            searcher.search_filtered_batch_device(qptr, B, k, ef, allow.data_ptr(), 0, keys.data_ptr(), dists.data_ptr(),
                                                  counts.data_ptr(), stats.data_ptr() + qb * B * 16, sp)
        else:
            searcher.search_batch_device(qptr, B, kk, ef, keys.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                         stats.data_ptr() + qb * B * 16, sp)
        if timed_events is not None:
            timed_events[1].record(stream)  # (the traversal is queued on `stream`; exchange + merge are not inside this bracket)
        if hybrid and hyb["pos"] is not None:  # injection of BM25-only hits + hybrid_rerank, on the device, same stream
            chk(L.leann_hybrid_rerank_device(keys.data_ptr(), dists.data_ptr(), counts.data_ptr(), B, kk, hyb["pos"][qb].data_ptr(),
                                             hyb["sc"][qb].data_ptr(), hyb["cnt"][qb].data_ptr(), BM_P, corpus_total, args.hybrid_alpha,
                                             1 if compat else 0, k, hyb["keys"].data_ptr(), hyb["scores"].data_ptr(), hyb["counts"].data_ptr(), sp))
            if timed_events is not None and len(timed_events) > 2:
                timed_events[2].record(stream)
        if shard and not rccl:
            # rehearsal without one GPU per rank (LEANN_BENCH_DIST_BACKEND=gloo): torch all-gather + the HIP merge kernel
            ev_search[b].record(stream)
            xstream.wait_event(ev_search[b])
            with torch.cuda.stream(xstream):
                g_keys, g_dists, g_counts = exchange_topk(keys, dists, counts, world)
                m_keys, _, _ = hip_merge(g_keys, g_dists, g_counts, k, False, xstream.cuda_stream)
                ev_xdone[b] = torch.cuda.Event()
                ev_xdone[b].record(xstream)
            return m_keys
        return keys

    def drain():
        """all outstanding exchanges have landed"""
        for b in range(2):
            if tickets[b] is not None:
                group.wait(tickets[b], sp)
                tickets[b] = None
        stream.synchronize(); xstream.synchronize()

    # ---- recall@10 against exact brute force on the same vectors ---------------------------------
    nrq = min(args.recall_queries, B)
    gt_k = torch.empty((nrq, k), dtype=torch.int64, device=dev)
    gt_s = torch.empty((nrq, k), dtype=torch.float32, device=dev)
    gt_c = torch.empty((nrq,), dtype=torch.int32, device=dev)
    t0 = time.time()
    if composite:  # per-device exact lists (scores descending), gathered on device 0 and merged there
        parts = []
        for g in range(G):
            dg = torch.device("cuda", devs[g])
            with torch.cuda.device(dg):
                Qg = Q[:nrq].to(dg)
                pk, ps, pc = (torch.empty((nrq, k), dtype=torch.int64, device=dg), torch.empty((nrq, k), dtype=torch.float32, device=dg),
                              torch.empty((nrq,), dtype=torch.int32, device=dg))
                if rgraph:
                    chk(L.leann_recompute_search_batch_device(keep[3 * g + 2], Qg.data_ptr(), nrq, k, None, pk.data_ptr(), ps.data_ptr(), pc.data_ptr(), None))
                else:
                    chk(L.leann_scan_topk_device(Xs[g].data_ptr(), rows, d, ld, Qg.data_ptr(), nrq, k, None, g * rows, pk.data_ptr(), ps.data_ptr(),
                                                 pc.data_ptr(), None))
                torch.cuda.synchronize(dg)
                parts.append((pk.to(dev), ps.to(dev), pc.to(dev)))
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            gt_k, gt_s, gt_c = hip_merge(torch.stack([p_[0] for p_ in parts]), torch.stack([p_[1] for p_ in parts]),
                                         torch.stack([p_[2] for p_ in parts]), k, True, stream.cuda_stream)
        stream.synchronize()
    elif rgraph:  # exact truth = the brute-force recompute search (fused MFMA kernel) over the same encoder
        chk(L.leann_recompute_search_batch_device(rc_h, Q.data_ptr(), nrq, k, allow.data_ptr() if allow is not None else None, gt_k.data_ptr(), gt_s.data_ptr(),
                                                  gt_c.data_ptr(), sp))
    else:
        chk(L.leann_scan_topk_device(X.data_ptr(), rows, d, ld, Q.data_ptr(), nrq, k, allow.data_ptr() if allow is not None else None, row0, gt_k.data_ptr(),
                                     gt_s.data_ptr(), gt_c.data_ptr(), sp))
    stream.synchronize()
    if shard:  # global truth = merge of per-shard exact lists (scores descending)
        with torch.cuda.stream(stream):
            gk, gs, gc = exchange_topk(gt_k, gt_s, gt_c, world)
            gt_k, gt_s, gt_c = hip_merge(gk, gs, gc, k, True, stream.cuda_stream)
        stream.synchronize(); xstream.synchronize()
    log(f"exact ground truth for {nrq} queries in {time.time() - t0:.2f}s")
    truth = gt_k.cpu().numpy()

    def measure_recall(lo_q, hi_q):
        found = search(0)
        drain()
        got = found[lo_q:hi_q, :k].cpu().numpy()  # (hybrid: the first k of the fetch_k ANN hits — recall of the ANN stage)
        return float(np.mean([len(set(got[i].tolist()) & set(truth[lo_q + i].tolist())) / k for i in range(hi_q - lo_q)]))

    # "QPS @ recall@10 >= 0.95": the cheapest beam that still meets the bar is picked on the FIRST half of the recall queries and the
    # recall that goes into the record is measured on the OTHER half (disjoint queries: no tuning on the reported set)
    half = nrq // 2 if ef_auto and nrq >= 200 else 0
    if ef_auto and not (allow is not None and args.filter_exact):
        for cand in (*range(40, 65, 2), 68, 72, 76, 80, 88, 96, 104, 112, 120, 128):
            ef = cand
            if measure_recall(0, half or nrq) >= 0.955:
                break
        log(f"--ef auto picked ef={ef} on recall queries [0, {half or nrq})")
    recall = measure_recall(half, nrq)
    bumped = False
    while ef_auto and half and recall < 0.95 and ef < 128:  # guard only: the metric is defined at recall >= 0.95 on the REPORTED queries
        ef = min(128, ef + (2 if ef < 64 else 8))
        recall, bumped = measure_recall(half, nrq), True
    log(f"recall@{k} = {recall:.4f} at ef={ef} (recall queries [{half}, {nrq}), corpus {corpus_total} x {d})" + (" [ef raised after the report-half check]" if bumped else ""))

    if hybrid:
        # Synthetic BM25 side ("synthetic BM25 score vector, seeded sparse positives"), resident in HBM before the
        # timed region as the output of the host's persistent Bm25Scorer would be: per query BM_P positives — 16 of them ANN hits of that
        # query (ranks 2, 5, 8, ...: a passage that matches lexically AND semantically), 48 elsewhere in the corpus — with quantised
        # scores (ties), sorted by (score desc, position asc) like Bm25Scorer::search (bm25.rs:109-122).
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x5EED0006)
        stride_ = max(1, corpus_total // 97)
        P_, S_, C_ = [], [], []
        for qb in range(n_pool):
            found = search(qb * 1)  # pool batch qb (step % n_pool == qb for the first n_pool steps)
            drain()
            ann = found[:, 2:2 + 3 * 16:3].clone()  # [B, 16] positions from the ANN list
            base = torch.randint(0, corpus_total, (B, 1), device=dev, generator=gen, dtype=torch.int64)
            rnd = (base + torch.arange(1, 49, device=dev, dtype=torch.int64)[None, :] * stride_) % corpus_total
            pos = torch.cat([ann, rnd], 1)
            sc = (torch.randint(1, 49, (B, BM_P), device=dev, generator=gen).to(torch.float32) * 0.25)
            sc = torch.where(pos < 0, torch.full_like(sc, -1.0), sc)  # short ANN lists carry key -1 (UINT64_MAX): not a passage
            pos, o = torch.sort(pos, dim=1, stable=True)             # duplicates adjacent ...
            sc = torch.gather(sc, 1, o)
            dup = torch.zeros_like(pos, dtype=torch.bool)
            dup[:, 1:] = pos[:, 1:] == pos[:, :-1]
            sc = torch.where(dup, torch.full_like(sc, -1.0), sc)     # ... and dropped
            sc, o = torch.sort(sc, dim=1, descending=True, stable=True)  # score desc, ties keep position asc
            pos = torch.gather(pos, 1, o)
            P_.append(pos.to(torch.int32).contiguous()); S_.append(sc.contiguous()); C_.append((sc > 0).sum(1).to(torch.int32).contiguous())
        hyb["pos"], hyb["sc"], hyb["cnt"] = P_, S_, C_
        torch.cuda.synchronize()

    # ---- warmup, then exactly K timed steps between barrier + synchronize -------------------------
    for w in range(args.warmup):
        search(w)
    drain()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3 if hybrid else 2)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        search(s_, ev[s_])
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- N > 1, second curve ("report both shard and replica curves"): no collective — every rank answers DIFFERENT
    # query batches from its own `rows`-row index, which is what N replicas of a `rows`-row corpus do.  Not `value`: a side key.
    replica_qps = None
    if shard and allow is None:
        def local(step):
            qb = (step + 3 * rank) % n_pool
            searcher.search_batch_device(Q.data_ptr() + qb * B * ld * 4, B, k, ef, kbuf[0].data_ptr(), dbuf[0].data_ptr(), cbuf[0].data_ptr(), None, sp)
        for w in range(args.warmup):
            local(w)
        stream.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(args.steps):
            local(s_)
        stream.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        replica_qps = B * args.steps * world / float(t2.item())

    # ---- roofline of the dominant kernel (beam_search_kernel): algorithmic bytes / HIP-event time --
    kern_ms = [e[0].elapsed_time(e[1]) for e in ev]
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3
    st = stats.cpu().numpy().astype(np.int64)  # every pool batch was searched at least once (steps >= n_pool or fewer batches)
    used = min(n_pool, max(args.steps, 1 + args.warmup))
    st = st[:used].reshape(-1, 4)
    evals, hops0, hopsU, ovf = [float(x) for x in st.sum(0)]
    nq_stat = st.shape[0]
    row_bytes = 2 * hfeat + 4 if rgraph else d * 4  # recompute-on: 256 bf16 features + the f32 norm per evaluated neighbour (round 2 counted the 520-B padded row)
    bytes_per_query = (evals * row_bytes + hops0 * gi["M0"] * 4 + hopsU * gi["M"] * 4) / nq_stat
    bytes_per_launch = bytes_per_query * B
    achieved = bytes_per_launch / kern_avg_s / 1e9
    if composite:  # counters are summed over the G shards and the bracket spans fan-out + all traversals + gather + merge: per-GPU figure
        achieved /= G
    # HBM traffic per launch comes from rocprofv3 --pmc passes (separate runs: counters cannot be read in-process); the committed figure
    # is only quoted when this run has the same shape as the profiled one, and its provenance goes into the record
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("ef_search") == ef and B == 16384 and k == 10 and allow is None and n_gpus == 1:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/pmc_traffic_{args.workload}.json — rocprofv3 --pmc of this command at ef={ef} in an earlier run "
                                  "(FETCH_SIZE x 2 + WRITE_SIZE at the L2<->fabric boundary: Infinity-Cache hits included); not measured in this run")
            else:
                traffic_source = f"none: profiles/pmc_traffic_{args.workload}.json was taken at ef={tj.get('ef_search')}, batch 16384, k 10 — this run differs"
        except Exception:
            traffic = None
    else:
        traffic_source = "none: no PMC profile committed for this workload"

    # value: queries answered per second over the WHOLE corpus.  shard mode: every query is searched on all N shards and answered
    # once (B * K / t); replica mode / N = 1: each rank answers its own batches (N * B * K / t)
    units = B * args.steps * (1 if shard else world)
    value = units / elapsed
    out = {
        "metric": "queries/sec @ recall@10>=0.95",
        "value": value,
        "unit": "queries/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "recall_at_10": recall,
        "config": {
            "workload": f"{args.workload}: {'Vamana R' if backend else 'HNSW M'}={M} efc={efc} ef={ef} k={k}, {rows} x {d} "
                        + (f"embeddings per GPU recomputed on the fly from [{rows} x {wl.get('h')}] bf16 features (no stored vectors), " if rgraph
                           else "f32 rows per GPU, ")
                        + f"batch {B} queries/step, clustered synthetic (r={GEN_R}, C={GEN_CLUSTERS}, sigma={GEN_SIGMA})",
            "rows_per_gpu": rows, "corpus_rows_total": corpus_total, "dims": d, "M": M, "ef_construction": efc,
            "ef_search": ef, "top_k": k, "batch": B,
            "parallelism": ((f"shard{world}+rccl_allgather" if rccl else f"shard{world}+{dist.get_backend()}_allgather(torch; rehearsal)") if shard
                            else (f"composite{G}: one process, {G} devices behind one handle, peer copies + merge on device 0 (no RCCL)" if composite
                                  else ("single" if world == 1 else f"replica{world}"))),
            "index_build_s": build_s,
            **({"filter_selectivity": args.filter_selectivity,
                "filter_note": "side experiment: allow-bitmap evaluated inside the traversal; recall vs the exact filtered top-k"}
               if allow is not None else {}),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "boundary_note": "achieved = algorithmic bytes / kernel time; the bytes cross the L2<->fabric boundary (all of the L2's read "
                             "requests take the DRAM path, TCC_EA0_RDREQ_DRAM = 100 %); the memory-side 256 MB Infinity Cache sits behind "
                             "that interface with no counter exposed on this pool, so the HBM / Infinity-Cache split is unmeasured "
                             "(profiles/r02_other_kernels.md)",
            "kernel": ("beam_search_filtered_kernel" if allow is not None else
                       ("beam_search_feat256_kernel<1,4>" if B > 512 else "beam_search_feat256_kernel<1,16>") + " (+ score_mfma_kernel query projection)" if rgraph else
                       "beam_search_kernel<3,4,4,false>" if ld == 768 else "beam_search_kernel"),
            "kernel_avg_ms": kern_avg_s * 1e3, "algorithmic_bytes_per_query": bytes_per_query,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "dist_evals_per_query": evals / nq_stat, "hops_per_query": (hops0 + hopsU) / nq_stat,
            "hbm_table_queries": ovf,
        },
    }
    if allow is not None and args.filter_exact:  # no graph involved: the scan of the allowed rows is bound by the f32 matrix cores
        n_allowed = int(torch.count_nonzero((allow.view(-1, 1) & wts.view(1, -1)) != 0).item())
        flops = 2.0 * n_allowed * d * B
        out["config"]["filter_note"] = ("side experiment: exact filtered search — the %d allowed rows are compacted and scanned "
                                        "(score_mfma_kernel<true>, candidate emission), no graph walk" % n_allowed)
        out["config"]["ef_search"] = None
        out["roofline"] = {"bound": "mfma", "achieved": flops / kern_avg_s / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops / kern_avg_s / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                           "kernel": "score_mfma_kernel<true> (+ compaction, fold, finalize inside the HIP-event bracket)",
                           "kernel_avg_ms": kern_avg_s * 1e3, "algorithmic_flops_per_launch": flops, "allowed_rows": n_allowed}
    if composite:
        out["config"]["value_unit_note"] = (f"value = end-to-end queries/s over the {corpus_total}-row corpus: every query is searched on all {G} shards "
                                            f"(one per device) by ONE process; roofline.achieved is per GPU (summed algorithmic bytes / {G} / the "
                                            "HIP-event bracket around fan-out + traversals + peer copies + merge)")
    if shard:
        if replica_qps is not None:
            out["replica_mode"] = {"value": replica_qps, "unit": "queries/s",
                                   "note": f"no collective: each of the {world} ranks answers its own query batches from its own {rows}-row index "
                                           f"(N replicas of a {rows}-row corpus); same K steps, barrier-bracketed, max over ranks"}
        out["shard_searches_per_s"] = B * args.steps * world / elapsed
        out["config"]["value_unit_note"] = (f"value = end-to-end queries/s over the {corpus_total}-row corpus (every query searched on all "
                                            f"{world} shards, lists all-gathered over {'RCCL inside the library' if rccl else dist.get_backend()}, merged on every rank); "
                                            "shard_searches_per_s = value x n_gpus")
    out["config"]["recall_protocol"] = (f"ef picked on recall queries [0, {half}) (smallest of the ladder with recall >= 0.955), recall reported on [{half}, {nrq})"
                                        + (" — ef then raised until the reported half reached 0.95" if bumped else "") if half else
                                        f"fixed ef; recall on {nrq} queries")

    if args.small_batch and n_gpus == 1 and allow is None:
        # the batch BASELINE configs[2] names (64 queries per call): a latency regime — one workgroup of 16 waves per query walking its
        # dependent hops, most of the chip idle — reported beside the throughput batch so that neither figure is mistaken for the other
        sb, reps = min(args.small_batch, B), 200
        for _ in range(20):
            searcher.search_batch_device(Q.data_ptr(), sb, kk, ef, kbuf[0].data_ptr(), dbuf[0].data_ptr(), cbuf[0].data_ptr(), None, sp)
        stream.synchronize()
        t0 = time.perf_counter()
        for r_ in range(reps):
            searcher.search_batch_device(Q.data_ptr() + (r_ % max(1, (n_pool * B) // sb)) * sb * ld * 4, sb, kk, ef, kbuf[0].data_ptr(),
                                         dbuf[0].data_ptr(), cbuf[0].data_ptr(), None, sp)
        stream.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["small_batch"] = {"batch": sb, "ms_per_call": dt * 1e3, "value": sb / dt, "unit": "queries/s",
                              "hbm_frac": bytes_per_query * sb / dt / 1e9 / HBM_PEAK_GBS,
                              "note": f"{reps} back-to-back device calls of {sb} queries (latency form of the hop loop, 16 waves per query): "
                                      "a latency regime, not a bandwidth one"}
        # the same calls with several of them in flight (a server's concurrent requests, one HIP stream each): a call of 64 queries
        # occupies 64 of the 256 CUs, so independent calls overlap; per-call latency stays what it is, the rate does not
        ns_ = 8
        sts = [torch.cuda.Stream(device=dev) for _ in range(ns_)]
        bufs = [(torch.empty((sb, kk), dtype=torch.int64, device=dev), torch.empty((sb, kk), dtype=torch.float32, device=dev),
                 torch.empty((sb,), dtype=torch.int32, device=dev)) for _ in range(ns_)]
        nslice = max(1, (n_pool * B) // sb)

        def fire(r_):
            i = r_ % ns_
            searcher.search_batch_device(Q.data_ptr() + (r_ % nslice) * sb * ld * 4, sb, kk, ef, bufs[i][0].data_ptr(), bufs[i][1].data_ptr(),
                                         bufs[i][2].data_ptr(), None, C.c_void_p(sts[i].cuda_stream))
        for r_ in range(4 * ns_):
            fire(r_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r_ in range(reps * 2):
            fire(r_)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / (reps * 2)
        # ... and the same 8 concurrent calls captured ONCE into a HIP graph (fork / join over the 8 streams) and replayed: the host then
        # issues one graph launch per 8 calls instead of 16 kernel launches through the interpreter (launch-bound loops belong in graphs)
        graph_qps = None
        try:
            cap = torch.cuda.Stream(device=dev)
            gr = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=cap):
                for i in range(ns_):  # fork ...
                    sts[i].wait_stream(cap)
                for i in range(ns_):
                    fire(i)
                for i in range(ns_):  # ... join
                    cap.wait_stream(sts[i])
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps // 2):
                gr.replay()
            torch.cuda.synchronize()
            graph_qps = ns_ * sb * (reps // 2) / (time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001  (capture is an extra; the line does not depend on it)
            log("HIP-graph capture of the small-batch calls failed:", str(e)[:200])
            torch.cuda.synchronize()
        out["small_batch"]["value_concurrent_hipgraph"] = graph_qps
        out["small_batch"].update({"streams_in_flight": ns_, "value_concurrent": sb / dt8, "ms_per_call_concurrent": dt8 * 1e3,
                                   "concurrent_note": f"{reps * 2} calls of {sb} queries issued round-robin on {ns_} HIP streams (per-call latency unchanged; "
                                                      "value_concurrent = queries/s with the calls overlapping)"})
    if hybrid:
        rr_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        out["hybrid"] = {"fetch_k": kk, "alpha": args.hybrid_alpha, "compat_polarity": compat,
                         "polarity_note": ("reference as written: backend DISTANCES enter hybrid_rerank as if larger were better (SURVEY.md N1)" if compat
                                           else "corrected: 1 - dist enters the blend"),
                         "bm25": f"synthetic, resident in HBM: {BM_P} positives per query (16 of them ANN hits), quantised scores, sorted like Bm25Scorer::search",
                         "rerank_ran_on": "device (csrc/hybrid.hip:hybrid_rerank_kernel, one workgroup per query), inside the timed region",
                         "rerank_avg_ms": rr_ms, "traversal_avg_ms": kern_avg_s * 1e3,
                         "reference": "src/index/searcher.rs:129-169 + src/index/bm25.rs:135-170"}
        out["config"]["workload"] += f"; HYBRID: fetch_k={kk}, BM25 injection + hybrid_rerank(alpha={args.hybrid_alpha}) on the device per step"
        out["config"]["top_k"] = k
        out["recall_note"] = f"recall@{k} of the ANN stage (first {k} of the {kk} fetched) against the exact scan; the reranked order is BM25-blended by design"

    # ---- PCIe-inclusive rate: the same batch through the host-pointer entry point (queries and results in host memory) ----
    if n_gpus == 1 and rank == 0 and allow is None and not rgraph:
        try:
            Qh = Q[:B, :d].contiguous().cpu().numpy()
            searcher.search_batch(Qh, k, ef)
            t0 = time.perf_counter()
            for _ in range(3):
                searcher.search_batch(Qh, k, ef)
            out["pcie_inclusive_qps"] = 3 * B / (time.perf_counter() - t0)  # leann_backend_search_batch: H2D of 3 KB per query, D2H of the results, one sync
        except Exception as e:  # noqa: BLE001
            log("host-pointer timing failed:", e)

    # ---- single-query latency: the call the reference actually makes — BackendSearcher::search(query, top_k, complexity)
    # (src/backend/traits.rs:16-21), one query per call from host memory; and the serve pattern: 64 concurrent single-query callers
    # (src/cli/serve.rs:289-292), plain and with the library's request coalescing ------------------------------------------------
    lat = None
    if n_gpus == 1 and rank == 0 and allow is None and not args.no_latency:
        try:
            import threading
            nlat = min(2000, n_pool * B)
            Ql = Q[:nlat, :d].contiguous().cpu().numpy()
            kb, db, nb = np.zeros(k, np.uint64), np.zeros(k, np.float32), C.c_size_t(0)
            f32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64)
            fn, hnd = L.leann_backend_search, searcher._h
            for i in range(50):  # warm: workspace, 16-wave instantiation
                fn(hnd, Ql[i].ctypes.data_as(f32p), k, ef, kb.ctypes.data_as(u64p), db.ctypes.data_as(f32p), C.byref(nb))
            ts = np.empty(nlat)
            for i in range(nlat):
                t0 = time.perf_counter()
                chk(fn(hnd, Ql[i].ctypes.data_as(f32p), k, ef, kb.ctypes.data_as(u64p), db.ctypes.data_as(f32p), C.byref(nb)))
                ts[i] = time.perf_counter() - t0
            lat = {"call": "leann_backend_search (host pointers, 1 query: staged in a pinned device-mapped block, kernel, sync)", "n": int(nlat), "ef": ef, "top_k": k,
                   "p50_ms": float(np.percentile(ts, 50) * 1e3), "p99_ms": float(np.percentile(ts, 99) * 1e3), "mean_ms": float(ts.mean() * 1e3)}

            def callers(nthreads, per):
                def work(t):
                    kk, dd, nn = np.zeros(k, np.uint64), np.zeros(k, np.float32), C.c_size_t(0)
                    for j in range(per):
                        q = Ql[(t * per + j) % nlat]
                        fn(hnd, q.ctypes.data_as(f32p), k, ef, kk.ctypes.data_as(u64p), dd.ctypes.data_as(f32p), C.byref(nn))
                th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
                t0 = time.perf_counter()
                [t.start() for t in th]
                [t.join() for t in th]
                return nthreads * per / (time.perf_counter() - t0)
            callers(64, 5)
            lat["callers64_qps_auto"] = callers(64, 40)      # the handle as opened: concurrent callers coalesce by themselves
            searcher.set_coalescing(0, 0)
            callers(64, 5)
            lat["callers64_qps_plain"] = callers(64, 40)     # coalescing switched off: one launch per caller
            searcher.set_coalescing(100, 64)
            callers(64, 5)
            lat["callers64_qps_coalesced"] = callers(64, 40)
            lat["coalescing"] = ("auto = default handle (dispatcher installed when a caller finds another in flight: 50 us, 64 queries); plain = "
                                 "leann_backend_set_coalescing(0, 0); coalesced = leann_backend_set_coalescing(wait_us=100, max_batch=64). "
                                 "Python caller threads (GIL hand-over per wake-up): leann-rs_amd/host/serve_bench measures the same natively")
            searcher.set_coalescing(0, 0)
            out["single_query"] = lat
            log(f"single query: p50 {lat['p50_ms']:.3f} ms, p99 {lat['p99_ms']:.3f} ms; 64 callers: {lat['callers64_qps_auto']:.0f} q/s as opened, "
                f"{lat['callers64_qps_plain']:.0f} q/s plain, {lat['callers64_qps_coalesced']:.0f} q/s coalesced")
        except Exception as e:  # noqa: BLE001
            log("single-query timing failed:", e)

    # ---- CPU baseline: the oracle (C restatement) walking the SAME graph on the host cores --------
    if n_gpus == 1 and not args.no_cpu_baseline and rank == 0 and allow is None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            t0 = time.time()
            ncpu = min(args.cpu_queries, B)
            if rgraph:
                g = searcher.graph_export()
                fh, rb = C.c_uint32(0), C.c_uint32(0)
                chk(L.leann_backend_feature_rows_export(searcher._h, C.byref(fh), C.byref(rb), None))
                rows_b = np.zeros((rows, rb.value), np.uint8)
                chk(L.leann_backend_feature_rows_export(searcher._h, None, None, rows_b.ctypes.data))
                G = po.Graph.from_arrays(np.zeros((rows, 1), np.float32), g["M"], g["M0"], g["max_level"], g["entry"], g["levels"],
                                         g["upper_off"], g["adj0"], g["adjU"])
                G.set_features(rows_b, fh.value, rb.value)
                ncpu = min(ncpu, 2048)
                Qh = po.project_queries(Wt.cpu().numpy().view(np.uint16), Q[:ncpu, :d].contiguous().cpu().numpy(), fh.value)
            else:
                g = searcher.graph_export(with_vectors=True)
                G = po.Graph.from_arrays(g["vectors"], g["M"], g["M0"], g["max_level"], g["entry"], g["levels"],
                                         g["upper_off"], g["adj0"], g["adjU"])
                Qh = Q[:ncpu, :d].contiguous().cpu().numpy()
            log(f"graph + rows copied to host in {time.time() - t0:.1f}s")
            cores = host_cores()
            G.search_batch(Qh[: min(256, ncpu)], kk, ef, 0, cores)  # touch
            t0 = time.perf_counter()
            ck, cd, cc, cs = G.search_batch(Qh, kk, ef, 0, cores)
            cpu_s = time.perf_counter() - t0
            # second figure: the same walk with a plain 4-accumulator AVX2 dot (oracle.c:orc_dot_fast — what a SIMD
            # library such as usearch's does) instead of the bit-exact canonical tree; timing only, never used for the identity check
            simd = None
            try:
                po.lib().orc_set_fast_dot(1)
                G.search_batch(Qh[: min(256, ncpu)], kk, ef, 0, cores)
                t0 = time.perf_counter()
                G.search_batch(Qh, kk, ef, 0, cores)
                simd = ncpu / (time.perf_counter() - t0)
            finally:
                po.lib().orc_set_fast_dot(0)
            search(0)
            stream.synchronize()
            gk_ = keys[:ncpu].cpu().numpy().view(np.uint64)
            gd_ = dists[:ncpu].cpu().numpy()
            same = bool((gk_ == ck).all() and (gd_.view(np.uint32) == cd.view(np.uint32)).all())
            rerank_note = ""
            if hybrid:
                # the reference's rerank on the host, per query: injection + hybrid_rerank over a DENSE N-long BM25 vector (its min / max
                # fold runs over all N scores, bm25.rs:152-154) — oracle/oracle.c:orc_hybrid_rerank, one query per thread
                from concurrent.futures import ThreadPoolExecutor
                import threading
                import searcher_oracle as so
                pp = hyb["pos"][0][:ncpu].cpu().numpy().view(np.uint32)
                ps = hyb["sc"][0][:ncpu].cpu().numpy()
                pc = hyb["cnt"][0][:ncpu].cpu().numpy()
                tls = threading.local()

                def cpu_rerank(q):
                    if not hasattr(tls, "dense"):
                        tls.dense = np.zeros(corpus_total, np.float32)
                    n_, c_ = int(cc[q]), int(pc[q])
                    vr = [(int(a_), float(d_) if compat else float(np.float32(1.0) - d_)) for a_, d_ in zip(ck[q, :n_], cd[q, :n_])]
                    have = {i for i, _ in vr}
                    vr += [(int(p_), 0.0) for p_ in pp[q, :min(c_, kk)] if int(p_) not in have]
                    tls.dense[pp[q, :c_]] = ps[q, :c_]
                    r_ = po.hybrid_rerank(vr, tls.dense, args.hybrid_alpha)[:k]
                    tls.dense[pp[q, :c_]] = 0.0
                    return r_
                nrr = min(ncpu, 2048)
                with ThreadPoolExecutor(cores) as ex:
                    list(ex.map(cpu_rerank, range(min(64, nrr))))
                    t0 = time.perf_counter()
                    cpu_rr = list(ex.map(cpu_rerank, range(nrr)))
                    rr_s = (time.perf_counter() - t0) * ncpu / nrr
                cpu_s += rr_s
                rerank_note = f" + hybrid rerank per query over a dense {corpus_total}-long BM25 vector ({rr_s / ncpu * 1e3 * cores:.2f} ms per query and thread)"
                # parity of the DEVICE rerank: (id, f32 score) lists of a >= 1000-query sample against oracle/searcher_oracle.py
                nsmp = min(ncpu, 1024)
                hk_ = hyb["keys"][:nsmp].cpu().numpy().view(np.uint64)
                hs_ = hyb["scores"][:nsmp].cpu().numpy()
                hc_ = hyb["counts"][:nsmp].cpu().numpy()
                bad = 0
                for q in range(nsmp):
                    exp = so.hybrid_leg_sparse(ck[q, :cc[q]], cd[q, :cc[q]], list(zip(pp[q, :pc[q]].tolist(), ps[q, :pc[q]])), corpus_total,
                                               args.hybrid_alpha, k, kk, compat)
                    ok_ = (hc_[q] == len(exp) and [int(x) for x in hk_[q, :hc_[q]]] == [i for i, _ in exp] and
                           (hs_[q, :hc_[q]].view(np.uint32) == np.array([x for _, x in exp], np.float32).view(np.uint32)).all())
                    bad += 0 if ok_ else 1
                out["hybrid"]["rerank_parity"] = {"sample": nsmp, "mismatching_queries": bad,
                                                  "against": "oracle/searcher_oracle.py:hybrid_leg_sparse on the oracle's own walk of the same graph (ids and f32 score bits)"}
                # the C port and the numpy restatement must agree too (two restatements of bm25.rs:135-170)
                out["hybrid"]["cpu_port_equals_numpy_restatement"] = all(
                    [i for i, _ in cpu_rr[q]] == [i for i, _ in so.hybrid_leg_sparse(ck[q, :cc[q]], cd[q, :cc[q]], list(zip(pp[q, :pc[q]].tolist(), ps[q, :pc[q]])),
                                                                                       corpus_total, args.hybrid_alpha, k, kk, compat)] for q in range(min(nrr, 256)))
            out["cpu_baseline"] = {
                "value": ncpu / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"{ncpu} queries of the timed batch, same graph + vectors copied from HBM, ef={ef}, k={kk}, "
                          f"one query per thread on {cores} host threads (oracle/oracle.c, AVX2 canonical dot)" + rerank_note,
                "gpu_results_bit_identical_on_sample": same,
            }
            if simd is not None and not rgraph:
                out["cpu_baseline"]["simd_value"] = simd
                out["cpu_baseline"]["simd_note"] = ("kind port-simd: the same walk on the same threads with a plain 4-accumulator AVX2 dot "
                                                    "(not bit-compatible with the GPU; the graph walk only, no rerank); `value` uses the bit-exact canonical dot")
            if lat is not None:  # the CPU port's latency for ONE query on ONE thread, beside the GPU's single-query latency
                tc = np.empty(min(300, ncpu))
                for i in range(len(tc)):
                    t0 = time.perf_counter()
                    G.search(Qh[i], k, ef, 0)
                    tc[i] = time.perf_counter() - t0
                lat["cpu_port_p50_ms"] = float(np.percentile(tc, 50) * 1e3)
                lat["cpu_port_p99_ms"] = float(np.percentile(tc, 99) * 1e3)
                lat["note"] = ("GPU single-query latency is one workgroup of 16 waves walking ~%d dependent hops; the CPU port walks the same "
                               "graph on one core (warm caches: the sample's rows were just touched by the throughput run)" % round((hops0 + hopsU) / nq_stat))
            log(f"cpu baseline: {ncpu / cpu_s:.0f} q/s on {cores} threads; GPU == oracle on sample: {same}")
        except Exception as e:  # the baseline is a reported side figure; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        emit(out)
    if group is not None:
        torch.cuda.synchronize()
        group.close()  # ncclCommDestroy: collective-free, but every rank gets here
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
