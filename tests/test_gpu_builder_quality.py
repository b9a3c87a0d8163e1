"""-m gpu: the GPU builder's graphs against the oracle's SEQUENTIAL builder.

Batched, permuted insertion (csrc/build.hip) is a different schedule from the reference's row-by-row `add`
(src/backend/hnsw.rs:112-130) / diskann-rs's build (src/backend/diskann.rs:88-92, alpha 1.2), so graphs cannot be compared bit for
bit; what CAN be pinned is that the batched schedule costs no quality: on the same rows, with the same M / ef_construction,
    recall@10 of the GPU-built graph at ef in {32, 64}  >=  recall@10 of the sequentially built graph - 0.01
    mean level-0 degree within 10 % of the sequential graph's
Both graphs are searched by the same oracle walk (so the comparison is about the graphs, not the searcher)."""
import numpy as np
import pytest

from util import SEED, recall_at_k

pytestmark = pytest.mark.gpu


def _rows(la, n, d, stream, clusters=256):
    buf = la.DeviceArray((n, d), np.float32)
    la._native.check(la.lib().leann_synth_rows_device(SEED, d, d, 64, clusters, 1.0, stream, 0, n, buf.ptr, None))
    la.sync()
    return buf


def _mean_degree(adj0):
    return float((np.asarray(adj0) != 0xFFFFFFFF).sum(1).mean())


@pytest.mark.parametrize("n,d", [(50_000, 128), (20_000, 768)])
def test_hnsw_builder_matches_sequential_quality(la, po, gpu, n, d):
    M, efc = 16, 64
    dX = _rows(la, n, d, 0)
    X, Q = dX.to_host(), _rows(la, 300, d, 1).to_host()
    truth = po.exact_topk(X, Q, 10)
    seq = po.Graph.build_hnsw(X, M=M, efc=efc)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, efc)
    g = s.graph_export()
    gpu_graph = po.Graph.from_arrays(X, M, 2 * M, g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    deg_seq, deg_gpu = _mean_degree(seq.export()[2]), _mean_degree(g["adj0"])
    print(f"hnsw {n}x{d}: mean level-0 degree sequential {deg_seq:.2f} / GPU {deg_gpu:.2f}")
    assert abs(deg_gpu - deg_seq) <= 0.10 * deg_seq
    for ef in (32, 64):
        r_seq = recall_at_k(seq.search_batch(Q, 10, ef, 0, nthreads=8)[0], truth)
        r_gpu = recall_at_k(gpu_graph.search_batch(Q, 10, ef, 0, nthreads=8)[0], truth)
        print(f"  ef={ef}: recall@10 sequential {r_seq:.4f} / GPU {r_gpu:.4f}")
        assert r_gpu >= r_seq - 0.01
    s.close()


@pytest.mark.parametrize("n,d", [(50_000, 128), (20_000, 768)])
def test_vamana_builder_matches_sequential_quality(la, po, gpu, n, d):
    R, L = 32, 64
    dX = _rows(la, n, d, 0)
    X, Q = dX.to_host(), _rows(la, 300, d, 1).to_host()
    truth = po.exact_topk(X, Q, 10)
    seq = po.Graph.build_vamana(X, R=R, L=L, alpha=1.2)
    s = la.BackendSearcher.build_device(la.BackendType.DiskAnn, dX.ptr, n, d, d, R, L)
    g = s.graph_export()
    gpu_graph = po.Graph.from_arrays(X, R, R, 0, g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    deg_seq, deg_gpu = _mean_degree(seq.export()[2]), _mean_degree(g["adj0"])
    print(f"vamana {n}x{d}: mean degree sequential {deg_seq:.2f} / GPU {deg_gpu:.2f}")
    assert deg_gpu >= 0.90 * deg_seq  # more edges within R is not a defect for a single-level graph; fewer would be
    for ef in (32, 64):
        r_seq = recall_at_k(seq.search_batch(Q, 10, ef, 1, nthreads=8)[0], truth)
        r_gpu = recall_at_k(gpu_graph.search_batch(Q, 10, ef, 1, nthreads=8)[0], truth)
        print(f"  L={ef}: recall@10 sequential {r_seq:.4f} / GPU {r_gpu:.4f}")
        assert r_gpu >= r_seq - 0.01
    s.close()


def test_vamana_r32_stays_navigable_at_4m_rows(la, gpu):
    """at the reference's default degree (graph_degree 32, src/cli/build.rs:79; alpha 1.2, src/backend/diskann.rs:91)
    the round-2 builder reached recall@10 0.98 at 1M rows and 0.60 at 10M x 1536 (beam 128).  Cause: the one-stage RobustPrune of the
    paper keeps the R nearest candidates on data of high intrinsic dimension; with DiskANN's two-stage form (build.hip:prune_core) the
    same rows give 0.975-0.98 at 10M (profiles/r03_vamana_scale.md).  Here: 4M x 256 rows, R = 32, build beam 128, search beam 128 ->
    recall@10 >= 0.95 (the one-stage rule measures ~0.92 at this size and fails), every node reachable from another one."""
    n, d, R, nq, k = 4_000_000, 256, 32, 1000, 10
    L_, chk = la.lib(), la._native.check
    dX = la.DeviceArray((n, d), np.float32)
    chk(L_.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 0, 0, n, dX.ptr, None))
    dQ = la.DeviceArray((nq, d), np.float32)
    chk(L_.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 1, 0, nq, dQ.ptr, None))
    la.sync()
    tk, ts, tc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(L_.leann_scan_topk_device(dX.ptr, n, d, d, dQ.ptr, nq, k, None, 0, tk.ptr, ts.ptr, tc.ptr, None))
    la.sync()
    truth = tk.to_host()
    s = la.BackendSearcher.build_device(la.BackendType.DiskAnn, dX.ptr, n, d, d, R, 128)
    gk, gd, gc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    s.search_batch_device(dQ.ptr, nq, k, 128, gk.ptr, gd.ptr, gc.ptr)
    la.sync()
    rec = recall_at_k(gk.to_host(), truth)
    adj = s.graph_export()["adj0"]
    valid = adj != 0xFFFFFFFF
    indeg = np.bincount(adj[valid].astype(np.int64), minlength=n)
    print(f"vamana R=32, {n} x {d}: recall@10 {rec:.4f} at L=128, in-degree 0: {(indeg == 0).mean():.2e}, mean out-degree {valid.sum(1).mean():.2f}")
    assert rec >= 0.95
    assert (indeg == 0).mean() < 1e-4
    s.close()
