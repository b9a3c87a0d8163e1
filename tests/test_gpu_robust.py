"""-m gpu: robustness of the C ABI: re-entrancy on one handle, large beams (64/128 KiB LDS tables),
large k, unpadded dims, degenerate sizes, add_to_index."""
import threading

import numpy as np
import pytest

from util import recall_at_k, synth

pytestmark = pytest.mark.gpu


def _oracle_twin(po, s, X):
    g = s.graph_export()
    return po.Graph.from_arrays(X, g["M"], g["M0"], g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])


def test_concurrent_search_on_one_handle(la, po, gpu):
    """BackendSearcher is Send + Sync (traits.rs:11); serve.rs:289-292 calls search from many threads."""
    X = synth(po, 20000, 128)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, 20000, 128, 128, 16, 64)
    Qs = [synth(po, 300, 128, stream=1, i0=1000 * t) for t in range(8)]
    serial = [s.search_batch(Q, 10, 64) for Q in Qs]
    out = [None] * 8

    def work(t):
        for _ in range(5):
            out[t] = s.search_batch(Qs[t], 10, 64)

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for a, b in zip(serial, out):
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    # single-query calls interleaved with batches
    k1, d1 = s.search(Qs[0][7], 10, 64)
    assert (k1 == serial[0][0][7]).all()
    s.close()


@pytest.mark.parametrize("ef,k", [(300, 10), (700, 200), (1200, 50)])
def test_large_beams_match_oracle(la, po, gpu, ef, k):
    X = synth(po, 6000, 64)
    Q = synth(po, 24, 64, stream=1)
    G = po.Graph.build_hnsw(X, M=16, efc=64)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 16, 32, G.max_level, G.entry, lv, uo, a0, aU)
    ok, od, oc, _ = G.search_batch(Q, k, ef, 0, 4)
    gk, gd, gc = s.search_batch(Q, k, ef)
    assert (gc == oc).all() and (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    s.close()


def test_unpadded_dims_through_build(la, po, gpu, tmp_path):
    n, d = 3000, 130  # ld = 132 inside the library
    X = synth(po, n, d, r=16)
    Q = synth(po, 30, d, stream=1, r=16)
    stem = str(tmp_path / "documents.leann")
    la.BackendBuilder(la.BackendType.Hnsw).build(X, [], stem, d, 8, 32)
    s = la.HnswSearcher.load(stem, d)
    assert s.dims() == d and s.graph_info()["ld"] == 132
    G = _oracle_twin(po, s, X)
    ok, od, oc, _ = G.search_batch(Q, 5, 40, 0, 2)
    gk, gd, gc = s.search_batch(Q, 5, 40)
    assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    with pytest.raises(la.LeannError, match="dimensions"):
        la.HnswSearcher.load(stem, 128)
    s.close()


def test_degenerate_sizes(la, po, gpu, tmp_path):
    for n in (0, 1, 2, 3):
        X = synth(po, max(n, 1), 32, r=0)[:n]
        stem = str(tmp_path / f"n{n}" / "documents.leann")
        (tmp_path / f"n{n}").mkdir()
        la.BackendBuilder(la.BackendType.Hnsw).build(X.reshape(n, 32), [], stem, 32, 4, 8)
        s = la.HnswSearcher.load(stem, 32)
        assert s.len() == n and s.is_empty() == (n == 0)
        q = synth(po, 1, 32, stream=1, r=0)[0]
        keys, dists = s.search(q, 5, 16)
        assert len(keys) == min(n, 5) and sorted(keys.tolist()) == list(range(min(n, 5)))
        kb, db, cb = s.search_batch(np.zeros((0, 32), np.float32), 5, 16)
        assert kb.shape == (0, 5)
        s.close()


def test_add_to_index_equals_one_shot_build(la, po, gpu, tmp_path):
    """add_to_index appends to the loaded graph (hnsw.rs:142-191): the new rows continue the batched insertion from the saved state
    (stored link distances recomputed).  The insertion order is a pseudo-random permutation of the rows being added, so an append is the same
    algorithm on another schedule than a one-shot build of the concatenation: same vectors and levels, valid lists, same recall."""
    d = 64
    Q = synth(po, 100, d, stream=1)
    B = la.BackendBuilder(la.BackendType.Hnsw)
    for case, (n0, n1) in enumerate(((4096, 2500), (4000, 2500), (300, 40000))):
        X = synth(po, n0 + n1, d)
        a, b = tmp_path / f"a{case}", tmp_path / f"b{case}"
        a.mkdir(); b.mkdir()
        B.build(X[:n0], [], str(a / "documents.leann"), d, 12, 48)
        with pytest.raises(la.LeannError, match="does not continue"):
            B.add_to_index(X[n0:], str(a / "documents.leann"), d, n0 + 5)
        B.add_to_index(X[n0:], str(a / "documents.leann"), d, n0)  # ids continue at start_id
        B.build(X, [], str(b / "documents.leann"), d, 12, 48)
        sa, sb = la.HnswSearcher.load(str(a / "documents.leann"), d), la.HnswSearcher.load(str(b / "documents.leann"), d)
        assert sa.len() == n0 + n1
        ga, gb = sa.graph_export(with_vectors=True), sb.graph_export()
        assert (ga["vectors"] == X).all() and (ga["levels"] == gb["levels"]).all()
        ka, da, _ = sa.search_batch(Q, 10, 64)
        kb, db, _ = sb.search_batch(Q, 10, 64)
        truth = po.exact_topk(X, Q, 10)
        a0 = ga["adj0"]
        valid = a0 != 0xFFFFFFFF
        assert (a0[valid] < n0 + n1).all() and valid.sum(1).min() >= 1
        assert (valid[:, :-1] >= valid[:, 1:]).all()  # compact lists
        assert abs(recall_at_k(ka, truth) - recall_at_k(kb, truth)) <= 0.02
        assert recall_at_k(ka, truth) >= 0.95
        sa.close(); sb.close()
    # an index whose level table did not come from this builder (oracle graph saved through from_arrays): rebuilt on append
    n0, n1 = 1500, 500
    X = synth(po, n0 + n1, d)
    G = po.Graph.build_hnsw(X[:n0], M=8, efc=32)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X[:n0], 8, 16, G.max_level, G.entry, lv, uo, a0, aU)
    c = tmp_path / "c"
    c.mkdir()
    s.save(str(c / "documents.leann"))
    s.close()
    B.add_to_index(X[n0:], str(c / "documents.leann"), d, n0)
    sc = la.HnswSearcher.load(str(c / "documents.leann"), d)
    assert sc.len() == n0 + n1
    kc, _, _ = sc.search_batch(Q, 10, 64)
    assert recall_at_k(kc, po.exact_topk(X, Q, 10)) >= 0.9
    sc.close()


def test_build_does_not_depend_on_storage_order(la, po, gpu):
    """Rows stored topic by topic (cluster-major) must index as well as the same rows in shuffled order: a batch is inserted against the
    graph of the batches before it, so the builder inserts in a pseudo-random permutation instead of storage order (in storage order this
    corpus reached recall@10 0.2)."""
    n, d, M = 60000, 64, 16
    X = synth(po, n, d, n_clusters=64)
    Q = synth(po, 300, d, stream=1, n_clusters=64)
    centre = np.argmax(X @ X[:64].T, axis=1)          # a cheap cluster label: the closest of the first 64 rows
    Xs = np.ascontiguousarray(X[np.argsort(centre, kind="stable")])
    rec = []
    for rows in (X, Xs):
        dX = la.DeviceArray.from_host(rows)
        s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, 100)
        k, _, _ = s.search_batch(Q, 10, 64)
        rec.append(recall_at_k(k, po.exact_topk(rows, Q, 10)))
        s.close()
    assert rec[0] >= 0.95 and rec[1] >= 0.95 and abs(rec[0] - rec[1]) <= 0.03, rec


def test_build_is_reproducible(la, po, gpu):
    X = synth(po, 30000, 96)
    dX = la.DeviceArray.from_host(X)
    g = []
    for _ in range(2):
        s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, 30000, 96, 96, 16, 64)
        g.append(s.graph_export())
        s.close()
    assert (np.sort(g[0]["adj0"], axis=1) == np.sort(g[1]["adj0"], axis=1)).all() and g[0]["entry"] == g[1]["entry"]
    assert (g[0]["adjU"] == g[1]["adjU"]).all()


@pytest.mark.parametrize("nw", ["4", "8", "16"])
def test_waves_per_query_do_not_change_results(la, po, gpu, monkeypatch, nw):
    """Small batches use 16 waves per query, throughput batches 4: same expansion order, same sums."""
    X = synth(po, 5000, 768)
    Q = synth(po, 40, 768, stream=1)
    G = po.Graph.build_hnsw(X, M=16, efc=48)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 16, 32, G.max_level, G.entry, lv, uo, a0, aU)
    monkeypatch.setenv("LEANN_DEBUG_NW", nw)
    la.lib().leann_debug_reload_env()
    ok, od, oc, ost = G.search_batch(Q, 10, 64, 0, 4)
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, 10, 64)
    assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert s.stats()["n_dist_evals"] == int(ost[:, 0].sum())
    s.close()


def test_request_coalescing_for_single_query_callers(la, po, gpu):
    """64 threads each issuing single-query search() calls (the serve.rs pattern) are answered by a few
    batched launches with identical results."""
    import time
    X = synth(po, 30000, 128)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, 30000, 128, 128, 16, 64)
    Q = synth(po, 64 * 20, 128, stream=1)
    ref_k, ref_d, _ = s.search_batch(Q, 10, 64)
    out = {}

    def work(t):
        for j in range(20):
            i = t * 20 + j
            out[i] = s.search(Q[i], 10, 64)

    def run():
        out.clear()
        th = [threading.Thread(target=work, args=(t,)) for t in range(64)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0

    # the handle as opened: a lone caller is answered directly (no dispatcher exists yet) ...
    assert (s.search(Q[3], 10, 64)[0] == ref_k[3]).all()
    with pytest.raises(la.LeannError):
        s.coalescing_stats()
    # ... concurrent callers install one by themselves (automatic mode) and are answered in batches, same results
    t_auto = run()
    for i in range(len(Q)):
        assert (out[i][0] == ref_k[i]).all() and (out[i][1] == ref_d[i]).all()
    st_auto = s.coalescing_stats()
    assert 0 < st_auto["queries"] <= len(Q) and st_auto["launches"] < st_auto["queries"]
    s.set_coalescing(0, 0)  # switched off: one launch per caller
    t_plain = run()
    for i in range(len(Q)):
        assert (out[i][0] == ref_k[i]).all() and (out[i][1] == ref_d[i]).all()
    with pytest.raises(la.LeannError):
        s.coalescing_stats()
    s.set_coalescing(300, 1024)
    t_coal = run()
    for i in range(len(Q)):
        assert (out[i][0] == ref_k[i]).all() and (out[i][1] == ref_d[i]).all()
    st = s.coalescing_stats()
    assert st["queries"] == len(Q) and st["launches"] < len(Q) / 4
    # mixed (top_k, complexity) in one window are split into separate launches
    r1, r2 = {}, {}
    th = [threading.Thread(target=lambda i=i: r1.__setitem__(i, s.search(Q[i], 5, 32))) for i in range(16)]
    th += [threading.Thread(target=lambda i=i: r2.__setitem__(i, s.search(Q[i], 10, 64))) for i in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    k5, d5, _ = s.search_batch(Q[:16], 5, 32)
    for i in range(16):
        assert (r1[i][0] == k5[i]).all() and (r2[i][0] == ref_k[i]).all()
    s.set_coalescing(0, 0)
    assert (s.search(Q[3], 10, 64)[0] == ref_k[3]).all()
    print(f"single-query callers: as opened {len(Q)/t_auto:.0f} q/s ({st_auto['launches']} launches for {st_auto['queries']} queued queries), "
          f"plain {len(Q)/t_plain:.0f} q/s, coalesced {len(Q)/t_coal:.0f} q/s, launches {st['launches']}")
    s.close()


def test_visited_set_never_runs_out(la, po, gpu, monkeypatch):
    """a query that outgrew even its pooled HBM table used to come back EMPTY with LEANN_OK.  Now it moves on to a
    second-level table sized to hold every node of the index; results and counters stay identical to the oracle's.  The debug knobs
    shrink the pools so that a 60k-row index exercises LDS -> pool 1 -> pool 2; with pool 2 shrunk below the index size the defensive
    path must report LEANN_ERR_OVERFLOW instead of an empty answer."""
    n, d, M = 60000, 64, 16
    X = synth(po, n, d, r=0)  # i.i.d.: searches wander
    Q = synth(po, 128, d, stream=1, r=0)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, 32)
    g = s.graph_export()
    G = po.Graph.from_arrays(X, M, 2 * M, g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    ok, od, oc, ost = G.search_batch(Q, 10, 400, 0, nthreads=8)
    assert int(ost[:, 0].min()) > 3500  # every query visits far more than a 2^10-slot (or 2^12-slot) table holds
    s.close()
    monkeypatch.setenv("LEANN_DEBUG_HASH_BITS", "8")     # LDS table: 256 slots
    la.lib().leann_debug_reload_env()
    monkeypatch.setenv("LEANN_DEBUG_GPOOL_BITS", "10")   # pool 1: 1 024 slots -> pool 2 (sized by n)
    la.lib().leann_debug_reload_env()
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, 32)  # (the pools are sized at first use, per handle)
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, 10, 400)
    st = s.stats()
    assert (gc == oc).all() and (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert st["n_dist_evals"] == int(ost[:, 0].sum()) and st["n_table_overflow"] == len(Q)
    k1, d1 = s.search(Q[7], 10, 400)  # 16-wave single-query mode through the same escalation
    assert (k1 == ok[7]).all() and (d1 == od[7]).all()
    s.close()
    monkeypatch.setenv("LEANN_DEBUG_GPOOL2_BITS", "12")  # defensive path: even the last level is too small
    la.lib().leann_debug_reload_env()
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, 32)
    with pytest.raises(la.LeannError) as e:
        s.search_batch(Q, 10, 400)
    assert e.value.code == 7 and "visited-set space" in str(e.value)
    gk, gd, gc = s.search_batch(Q, 10, 8)  # a narrow beam on the same handle still fits and still matches
    ok8, od8, oc8, _ = G.search_batch(Q, 10, 8, 0, nthreads=8)
    assert (gk == ok8).all() and (gd.view(np.uint32) == od8.view(np.uint32)).all()
    s.close()


def test_coalescing_can_be_reconfigured_while_searches_are_in_flight(la, po, gpu):
    """set_coalescing used to hold the handle's mutex while joining the dispatcher (deadlock with a pending launch) and
    deleted the object waiters were blocked on.  Now: reconfigure / disable from one thread while 32 threads search."""
    import time
    X = synth(po, 20000, 64)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, 20000, 64, 64, 16, 64)
    Q = synth(po, 32 * 40, 64, stream=1)
    ref_k, ref_d, _ = s.search_batch(Q, 10, 48)
    bad, stop = [], threading.Event()

    def work(t):
        for j in range(40):
            i = t * 40 + j
            k, dd = s.search(Q[i], 10, 48)
            if not ((k == ref_k[i]).all() and (dd == ref_d[i]).all()):
                bad.append(i)

    def flip():
        modes = [(200, 64), (0, 0), (50, 8), (1000, 4096), (0, 0), (300, 16)]
        j = 0
        while not stop.is_set():
            s.set_coalescing(*modes[j % len(modes)])
            j += 1
            time.sleep(0.002)

    th = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    fl = threading.Thread(target=flip)
    fl.start()
    [t.start() for t in th]
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive(), "a searcher thread is stuck (deadlock between set_coalescing and a pending launch)"
    stop.set()
    fl.join(timeout=30)
    assert not fl.is_alive() and not bad
    s.set_coalescing(0, 0)
    s.close()


def test_recycled_scratch_is_never_read_before_it_is_written(la, po, gpu):
    """DESIGN.md §7 (the hipMallocAsync episode): scratch blocks of the exact scan come back from a free list with their previous
    contents.  Alternate calls of very different shapes — every later call sees blocks dirtied by a different problem — and check
    each against the oracle: a read of a word not written in the same call would surface here as a wrong key."""
    d = 96
    X = synth(po, 70_000, d)
    dX = la.DeviceArray.from_host(X)
    shapes = [(70_000, 64, 10), (3_000, 3, 2), (66_000, 17, 64), (2_049, 1, 1), (70_000, 5, 100), (4_100, 64, 10), (70_000, 64, 10)]
    for n, nq, k in shapes:
        Q = synth(po, nq, d, stream=1, i0=n)
        dQ = la.DeviceArray.from_host(Q)
        dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
        la._native.check(la.lib().leann_scan_topk_device(dX.ptr, n, d, d, dQ.ptr, nq, k, None, 0, dk.ptr, ds.ptr, dc.ptr, None))
        la.sync()
        gk, gs = dk.to_host(), ds.to_host()
        for i in range(0, nq, max(1, nq // 4)):
            k0, s0 = po.scan_topk(X[:n], Q[i], k, mode=1)
            assert (gk[i] == k0).all() and (gs[i] == s0).all(), (n, nq, k, i)
