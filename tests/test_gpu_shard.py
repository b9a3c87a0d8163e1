"""-m gpu: simulated shards on one device: G sub-indexes built and searched on the
GPU, merged by the HIP merge kernel, against (a) the oracle doing the same and (b) exact search."""
import numpy as np
import pytest

from util import recall_at_k, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_simulated_shards_match_oracle(la, po, gpu, G):
    from leann_rs_amd.shard import shard_range
    n, d, nq, k, ef, M = 8000, 128, 40, 10, 48, 12
    X = synth(po, n, d)
    Q = synth(po, nq, d, stream=1)
    gk, gd, gc, ok, od, oc = [], [], [], [], [], []
    for g in range(G):
        lo, hi = shard_range(n, G, g)
        dX = la.DeviceArray.from_host(X[lo:hi])
        s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, hi - lo, d, d, M, 48, key_offset=lo)
        kk, dd, cc = s.search_batch(Q, k, ef)
        gr = s.graph_export()
        Gr = po.Graph.from_arrays(X[lo:hi], M, 2 * M, gr["max_level"], gr["entry"], gr["levels"], gr["upper_off"],
                                  gr["adj0"], gr["adjU"])
        k0, d0, c0, _ = Gr.search_batch(Q, k, ef, 0, 4)
        assert (kk == k0 + np.uint64(lo)).all() and (dd == d0).all()  # key rebasing
        gk.append(kk); gd.append(dd); gc.append(cc)
        ok.append(k0 + np.uint64(lo)); od.append(d0); oc.append(c0)
        s.close()
    gk, gd, gc = np.stack(gk), np.stack(gd), np.stack(gc)
    dk, dd_, dc = la.DeviceArray.from_host(gk), la.DeviceArray.from_host(gd), la.DeviceArray.from_host(gc)
    mk, md, mc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    la._native.check(la.lib().leann_merge_topk_device(dk.ptr, dd_.ptr, dc.ptr, G, nq, k, k, 0, mk.ptr, md.ptr, mc.ptr, None))
    la.sync()
    mk, md = mk.to_host(), md.to_host()
    for q in range(nq):
        rk, rd = po.merge_topk(np.stack(ok)[:, q], np.stack(od)[:, q], np.stack(oc)[:, q], k)
        assert (mk[q] == rk).all() and (md[q] == rd).all()
    assert recall_at_k(mk, po.exact_topk(X, Q, k)) >= 0.93
