"""CPU suite: the N > 1 path (partitioning, key rebasing, all-gather exchange, merge order) with
world_size 2 and 3 over the gloo backend.  Local search and merge are the ORACLE here (there is no GPU
in this container); on the GPU box tests/test_gpu_shard.py runs the same logic with the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, nq, k, ef, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import pyoracle as po
    import leann_rs_amd as la
    from leann_rs_amd.shard import ShardedSearcher, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, world, rank)
    X = po.gen_rows(0x5EED0001, d, 32, 64, 1.0, 0, lo, hi - lo)  # this rank's rows of the global corpus
    Q = po.gen_rows(0x5EED0001, d, 32, 64, 1.0, 1, 0, nq)
    G = po.Graph.build_hnsw(X, M=8, efc=32)

    def local_search(queries, top_k, complexity, stream):
        kk, dd, cc, _ = G.search_batch(queries.numpy(), top_k, complexity, 0, 1)
        kk = kk + np.uint64(lo)  # key rebasing == key_offset of the C ABI
        kk[dd == np.inf] = np.iinfo(np.uint64).max
        return (torch.from_numpy(kk.view(np.int64)), torch.from_numpy(dd), torch.from_numpy(cc.view(np.int32)))

    def merge(keys, dists, counts, k_out, descending, stream):
        S, n, _ = keys.shape
        ok = np.full((n, k_out), np.iinfo(np.uint64).max, np.uint64)
        od = np.full((n, k_out), np.inf, np.float32)
        oc = np.zeros(n, np.int32)
        kn, dn, cn = keys.numpy().view(np.uint64), dists.numpy(), counts.numpy().view(np.uint32)
        for q in range(n):
            mk, md = po.merge_topk(kn[:, q], dn[:, q], cn[:, q], k_out)
            ok[q, :len(mk)], od[q, :len(mk)], oc[q] = mk, md, len(mk)
        return torch.from_numpy(ok.view(np.int64)), torch.from_numpy(od), torch.from_numpy(oc)

    ss = ShardedSearcher(None, n_total, world, rank, local_search=local_search, merge=merge)
    assert (ss.lo, ss.hi) == (lo, hi) and ss.len() == n_total
    keys, dists, counts = ss.search_batch(torch.from_numpy(Q), k, ef)
    # pipelined form: the exchange of batch i is in flight while batch i + 1 is searched locally — same answers, in order
    parts = [torch.from_numpy(Q[:7]), torch.from_numpy(Q[7:8]), torch.from_numpy(Q[8:])]
    out = list(ss.search_batches(parts, k, ef))
    assert len(out) == 3
    pk = torch.cat([o[0] for o in out]); pd = torch.cat([o[1] for o in out]); pc = torch.cat([o[2] for o in out])
    assert torch.equal(pk, keys) and torch.equal(pd, dists) and torch.equal(pc, counts)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), keys=keys.numpy().view(np.uint64), dists=dists.numpy(),
             counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_gloo(world, tmp_path, po):
    from leann_rs_amd.shard import shard_range
    n_total, d, nq, k, ef = 3001, 48, 24, 10, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, d, nq, k, ef, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in res[1:]:  # every rank holds the same merged answer
        assert (r["keys"] == res[0]["keys"]).all() and (r["dists"] == res[0]["dists"]).all()
    # simulated shards in ONE process: same sub-indexes searched serially, same merge
    Q = po.gen_rows(0x5EED0001, d, 32, 64, 1.0, 1, 0, nq)
    per_k, per_d, per_c = [], [], []
    ranges = [shard_range(n_total, world, g) for g in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    Xall = po.gen_rows(0x5EED0001, d, 32, 64, 1.0, 0, 0, n_total)
    for lo, hi in ranges:
        G = po.Graph.build_hnsw(Xall[lo:hi], M=8, efc=32)
        kk, dd, cc, _ = G.search_batch(Q, k, ef, 0, 1)
        per_k.append(kk + np.uint64(lo)); per_d.append(dd); per_c.append(cc)
    per_k, per_d, per_c = np.stack(per_k), np.stack(per_d), np.stack(per_c)
    for q in range(nq):
        mk, md = po.merge_topk(per_k[:, q], per_d[:, q], per_c[:, q], k)
        assert (res[0]["keys"][q, :len(mk)] == mk).all() and (res[0]["dists"][q, :len(mk)] == md).all()
        assert res[0]["counts"][q] == len(mk)
    # sharded ANN recall against exact search over the union
    truth = po.exact_topk(Xall, Q, k)
    hits = sum(len(set(res[0]["keys"][q].tolist()) & set(truth[q].tolist())) for q in range(nq))
    assert hits / (nq * k) >= 0.9
