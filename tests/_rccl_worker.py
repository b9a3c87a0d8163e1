"""Worker of tests/test_gpu_rccl.py: one rank of an RCCL group, one rank per visible GPU.  The sharded search runs in the LIBRARY
(csrc/shard.hip: local traversal + ncclAllGather of the packed per-shard block + merge kernel); torch only carries the 128-byte
RCCL id to the other ranks and, at the end, the results to rank 0 for comparison."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out):
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import leann_rs_amd as la
    from leann_rs_amd.shard import ShardedSearcher, rccl_group, shard_range
    L, chk = la.lib(), la._native.check
    n, d, nq, k, ef = 6000 * world, 128, 48, 10, 64
    g = torch.Generator(device="cpu").manual_seed(7)
    centres = torch.randn(96, d, generator=g)  # clustered rows (i.i.d. 128-d Gaussians are adversarial for any graph index)
    X = torch.nn.functional.normalize(centres[torch.randint(0, 96, (n,), generator=g)] + 0.35 * torch.randn(n, d, generator=g), dim=1).to(dev)
    Q = torch.nn.functional.normalize(centres[torch.randint(0, 96, (nq,), generator=g)] + 0.35 * torch.randn(nq, d, generator=g), dim=1).to(dev)
    # (the whole corpus on every rank — it is small — serves as ground truth)
    lo, hi = shard_range(n, world, rank)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, X[lo:hi].contiguous().data_ptr(), hi - lo, d, d, 12, 48, device=local,
                                        key_offset=lo, take_copy=True)
    lk, ld_, lc = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                   torch.empty((nq,), dtype=torch.int32, device=dev))
    st = torch.cuda.current_stream(dev)
    s.search_batch_device(Q.data_ptr(), nq, k, ef, lk.data_ptr(), ld_.data_ptr(), lc.data_ptr(), None, C.c_void_p(st.cuda_stream))
    # the library's RCCL group (a world of one rank still goes through ncclCommInitRank + ncclAllGather + the merge kernel)
    grp = rccl_group(s, n, world, rank)
    assert grp.len() == n and grp.n_shards() == world
    outs = [(torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(3)]
    grp.search_batch_device(Q.data_ptr(), nq, k, ef, outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr(), None, C.c_void_p(st.cuda_stream))
    t1 = grp.search_batch_device_async(Q.data_ptr(), nq, k, ef, outs[1][0].data_ptr(), outs[1][1].data_ptr(), outs[1][2].data_ptr(), None, C.c_void_p(st.cuda_stream))
    t2 = grp.search_batch_device_async(Q.data_ptr(), nq, k, ef, outs[2][0].data_ptr(), outs[2][1].data_ptr(), outs[2][2].data_ptr(), None, C.c_void_p(st.cuda_stream))
    grp.wait(t1, C.c_void_p(st.cuda_stream))
    grp.wait(t2, C.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    keys, dists, counts = outs[0]
    if world == 1:  # one shard: the exchange is the identity
        assert torch.equal(keys, lk) and torch.equal(dists, ld_) and torch.equal(counts, lc)
    # ShardedSearcher, the thin caller (world > 1: the same library group; world 1: local search only)
    ss = ShardedSearcher(s, n, world, rank)
    k2, d2, c2 = ss.search_batch(Q, k, ef)
    parts = list(ss.search_batches([Q[:16], Q[16:]], k, ef))
    torch.cuda.synchronize()
    assert torch.equal(k2, keys) and torch.equal(d2, dists) and torch.equal(torch.cat([p[0] for p in parts]), keys)
    # exact ground truth over the whole corpus (scan kernel) -> recall of the sharded ANN answer
    gk, gs, gc = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                  torch.empty((nq,), dtype=torch.int32, device=dev))
    chk(L.leann_scan_topk_device(X.data_ptr(), n, d, d, Q.data_ptr(), nq, k, None, 0, gk.data_ptr(), gs.data_ptr(), gc.data_ptr(), C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    kn, tn = keys.cpu().numpy(), gk.cpu().numpy()
    recall = float(np.mean([len(set(kn[i].tolist()) & set(tn[i].tolist())) / k for i in range(nq)]))
    # every rank must hold the same merged answer
    allk = [torch.empty_like(keys) for _ in range(world)]
    dist.all_gather(allk, keys)
    same = all(torch.equal(a, keys) for a in allk)
    dist.barrier()
    if rank == 0:
        np.savez(out, keys=kn, dists=dists.cpu().numpy(), counts=counts.cpu().numpy(), recall=recall, same=same, world=world,
                 local_keys=lk.cpu().numpy())
    ss.close()
    grp.close()
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
