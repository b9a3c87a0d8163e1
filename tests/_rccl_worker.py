"""Worker of tests/test_gpu_rccl.py: one rank of an RCCL ("nccl") process group on the box's GPU.  Runs the exchange step of the
sharded search (leann-rs_amd/shard.py: packed all-gather + HIP merge kernel) and writes what it got."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    import leann_rs_amd as la
    from leann_rs_amd.shard import ShardedSearcher, exchange_topk, _hip_merge
    n, d, nq, k, ef = 6000, 128, 48, 10, 48
    g = torch.Generator(device="cpu").manual_seed(7)
    X = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(dev)
    Q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).to(dev)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, X.data_ptr(), n, d, d, 12, 48)
    ss = ShardedSearcher(s, n, world, rank)
    keys, dists, counts = ss.search_batch(Q, k, ef)  # world 1: no exchange
    torch.cuda.synchronize()
    # the exchange step itself, over RCCL: gather (a world of one rank returns its own lists), merge with the HIP kernel
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(stream):
        gk, gd, gc = exchange_topk(keys, dists, counts, world)
        mk, md, mc = _hip_merge(gk, gd, gc, k, False, stream.cuda_stream)
    stream.synchronize()
    # the asynchronous form used by ShardedSearcher.search_batches: start, queue another search behind it, finish
    from leann_rs_amd.shard import start_exchange, finish_exchange
    hnd = start_exchange(keys, dists, counts, world)
    k2, d2, c2 = ss.search_batch(Q, k, ef)             # overlaps the all-gather on the process group's stream
    ak, ad, ac = finish_exchange(hnd)
    torch.cuda.synchronize()
    assert torch.equal(ak[0], keys) and torch.equal(ad[0], dists) and torch.equal(ac[0], counts)
    assert torch.equal(k2, keys) and torch.equal(d2, dists)
    outs = list(ss.search_batches([Q[:16], Q[16:]], k, ef))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o[0] for o in outs]), keys)
    dist.barrier()
    assert gk.shape == (world, nq, k) and gc.shape == (world, nq)
    np.savez(out, keys=keys.cpu().numpy(), dists=dists.cpu().numpy(), counts=counts.cpu().numpy(), gk=gk.cpu().numpy(),
             gd=gd.cpu().numpy(), gc=gc.cpu().numpy(), mk=mk.cpu().numpy(), md=md.cpu().numpy(), mc=mc.cpu().numpy())
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
