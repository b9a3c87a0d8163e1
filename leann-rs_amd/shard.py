"""Corpus sharding across GPUs (new — the reference is single-process).

One process per GPU.  The corpus is split into contiguous position ranges; rank g owns rows
[g*N/G, (g+1)*N/G) with its own graph and returns keys rebased by its range start, so keys stay
global positions into ids.txt (src/index/searcher.rs:180-184).  Every rank searches the same query
batch; the only exchange step is one all-gather of the per-shard top-k lists (RCCL over xGMI when
the process group backend is "nccl"), followed by the G-way merge kernel on every rank — ordered by
(dist, key), hence independent of the rank count.

torch is plumbing here (device tensors + the process group); search and merge are HIP kernels
behind the C ABI.  `local_search` / `merge` can be injected so that the partition / exchange logic
is testable on CPU with the gloo backend (tests/test_shard_gloo.py uses the oracle for both).
"""
import ctypes as C

import torch

from . import _native as N


def shard_range(n_total, world, rank):
    """contiguous range [lo, hi) of positions owned by `rank`"""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def _hip_local_search(searcher):
    def fn(queries, top_k, complexity, stream):
        if not queries.is_cuda:
            raise RuntimeError("ShardedSearcher: the HIP search path needs CUDA/HIP tensors (no CPU fallback)")
        nq = queries.shape[0]
        dev = queries.device
        keys = torch.empty((nq, top_k), dtype=torch.int64, device=dev)
        dists = torch.empty((nq, top_k), dtype=torch.float32, device=dev)
        counts = torch.empty((nq,), dtype=torch.int32, device=dev)
        searcher.search_batch_device(queries.data_ptr(), nq, top_k, complexity, keys.data_ptr(), dists.data_ptr(),
                                     counts.data_ptr(), None, C.c_void_p(stream))
        return keys, dists, counts
    return fn


def _hip_merge(keys, dists, counts, k_out, descending, stream):
    if not keys.is_cuda:
        raise RuntimeError("ShardedSearcher: the HIP merge kernel needs CUDA/HIP tensors (no CPU fallback)")
    S, nq, k_in = keys.shape
    dev = keys.device
    ok = torch.empty((nq, k_out), dtype=torch.int64, device=dev)
    od = torch.empty((nq, k_out), dtype=torch.float32, device=dev)
    oc = torch.empty((nq,), dtype=torch.int32, device=dev)
    N.check(N.lib().leann_merge_topk_device(keys.data_ptr(), dists.data_ptr(), counts.data_ptr(), S, nq, k_in, k_out,
                                            1 if descending else 0, ok.data_ptr(), od.data_ptr(), oc.data_ptr(),
                                            C.c_void_p(stream)))
    return ok, od, oc


def start_exchange(keys, dists, counts, world, group=None):
    """The exchange of `exchange_topk`, started asynchronously: returns a handle for `finish_exchange`.  Over RCCL the all-gather
    runs on the process group's own stream behind the work already queued on the current one, so a search queued between start and
    finish overlaps it (the exchange is latency-bound: 12 B per entry)."""
    import torch.distributed as dist
    nq, k = keys.shape
    pack = torch.empty((nq, k, 3), dtype=torch.int32, device=keys.device)
    pack[..., 0:2] = keys.contiguous().view(torch.int32).view(nq, k, 2)
    pack[..., 2] = dists.contiguous().view(torch.int32)
    cnt = counts.contiguous()
    gathered = torch.empty((world * nq, k, 3), dtype=torch.int32, device=keys.device)
    cnt_all = torch.empty((world * nq,), dtype=torch.int32, device=keys.device)
    works = (dist.all_gather_into_tensor(gathered, pack, group=group, async_op=True),
             dist.all_gather_into_tensor(cnt_all, cnt, group=group, async_op=True))
    return works, gathered, cnt_all, pack, cnt, (world, nq, k)


def finish_exchange(handle):
    works, gathered, cnt_all, _pack, _cnt, (world, nq, k) = handle
    for w in works:
        w.wait()  # RCCL: the current stream waits for the collective (the host does not block); gloo: blocks until done
    g_keys = gathered[..., 0:2].contiguous().view(torch.int64).view(world, nq, k)
    g_dists = gathered[..., 2].contiguous().view(torch.float32).view(world, nq, k)
    return g_keys, g_dists, cnt_all.view(world, nq)


def exchange_topk(keys, dists, counts, world, group=None):
    """all-gather of per-shard lists: [nq,k] x3 -> [world,nq,k] x2 + [world,nq].
    One packed int32 buffer {key lo, key hi, dist bits} per rank + the counts (8 B + 4 B per entry:
    64 queries x 10 = 7.7 KB per GPU, latency-bound)."""
    import torch.distributed as dist
    nq, k = keys.shape
    pack = torch.empty((nq, k, 3), dtype=torch.int32, device=keys.device)
    pack[..., 0:2] = keys.contiguous().view(torch.int32).view(nq, k, 2)
    pack[..., 2] = dists.contiguous().view(torch.int32)
    gathered = torch.empty((world * nq, k, 3), dtype=torch.int32, device=keys.device)  # concatenated along dim 0
    cnt_all = torch.empty((world * nq,), dtype=torch.int32, device=keys.device)
    dist.all_gather_into_tensor(gathered, pack, group=group)
    dist.all_gather_into_tensor(cnt_all, counts.contiguous(), group=group)
    g_keys = gathered[..., 0:2].contiguous().view(torch.int64).view(world, nq, k)
    g_dists = gathered[..., 2].contiguous().view(torch.float32).view(world, nq, k)
    return g_keys, g_dists, cnt_all.view(world, nq)


class ShardedSearcher:
    """BackendSearcher over a corpus partitioned across the ranks of a process group."""

    def __init__(self, searcher, n_total, world, rank, group=None, local_search=None, merge=None):
        self.searcher = searcher
        self.n_total, self.world, self.rank, self.group = n_total, world, rank, group
        self.lo, self.hi = shard_range(n_total, world, rank)
        self._local = local_search or _hip_local_search(searcher)
        self._merge = merge or _hip_merge

    def len(self):
        return self.n_total

    def search_batches(self, batches, top_k, complexity, descending=False):
        """Pipelined form of search_batch over an iterable of query batches (each identical on every rank): the all-gather of
        batch i is in flight while the local search of batch i + 1 runs; yields the same (keys, dists, counts) as search_batch,
        in order, one batch behind."""
        pending = None
        for queries in batches:
            stream = torch.cuda.current_stream(queries.device).cuda_stream if queries.is_cuda else 0
            keys, dists, counts = self._local(queries, top_k, complexity, stream)
            if pending is not None:
                yield self._finish(pending, top_k, descending)
            pending = (keys, dists, counts) if self.world == 1 else start_exchange(keys, dists, counts, self.world, self.group)
            pending = (pending, stream)
        if pending is not None:
            yield self._finish(pending, top_k, descending)

    def _finish(self, pending, top_k, descending):
        handle, stream = pending
        if self.world == 1:
            return handle
        g_keys, g_dists, g_counts = finish_exchange(handle)
        return self._merge(g_keys, g_dists, g_counts, top_k, descending, stream)

    def search_batch(self, queries, top_k, complexity, descending=False):
        """queries: [nq, dims] tensor, identical on every rank.  Returns global (keys, dists, counts)."""
        stream = torch.cuda.current_stream(queries.device).cuda_stream if queries.is_cuda else 0
        keys, dists, counts = self._local(queries, top_k, complexity, stream)
        if self.world == 1:
            return keys, dists, counts
        g_keys, g_dists, g_counts = exchange_topk(keys, dists, counts, self.world, self.group)
        return self._merge(g_keys, g_dists, g_counts, top_k, descending, stream)
