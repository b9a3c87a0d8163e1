"""Corpus sharding across GPUs, one process per GPU (new — the reference is single-process).

The corpus is split into contiguous position ranges; rank g owns rows [g*N/G, (g+1)*N/G) with its own graph and returns keys
rebased by its range start, so keys stay global positions into ids.txt (src/index/searcher.rs:180-184).  Every rank searches the
same query batch; the only exchange step is one all-gather of the per-shard top-k lists, followed by the G-way merge kernel on
every rank — ordered by (dist, key), hence independent of the rank count.

The data path lives in the LIBRARY (csrc/shard.hip behind include/leann_backend.h "sharded indexes"): local traversal, ONE
ncclAllGather of the packed per-shard block over RCCL/xGMI, merge kernel, exchange overlapped with the next batch's traversal.
`ShardedSearcher` is a thin caller: torch supplies device tensors, the current stream and — once — the rendezvous that hands
rank 0's 128-byte RCCL id to the other ranks.

Two other ways through this module exist for machines without one GPU per rank:
  * `local_search` / `merge` injected (tests/test_shard_gloo.py: the oracle for both, gloo on CPU) — partition / exchange logic only;
  * CUDA tensors with a non-RCCL process group (LEANN_BENCH_DIST_BACKEND=gloo rehearsal: several ranks share one GPU, which RCCL
    refuses): HIP search + torch all-gather + HIP merge kernel.
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
    """torch-side exchange (gloo paths), started asynchronously: returns a handle for `finish_exchange`."""
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
        w.wait()
    g_keys = gathered[..., 0:2].contiguous().view(torch.int64).view(world, nq, k)
    g_dists = gathered[..., 2].contiguous().view(torch.float32).view(world, nq, k)
    return g_keys, g_dists, cnt_all.view(world, nq)


def exchange_topk(keys, dists, counts, world, group=None):
    """torch-side all-gather of per-shard lists: [nq,k] x3 -> [world,nq,k] x2 + [world,nq] (one packed int32 buffer
    {key lo, key hi, dist bits} per rank + the counts).  The RCCL path does this inside the library instead."""
    return finish_exchange(start_exchange(keys, dists, counts, world, group))


def rccl_group(searcher, n_total, world, rank, group=None):
    """attach `searcher` (this rank's shard) to a library-side RCCL group; the 128-byte id travels over the torch process group"""
    import torch.distributed as dist
    from .backend import ShardedIndex
    box = [ShardedIndex.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return ShardedIndex.attach(searcher, box[0], world, rank, n_total)


class ShardedSearcher:
    """BackendSearcher over a corpus partitioned across the ranks of a process group."""

    def __init__(self, searcher, n_total, world, rank, group=None, local_search=None, merge=None):
        self.searcher = searcher
        self.n_total, self.world, self.rank, self.group = n_total, world, rank, group
        self.lo, self.hi = shard_range(n_total, world, rank)
        self._injected = local_search is not None or merge is not None
        self._local = local_search or _hip_local_search(searcher)
        self._merge = merge or _hip_merge
        self._lib_group = None

    def len(self):
        return self.n_total

    def _library(self, queries):
        """the library's RCCL group, when this is the real thing: HIP tensors, world > 1, backend "nccl" (= RCCL on ROCm)"""
        if self._injected or self.world == 1 or not queries.is_cuda:
            return None
        import torch.distributed as dist
        if dist.get_backend(self.group) != "nccl":
            return None
        if self._lib_group is None:
            self._lib_group = rccl_group(self.searcher, self.n_total, self.world, self.rank, self.group)
        return self._lib_group

    @staticmethod
    def _outputs(queries, top_k):
        nq, dev = queries.shape[0], queries.device
        return (torch.empty((nq, top_k), dtype=torch.int64, device=dev), torch.empty((nq, top_k), dtype=torch.float32, device=dev),
                torch.empty((nq,), dtype=torch.int32, device=dev))

    def search_batches(self, batches, top_k, complexity, descending=False):
        """Pipelined form of search_batch over an iterable of query batches (each identical on every rank): the exchange of batch i
        is in flight while the local search of batch i + 1 runs; yields the same (keys, dists, counts) as search_batch, in order,
        one batch behind."""
        pending = None
        for queries in batches:
            stream = torch.cuda.current_stream(queries.device).cuda_stream if queries.is_cuda else 0
            lib = self._library(queries)
            if lib is not None and descending:
                raise RuntimeError("ShardedSearcher: the library's RCCL path merges ascending (dist, key); descending (recompute scores) "
                                   "lists go through leann_recompute_create_sharded or the torch exchange")
            if lib is not None:
                out = self._outputs(queries, top_k)
                ticket = lib.search_batch_device_async(queries.data_ptr(), queries.shape[0], top_k, complexity, out[0].data_ptr(),
                                                       out[1].data_ptr(), out[2].data_ptr(), None, C.c_void_p(stream))
                cur = ("lib", lib, ticket, out, stream, queries)
            else:
                keys, dists, counts = self._local(queries, top_k, complexity, stream)
                cur = ("torch", (keys, dists, counts) if self.world == 1 else start_exchange(keys, dists, counts, self.world, self.group), stream)
            if pending is not None:
                yield self._finish(pending, top_k, descending)
            pending = cur
        if pending is not None:
            yield self._finish(pending, top_k, descending)

    def _finish(self, pending, top_k, descending):
        if pending[0] == "lib":
            _, lib, ticket, out, stream, _q = pending
            lib.wait(ticket, C.c_void_p(stream))
            return out
        _, handle, stream = pending
        if self.world == 1:
            return handle
        g_keys, g_dists, g_counts = finish_exchange(handle)
        return self._merge(g_keys, g_dists, g_counts, top_k, descending, stream)

    def search_batch(self, queries, top_k, complexity, descending=False):
        """queries: [nq, dims] tensor, identical on every rank.  Returns global (keys, dists, counts)."""
        stream = torch.cuda.current_stream(queries.device).cuda_stream if queries.is_cuda else 0
        lib = self._library(queries)
        if lib is not None and descending:
            raise RuntimeError("ShardedSearcher: the library's RCCL path merges ascending (dist, key); descending (recompute scores) "
                               "lists go through leann_recompute_create_sharded or the torch exchange")
        if lib is not None:
            out = self._outputs(queries, top_k)
            lib.search_batch_device(queries.data_ptr(), queries.shape[0], top_k, complexity, out[0].data_ptr(), out[1].data_ptr(),
                                    out[2].data_ptr(), None, C.c_void_p(stream))
            return out
        keys, dists, counts = self._local(queries, top_k, complexity, stream)
        if self.world == 1:
            return keys, dists, counts
        g_keys, g_dists, g_counts = exchange_topk(keys, dists, counts, self.world, self.group)
        return self._merge(g_keys, g_dists, g_counts, top_k, descending, stream)

    def close(self):
        if self._lib_group is not None:
            self._lib_group.close()
            self._lib_group = None
