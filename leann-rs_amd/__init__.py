"""leann-rs_amd — MI355X (gfx950) native ANN search path behind leann-rs's backend surface.

Layout: csrc/ (HIP kernels + C ABI, built into csrc/libleann_hip.so), host/ (C++ mirror of the
reference's index layer + `leann search` CLI), and this thin ctypes mirror used by tests and bench.
"""
from ._native import LeannError, lib, device_count, LIB_PATH  # noqa: F401
from .backend import (BackendBuilder, BackendSearcher, BackendType, DiskAnnSearcher,  # noqa: F401
                      HnswSearcher, ShardedIndex)
from .device import DeviceArray, sync  # noqa: F401


def __getattr__(name):  # torch is only needed for the multi-GPU helpers: import them lazily
    if name in ("ShardedSearcher", "shard_range", "exchange_topk"):
        from . import shard
        return getattr(shard, name)
    raise AttributeError(name)
