"""neumann_amd — MI355X-native SIMILAR TOP-K for Neumann's `vector_engine`.

Only what the hot path needs: `csrc/` (HIP kernels + C ABI), the ctypes binding of that ABI, the
host-side mirror of the reference's `VectorEngine` interface for this path, and the row-range
sharding layer (one process per GPU, RCCL all-gather of per-shard top-k).
"""
from ._capi import NeumannGpuError, load as load_library  # noqa: F401
from .flat_index import (DistanceMetric, GpuFlatIndex, merge_topk_device, merge_topk_device_packed,  # noqa: F401
                         merge_topk_host, packed_layout, synth_rows)

from .sharded import GpuShardedIndex  # noqa: F401,E402

__version__ = "0.3.0"
