"""Host-side sharding of a batch of independent zlib streams over the GPUs of one node.

The reference has no parallelism of any kind; streams are independent, so the
multi-GPU form of decompressMany is a partition of the stream list: one process per GPU, every
rank decodes its own shard into its own output arena, and NO data-path collective exists (no RCCL
traffic, nothing crosses xGMI).  The only cross-rank communication is the caller's barrier/timing.
"""
from typing import List, Sequence

import numpy as np


def plan_shards(weights: Sequence[int], world: int) -> List[np.ndarray]:
    """Partition stream indices 0..n-1 into `world` shards balanced by `weights`
    (decoded bytes or capacities).  Longest-processing-time-first greedy; within a shard the
    indices are returned in descending weight so the longest streams launch first and the tail
    of the launch is filled by short ones.  Deterministic for equal inputs."""
    w = np.asarray(weights, dtype=np.int64)
    n = len(w)
    if world <= 0:
        raise ValueError("world must be positive")
    if world == 1:
        return [np.argsort(-w, kind="stable").astype(np.int64)]
    order = np.argsort(-w, kind="stable")
    if n and w.min() == w.max():
        # uniform batch: contiguous equal ranges (keeps each shard's arena one contiguous slice)
        bounds = [(n * r) // world for r in range(world + 1)]
        return [np.arange(bounds[r], bounds[r + 1], dtype=np.int64) for r in range(world)]
    loads = np.zeros(world, dtype=np.int64)
    buckets: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(loads))
        buckets[r].append(int(i))
        loads[r] += int(w[i])
    return [np.asarray(b, dtype=np.int64) for b in buckets]


def shard_imbalance(weights: Sequence[int], shards: List[np.ndarray]) -> float:
    """max shard load / mean shard load (1.0 = perfect)."""
    w = np.asarray(weights, dtype=np.int64)
    loads = np.array([int(w[s].sum()) for s in shards], dtype=np.float64)
    return float(loads.max() / max(loads.mean(), 1.0))
