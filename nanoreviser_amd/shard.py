"""Read / window sharding across the GPUs of a node.

The reference parallelises over fast5 files with a multiprocessing.Pool
(NanoReviser.py:203-219); windows and reads are independent (zero initial LSTM state per
window), so the MI355X node is fed the same way: one process per GPU, each owning a shard,
with NO data-path collective.  Unlike NanoReviser.py:212, no file is dropped
when the count is not a multiple of the pool size.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n units for `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(max(n, 0), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_reads(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first greedy balancing of reads (by base count) over `world` workers.
    Returns, per worker, the indices of its reads.  Every read lands on exactly one worker."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        w = min(range(world), key=lambda k: (loads[k], k))
        out[w].append(i)
        loads[w] += int(sizes[i])
    return out


def split_read_windows(n_events: int, T: int, parts: int) -> List[Tuple[int, int]]:
    """Split ONE long read's window range [0, N-T) into `parts` event slices with a (T-1)-event
    halo, so each slice can be revised independently: returns [(ev_lo, ev_hi)] such that the
    windows of slice k are exactly windows [w_lo, w_hi) of the read, ev_lo = w_lo,
    ev_hi = w_hi + T - 1 (+1: nrv_predict_read yields N-T windows for N events)."""
    n = max(n_events - T, 0)
    out = []
    for k in range(parts):
        lo, hi = shard_range(n, k, parts)
        out.append((lo, hi + T) if hi > lo else (lo, lo))
    return out
