"""Frame orchestration shared by bench.py, the smoke check and the multi-process tests.

The reference's dispatcher splits a frame into row blocks, one per thread
(src/dispatcher.c:12,86).  Here a frame is split by SAMPLE RANGE, one range per GPU: every
rank renders all pixels for samples [rank*spp/world, (rank+1)*spp/world), the linear radiance
sums are added across ranks (they must be added before the 1/spp scale and the sqrt of
src/dispatcher.c:29-30, which are not linear), and rank 0 resolves the pixels.  Random draws
are keyed by (seed, pixel, sample), so the set of paths does not depend on the number of ranks.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def sample_partition(spp: int, world: int, rank: int) -> Tuple[int, int]:
    """(first sample, sample count) of `rank` out of `world`; the ranges tile [0, spp) exactly."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if spp < 0:
        raise ValueError("spp must be >= 0")
    begin = spp * rank // world
    end = spp * (rank + 1) // world
    return begin, end - begin


def render_frame(render_range: Callable[[int, int], None], reduce_sum: Optional[Callable[[], None]],
                 spp: int, world: int, rank: int) -> Tuple[int, int]:
    """One frame on this rank: render the rank's sample range into its accumulation buffer, then
    sum the buffers over ranks onto rank 0 (reduce_sum is None for a single rank)."""
    begin, count = sample_partition(spp, world, rank)
    render_range(begin, count)
    if world > 1 and reduce_sum is not None:
        reduce_sum()
    return begin, count
