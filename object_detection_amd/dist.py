"""Image-sharded multi-GPU execution: one process per GPU, the batch split by image index, and the
single exchange step of the path -- an all-gather of the per-image result rows so that every rank
holds the full result list that `Evaluator.dump_pred` consumes (src/utilities/evaluator.py:76-79,
fed by src/test.py:99).  The reference itself is single-device; this module has no counterpart there.

Rows travel as int64 [N,2] (image_id, category_id) + float64 [N,5] (x, y, w, h, score).  Counts are
gathered first, payloads are padded to the largest shard (RCCL all_gather wants equal sizes), and the
result is concatenated in rank order, which is image order because shards are contiguous image ranges.
Messages are a few KB..MB, so the step is latency-bound and uses exactly two collectives.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(num_images: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous image range [lo, hi) of `rank`; the first `num_images % world_size` ranks get one extra."""
    base, extra = divmod(num_images, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_list(items: list, rank: int, world_size: int) -> list:
    lo, hi = shard_range(len(items), rank, world_size)
    return items[lo:hi]


def gather_detections(rows_ids: torch.Tensor, rows_vals: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks contribute their (rows_ids [n_r,2] i64, rows_vals [n_r,5] f64); all ranks receive the
    concatenation over ranks in rank order."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows_ids, rows_vals
    world = dist.get_world_size(group)
    dev = rows_ids.device
    n = torch.tensor([rows_ids.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    # one payload: ids bit-cast next to the values so that a single all_gather moves both
    pay = torch.zeros((n_max, 7), dtype=torch.float64, device=dev)
    pay[:rows_ids.shape[0], :2] = rows_ids.view(torch.float64) if rows_ids.numel() else rows_ids.to(torch.float64)
    pay[:rows_vals.shape[0], 2:] = rows_vals
    out = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(out, pay, group=group)
    ids = torch.cat([o[:c, :2].contiguous().view(torch.int64) for o, c in zip(out, counts)])
    vals = torch.cat([o[:c, 2:] for o, c in zip(out, counts)])
    return ids, vals
