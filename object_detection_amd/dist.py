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


# --------------------------------------------------------------------------------------------------
# Pipelined exchange: compact messages, gathered on a side stream while the next batch is computed
# --------------------------------------------------------------------------------------------------

def unpack_wire(msg: torch.Tensor, header_rows: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """One rank's message (int32 [header_rows + rows.., 6], see detbox_pack_wire) -> the result rows
    (rows_ids [n,2] i64, rows_vals [n,5] f64) exactly as detbox_pack_detections writes them."""
    words = msg.reshape(-1)
    n, B = int(words[0]), int(words[1])
    row_off = words[2:2 + B + 1].to(torch.int64)
    idw = words[2 + B + 1:2 + B + 1 + 2 * B].to(torch.int64).reshape(B, 2)
    ids = (idw[:, 0] & 0xffffffff) | (idw[:, 1] << 32)
    sizes = words[2 + 3 * B + 1:2 + 5 * B + 1].reshape(B, 2)
    rows = msg[header_rows:header_rows + n]
    img = torch.searchsorted(row_off[1:].contiguous(), torch.arange(n, device=msg.device), right=True)
    f = rows[:, 1:].contiguous().view(torch.float32)
    W, H = sizes[img, 1].to(torch.float64), sizes[img, 0].to(torch.float64)
    vals = torch.stack([f[:, 0].to(torch.float64) * W, f[:, 1].to(torch.float64) * H,
                        (f[:, 2] - f[:, 0]).to(torch.float64) * W, (f[:, 3] - f[:, 1]).to(torch.float64) * H,
                        f[:, 4].to(torch.float64)], dim=1)
    rid = torch.stack([ids[img], rows[:, 0].to(torch.int64)], dim=1)
    return rid, vals


class RowExchange:
    """All-gather of every batch's detections, one batch behind the computation.

    submit(wire, images_cap) hands over this rank's message for the batch just launched (its row count is
    still being computed on the device); the gather of the PREVIOUS batch is started at that moment, on a
    side stream, so its transfer overlaps the kernels of the current batch and the host never waits for
    the batch in flight.  finish() starts the last gather and waits for all of them.  Two collectives per
    batch: the counts (to agree on the padded length) and the messages.  Works on gloo/CPU tensors too
    (no streams there).

    root=None: every rank receives every message (all-gather).  root=r: only global rank r receives them
    (gather) -- what a single evaluator process needs (src/test.py:98-99 builds one list in one process);
    the other ranks only send, so the traffic per batch is one message per link instead of world of them.
    On those ranks the gathered buffer of a batch is None."""

    def __init__(self, group=None, keep_results: bool = True, root=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.root = root
        self.rank = dist.get_rank() if self.world > 1 else 0
        # RCCL moves device buffers; any other backend (gloo rehearsals) gets the message staged on the host
        self.device_collectives = self.world > 1 and dist.get_backend(group) == 'nccl'
        self.keep_results = keep_results
        self.waiting = []        # submitted, gather not started: (wire, header_rows, event, pinned count)
        self.flying = []         # gathers started: (work, out, header_rows, counts)
        self.results = []        # per batch: (out [world, rows, 6], header_rows, counts)
        self.comm = None
        self.pins, self.pin_next = [], 0

    def submit(self, wire: torch.Tensor, images_cap: int, header_rows: int):
        ev = pin = None
        if wire.is_cuda:
            if self.comm is None:
                self.comm = torch.cuda.Stream(wire.device)
            if not self.pins:
                self.pins = [torch.empty(1, dtype=torch.int32).pin_memory() for _ in range(4)]
            pin = self.pins[self.pin_next % 4]       # at most two batches wait at any time
            self.pin_next += 1
            pin.copy_(wire.reshape(-1)[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(wire.device))
        item = (wire, header_rows, ev, pin)
        if self.waiting:
            self._start(self.waiting.pop(0))
        self.waiting.append(item)

    def _start(self, item):
        wire, header_rows, ev, pin = item
        if self.world == 1:
            n = int(wire.reshape(-1)[0]) if ev is None else None
            if ev is not None:
                ev.synchronize()
                n = int(pin[0])
            self.flying.append((None, wire[:header_rows + n].unsqueeze(0), header_rows, [n]))
            return
        import contextlib
        if ev is not None:
            ev.synchronize()                       # the batch whose rows these are finished long ago
            n = int(pin[0])
        else:
            n = int(wire.reshape(-1)[0])
        if wire.is_cuda and self.device_collectives:
            ctx = torch.cuda.stream(self.comm)
            self.comm.wait_event(ev)
        else:
            ctx = contextlib.nullcontext()
            if wire.is_cuda:
                wire = wire[:header_rows + n].cpu()
        with ctx:
            mine = torch.tensor([n], dtype=torch.int64, device=wire.device)
            counts = torch.empty(self.world, dtype=torch.int64, device=wire.device)
            dist.all_gather_into_tensor(counts, mine, group=self.group)
            counts = [int(c) for c in counts.tolist()]
            rows = header_rows + max(max(counts), 0)
            if wire.shape[0] >= rows:
                send = wire[:rows]                 # rows beyond n are padding
            else:                                  # another rank has more rows than this buffer holds
                send = torch.zeros((rows, 6), dtype=torch.int32, device=wire.device)
                send[:wire.shape[0]] = wire
            work = None
            if self.root is not None:
                send = send.contiguous()
                out = parts = None
                if self.rank == self.root:             # one buffer per sender, as dist.gather is most commonly given
                    out = parts = [torch.empty((rows, 6), dtype=torch.int32, device=wire.device) for _ in range(self.world)]
                try:
                    work = dist.gather(send, parts, dst=self.root, group=self.group, async_op=True)
                except (RuntimeError, NotImplementedError, ValueError):
                    # a backend without gather refuses on every rank alike: deliver to all ranks from here on
                    self.root = None
            if self.root is None:
                out = torch.empty((self.world, rows, 6), dtype=torch.int32, device=wire.device)
                work = dist.all_gather_into_tensor(out.view(self.world * rows, 6), send, group=self.group, async_op=True)
            if wire.is_cuda:
                wire.record_stream(self.comm)
        self.flying.append((work, out, header_rows, counts, send))
        while len(self.flying) > 2:                # retire old gathers (they finished long ago)
            self._retire(self.flying.pop(0))

    def _retire(self, f):
        work, out, header_rows, counts = f[:4]
        if work is not None:
            work.wait()
        if self.keep_results:
            self.results.append((out, header_rows, counts))

    def finish(self):
        """Starts what is left, waits for every gather; returns the gathered messages per batch."""
        while self.waiting:
            self._start(self.waiting.pop(0))
        for f in self.flying:
            self._retire(f)
        self.flying = []
        if self.comm is not None:
            torch.cuda.current_stream(self.comm.device).wait_stream(self.comm)
        res, self.results = self.results, []
        return res


def unpack_gathered(batch) -> Tuple[torch.Tensor, torch.Tensor]:
    """(out [world, rows, 6] or a list of world [rows, 6] messages, header_rows, counts) of one batch -> rows of
    all ranks in rank order."""
    out, header_rows, counts = batch
    if out is None:
        raise ValueError('this rank did not receive the batch (RowExchange(root=...) delivers to the root only)')
    parts = [unpack_wire(out[r], header_rows) for r in range(len(out))]
    return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
