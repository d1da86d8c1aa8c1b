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
    """All-gather of every batch's detections, behind the computation and without host stalls.

    submit(wire, images_cap, header_rows) hands over this rank's message for the batch just launched (its row
    count is still being computed on the device).  Each call issues, on a side stream, in the same order on
    every rank:
      1. the gather of the messages of the batch submitted two calls ago, sized by the row counts that were
         exchanged one call ago and have long since arrived in pinned host memory;
      2. the exchange of the row counts of the batch just submitted -- its input is the count word of the
         message itself, still on the device, so the host does not need to know it; the result is copied to
         pinned memory behind it.
    The host therefore never waits for a collective it has just issued, only (and normally not at all) for one
    issued a whole batch earlier; the transfers overlap the kernels of the following batches.  finish() drains
    the pipeline and waits for every gather.  Works on gloo / CPU tensors too (no streams there; messages on a
    GPU are staged through the host for a backend that cannot move device memory).

    root=None: every rank receives every message (all-gather: the path's one collective).  root=r: only the rank
    whose GLOBAL rank is r receives them (gather) -- enough for a single evaluator process (src/test.py:98-99 builds
    one list in one process); r must belong to `group`.  The mode is fixed at construction; a backend that lacks the
    collective raises.

    payload='rows' (default): the detections travel.  payload='header': only the header of every message does (row
    count and row offsets per image, image ids and sizes: 24 bytes per image instead of 24 bytes per detection) --
    for a consumer that tallies detections per image and needs no boxes.

    submit() takes the message over: its buffer must not be written afterwards (the rows behind the message's own are
    zeroed as padding, on the side stream, when another rank's message is longer).

    With timing=True (device collectives only) every message gather is bracketed by events on the side stream;
    `gather_ms` and `gather_bytes_in` (bytes this rank received) accumulate over the retired batches."""

    def __init__(self, group=None, keep_results: bool = True, root=None, payload: str = 'rows', timing: bool = False):
        if payload not in ('rows', 'header'):
            raise ValueError("payload must be 'rows' or 'header'")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.root = root
        self.rank = dist.get_rank() if self.world > 1 else 0                 # global rank: what `root` and gather's dst name
        self.grank = dist.get_rank(group) if self.world > 1 else 0           # rank inside the group: the order of gathered data
        if root is not None and self.world > 1 and root not in dist.get_process_group_ranks(group if group is not None else dist.group.WORLD):
            raise ValueError(f'root {root} is not a member of the group')
        # RCCL moves device buffers; any other backend (gloo rehearsals) gets the message staged on the host
        self.device_collectives = self.world > 1 and dist.get_backend(group) == 'nccl'
        self.keep_results = keep_results
        self.payload = payload
        self.timing = timing
        self.counted = []        # counts exchange issued: (wire, header_rows, counts_host, counts_event, counts_work)
        self.flying = []         # message gathers issued: (work, out, header_rows, counts, send, events)
        self.results = []        # per batch: (out [world, rows, 6] or list of world [rows, 6], header_rows, counts)
        self.comm = None
        self.host_waits = 0      # times the host found a counts exchange of an earlier batch still unfinished
        self.gather_ms, self.gather_bytes_in, self.gathers = 0.0, 0, 0

    # -- stage 2: the row counts of the batch just submitted
    def _issue_counts(self, wire: torch.Tensor, header_rows: int):
        word = wire.reshape(-1)[:1]
        if self.world == 1:
            if wire.is_cuda:
                host = torch.empty(1, dtype=torch.int32).pin_memory()
                host.copy_(word, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(wire.device))
                return (wire, header_rows, host, ev, None)
            return (wire, header_rows, word.clone(), None, None)
        if wire.is_cuda and self.device_collectives:
            if self.comm is None:
                self.comm = torch.cuda.Stream(wire.device)
            self.comm.wait_stream(torch.cuda.current_stream(wire.device))      # the message is being written there
            with torch.cuda.stream(self.comm):
                counts = torch.empty(self.world, dtype=torch.int32, device=wire.device)
                dist.all_gather_into_tensor(counts, word, group=self.group)    # enqueued on the side stream
                host = torch.empty(self.world, dtype=torch.int32).pin_memory()
                host.copy_(counts, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.comm)
                wire.record_stream(self.comm)
            return (wire, header_rows, host, ev, None)
        # host-staged backends: the count has to come down first (a rehearsal path, not the product path)
        n = torch.tensor([int(word.item())], dtype=torch.int32)
        host = torch.empty(self.world, dtype=torch.int32)
        work = dist.all_gather_into_tensor(host, n, group=self.group, async_op=True)
        return (wire, header_rows, host, None, work)

    # -- stage 1: the messages of a batch whose counts have arrived
    def _issue_messages(self, item):
        wire, header_rows, host, ev, cwork = item
        if cwork is not None:
            cwork.wait()
        if ev is not None:
            if not ev.query():
                self.host_waits += 1
            ev.synchronize()
        counts = [int(c) for c in host.tolist()]
        if self.world == 1:
            self.flying.append((None, wire[:header_rows + counts[0]].unsqueeze(0), header_rows, counts, None, None, 0))
            return
        import contextlib
        n = counts[self.grank]                     # the gathered counts are in group-rank order
        if self.payload == 'header':
            n, counts = 0, [0] * self.world        # nothing behind the header travels
        rows = header_rows + max(max(counts), 0)
        on_device = wire.is_cuda and self.device_collectives
        ctx = torch.cuda.stream(self.comm) if on_device else contextlib.nullcontext()
        events = None
        with ctx:
            if wire.is_cuda and not on_device:
                wire = wire[:header_rows + n].cpu()
            if wire.shape[0] >= rows:
                send = wire[:rows]
                send[header_rows + n:].zero_()        # padding up to the longest message
            else:                                  # another rank has more rows than this buffer holds
                send = torch.zeros((rows, 6), dtype=torch.int32, device=wire.device)
                send[:header_rows + n] = wire[:header_rows + n]
            send = send.contiguous()
            if on_device and self.timing:
                events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events[0].record(self.comm)
            if self.root is not None:
                out = None
                if self.rank == self.root:
                    out = [torch.empty((rows, 6), dtype=torch.int32, device=send.device) for _ in range(self.world)]
                work = dist.gather(send, out, dst=self.root, group=self.group, async_op=True)
            else:
                out = torch.empty((self.world, rows, 6), dtype=torch.int32, device=send.device)
                work = dist.all_gather_into_tensor(out.view(self.world * rows, 6), send, group=self.group, async_op=True)
            if on_device:
                if events is not None:
                    work.wait()                    # orders the side stream behind the collective; the host does not block
                    events[1].record(self.comm)
                    work = None
                wire.record_stream(self.comm)
        received = 0 if out is None else (self.world - 1) * rows * 24
        self.flying.append((work, out, header_rows, counts, send, events, received))
        while len(self.flying) > 2:                # retire old gathers (they finished long ago)
            self._retire(self.flying.pop(0))

    def submit(self, wire: torch.Tensor, images_cap: int, header_rows: int):
        if len(self.counted) >= 2:
            self._issue_messages(self.counted.pop(0))
        self.counted.append(self._issue_counts(wire, header_rows))

    def _retire(self, f):
        work, out, header_rows, counts = f[:4]
        if work is not None:
            work.wait()
        if len(f) > 5:
            self.gathers += 1
            self.gather_bytes_in += f[6]
            if f[5] is not None:
                f[5][1].synchronize()
                self.gather_ms += f[5][0].elapsed_time(f[5][1])
        if self.keep_results:
            self.results.append((out, header_rows, counts))

    def finish(self):
        """Drains the pipeline, waits for every gather; returns the gathered messages per batch."""
        while self.counted:
            self._issue_messages(self.counted.pop(0))
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
