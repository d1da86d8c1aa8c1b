"""Independent batches in flight on alternating HIP streams.

The path of one batch is a chain of some twenty-five kernels; about a third of its time goes to kernels
that wait on latency (merge levels, per-segment planning and finalising, the small bookkeeping kernels)
and leave most of the chip idle.  Issuing consecutive batches on different streams lets those run under
the throughput-bound kernels (NMS, the streaming scan) of the neighbouring batch.  The library is
stateless apart from caller-owned workspaces, and `ops` keeps one workspace per (device, operator, stream),
so batches on different streams never share scratch memory.  The reference runs one batch at a time
(src/test.py:90-99); this module has no counterpart there.
"""
from __future__ import annotations

import torch


class StreamRing:
    """`with ring.next(): ...` runs the block on the next of `n` streams; `synchronize()` waits for all."""

    def __init__(self, device, n: int = 2):
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(max(int(n), 1))]
        self._i = 0

    def __len__(self):
        return len(self.streams)

    def next(self):
        st = self.streams[self._i % len(self.streams)]
        self._i += 1
        st.wait_stream(torch.cuda.current_stream(self.device))      # the block's inputs were produced on the current stream
        return torch.cuda.stream(st)

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def join(self):
        """Makes the current stream wait for everything issued on the ring (no host wait)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)
