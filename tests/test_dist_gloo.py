"""CPU, world_size 2 over gloo: image sharding and the all-gather of result rows reproduce the
single-process result order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _fake_rows(image_index: int):
    """Deterministic stand-in for one image's detections (row count varies, may be zero)."""
    g = torch.Generator().manual_seed(100 + image_index)
    n = int(torch.randint(0, 7, (1,), generator=g))
    ids = torch.stack([torch.full((n,), 1000 + image_index, dtype=torch.int64),
                       torch.randint(1, 21, (n,), generator=g)], dim=1)
    vals = torch.rand(n, 5, generator=g, dtype=torch.float64)
    return ids, vals


def _worker(rank, world, port, num_images, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from object_detection_amd.dist import gather_detections, shard_range
    lo, hi = shard_range(num_images, rank, world)
    rows = [_fake_rows(i) for i in range(lo, hi)]
    ids = torch.cat([r[0] for r in rows]) if rows else torch.zeros(0, 2, dtype=torch.int64)
    vals = torch.cat([r[1] for r in rows]) if rows else torch.zeros(0, 5, dtype=torch.float64)
    gids, gvals = gather_detections(ids, vals)
    q.put((rank, gids.numpy(), gvals.numpy()))      # arrays, not tensors: a tensor's shared-memory handle dies with this process
    dist.barrier()
    dist.destroy_process_group()


def _run(num_images, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = [_fake_rows(i) for i in range(num_images)]
    want_ids = torch.cat([r[0] for r in rows]) if rows else torch.zeros(0, 2, dtype=torch.int64)
    want_vals = torch.cat([r[1] for r in rows]) if rows else torch.zeros(0, 5, dtype=torch.float64)
    for _, gids, gvals in got:
        assert torch.equal(torch.from_numpy(gids), want_ids) and torch.equal(torch.from_numpy(gvals), want_vals)


def test_shard_ranges_cover_batch():
    from object_detection_amd.dist import shard_range
    for n in (0, 1, 7, 16, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_gather_two_ranks_even_and_ragged():
    _run(8)
    _run(5)       # uneven shards (3 + 2 images)
    _run(1)       # one rank has no image at all


# ---- pipelined exchange of compact messages ------------------------------------------------------

def _fake_dets(image_index: int):
    g = torch.Generator().manual_seed(500 + image_index)
    n = int(torch.randint(0, 9, (1,), generator=g))
    return (torch.randint(1, 81, (n,), generator=g), torch.rand(n, 4, generator=g), torch.rand(n, generator=g))


def _images_of(k, rank, world, per_rank):
    """Images of `rank` in batch k: rank 1 is one image short in batch 1; in a world of four, rank 2 has none in batch 0."""
    n = per_rank - (1 if (rank == 1 and k == 1) else 0)
    if world >= 4 and rank == 2 and k == 0:
        n = 0
    return [k * world * per_rank + rank * per_rank + i for i in range(n)]


def _exchange_worker(rank, world, port, batches, per_rank, q, root=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import wire_from_rows
    from object_detection_amd.dist import RowExchange, unpack_gathered
    hdr = (2 + (per_rank + 1) + 4 * per_rank + 5) // 6
    ex = RowExchange(root=root)
    for k in range(batches):
        imgs = _images_of(k, rank, world, per_rank)
        dets = [_fake_dets(i) for i in imgs]
        wire = wire_from_rows(dets, [(1 << 33) + i for i in imgs], [(100 + i, 200 + i) for i in imgs], hdr)
        ex.submit(wire, per_rank, hdr)
    got = [unpack_gathered(bt) for bt in ex.finish() if bt[0] is not None]
    q.put((rank, [(a.numpy(), b.numpy()) for a, b in got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,root', [(2, None), (2, 1), (4, None), (4, 3)])
def test_row_exchange_pipelined(world, root):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import rows_from_detections
    batches, per_rank = 5, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, batches, per_rank, q, root)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank_, per_batch in got:
        if root is not None and rank_ != root:          # gather to the root: the other ranks only send
            assert per_batch == []
            continue
        assert len(per_batch) == batches
        for k, (gids, gvals) in enumerate(per_batch):
            imgs = []
            for rank in range(world):
                imgs += _images_of(k, rank, world, per_rank)
            want_ids, want_vals = rows_from_detections([_fake_dets(i) for i in imgs], [(1 << 33) + i for i in imgs],
                                                       [(100 + i, 200 + i) for i in imgs])
            assert torch.equal(torch.from_numpy(gids), want_ids) and torch.equal(torch.from_numpy(gvals), want_vals)


def _subgroup_worker(rank, world, port, members, batches, per_rank, q, root, payload):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    group = dist.new_group(ranks=members)            # every rank of the world takes part in creating it
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import wire_from_rows
    from object_detection_amd.dist import RowExchange, unpack_gathered
    out = []
    if rank in members:
        hdr = (2 + (per_rank + 1) + 4 * per_rank + 5) // 6
        ex = RowExchange(group=group, root=root, payload=payload)
        assert ex.world == len(members) and ex.rank == rank and ex.grank == members.index(rank)
        for k in range(batches):
            imgs = _images_of(k, rank, world, per_rank)
            dets = [_fake_dets(i) for i in imgs]
            wire = wire_from_rows(dets, [(1 << 33) + i for i in imgs], [(100 + i, 200 + i) for i in imgs], hdr)
            ex.submit(wire, per_rank, hdr)
        for bt in ex.finish():
            if bt[0] is None:
                continue
            if payload == 'header':                  # only the headers travelled: row counts per image of every member
                out.append([[int(x) for x in bt[0][r].reshape(-1)[2:2 + int(bt[0][r].reshape(-1)[1]) + 1]] for r in range(len(bt[0]))])
            else:
                a, b = unpack_gathered(bt)
                out.append((a.numpy(), b.numpy()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('root,payload', [(None, 'rows'), (3, 'rows'), (None, 'header')])
def test_row_exchange_in_a_sub_group(root, payload):
    """A group that is not the whole world (global ranks 1 and 3 of four): the gathered counts come in GROUP-rank
    order, `root` names a GLOBAL rank.  Rank 1 is one image short in batch 1, so the two messages differ in length."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import rows_from_detections
    world, members, batches, per_rank = 4, [1, 3], 4, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, members, batches, per_rank, q, root, payload))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == [] and got[2] == []
    for rank_ in members:
        per_batch = got[rank_]
        if root is not None and rank_ != root:
            assert per_batch == []
            continue
        assert len(per_batch) == batches
        for k, item in enumerate(per_batch):
            per_member = [[_fake_dets(i) for i in _images_of(k, r, world, per_rank)] for r in members]
            if payload == 'header':
                for offs, dets in zip(item, per_member):
                    want = [0]
                    for d in dets:
                        want.append(want[-1] + int(d[0].numel()))
                    assert offs == want
                continue
            imgs = [i for r in members for i in _images_of(k, r, world, per_rank)]
            want_ids, want_vals = rows_from_detections([_fake_dets(i) for i in imgs], [(1 << 33) + i for i in imgs],
                                                       [(100 + i, 200 + i) for i in imgs])
            assert torch.equal(torch.from_numpy(item[0]), want_ids) and torch.equal(torch.from_numpy(item[1]), want_vals)


def test_row_exchange_rejects_an_unknown_payload():
    from object_detection_amd.dist import RowExchange
    with pytest.raises(ValueError):
        RowExchange(payload='boxes')
