"""world_size-2 gloo tests (CPU) of the N>1 path: batch/seed sharding, chunked flat-gradient all-reduce with the
1/world average folded in as pre_scale, and the bench's max-over-ranks timing reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uwudiff_amd.gradsync import FlatGradSync, shard_range

    torch.manual_seed(1215 + rank)  # test_train.py:68-69: seed + global_rank
    n = 100_003
    g = torch.randn(n)
    mine = g.clone()
    sync = FlatGradSync(world, chunk_elems=30_000)
    chunks = sync.all_reduce(g)
    assert [c[1] for c in chunks] == [30_000, 30_000, 30_000, 10_003]
    sync.wait_all()
    # reference semantics: mean over ranks (DDP) == sum * pre_scale
    gathered = [torch.empty(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ref = torch.stack(gathered).sum(0)
    ok = torch.allclose(g, ref, rtol=0, atol=1e-6) and abs(sync.pre_scale - 1.0 / world) < 1e-12
    # slices reduced early (what DiT.set_grad_ready_hook reports from inside backward) are not reduced twice, and the
    # returned chunks cover the buffer exactly once
    g2 = mine.clone()
    sync2 = FlatGradSync(world, chunk_elems=30_000)
    sync2._on_ready(g2, [(70_000, 20_003, None), (40_000, 30_000, None)])
    chunks2 = sync2.all_reduce(g2)
    cover = sorted(chunks2)
    ok = ok and torch.allclose(g2, ref, rtol=0, atol=1e-6)
    ok = ok and cover[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(cover, cover[1:])) and sum(c[1] for c in cover) == n
    ok = ok and chunks2[:2] == [(40_000, 30_000), (70_000, 20_003)] and sync2._early == []
    # single=True: north_star's exchange as stated -- one call over the whole buffer, no early slices
    g3 = mine.clone()
    sync3 = FlatGradSync(world, chunk_elems=30_000, single=True)

    class _M:  # a model that would report early slices: single mode must not attach to it
        hook = "unset"

        def set_grad_ready_hook(self, h):
            self.hook = h

    m = _M()
    sync3.attach(m)
    chunks3 = sync3.all_reduce(g3)
    ok = ok and m.hook == "unset" and chunks3 == [(0, n)] and torch.allclose(g3, ref, rtol=0, atol=1e-6)
    lo, hi = shard_range(50, rank, world)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, (lo, hi), float(t), float(mine[0])))
    dist.destroy_process_group()


def test_flat_gradsync_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert [r[2] for r in res] == [(0, 25), (25, 50)]
    assert all(abs(r[3] - 0.2) < 1e-12 for r in res)  # max over ranks
    assert res[0][4] != res[1][4]  # per-rank seeds differ


def test_shard_range_covers_everything():
    from uwudiff_amd.gradsync import shard_range

    for n, w in [(50, 8), (7, 3), (16, 16), (3, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_world1_is_noop():
    from uwudiff_amd.gradsync import FlatGradSync

    s = FlatGradSync(1)
    g = torch.ones(10)
    assert s.all_reduce(g) == [(0, 10)] and s.pre_scale == 1.0 and torch.equal(g, torch.ones(10))
