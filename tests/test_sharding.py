"""CPU, gloo, world_size 2 and 3: the sharding layer partitions units and reassembles them
exactly as the unsharded call, ragged and empty shards included."""
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


class FakeServer:
    """Stands in for HomConv2DSS on CPU: deterministic functions of the filter slice, so that
    any mix-up of ranges or ordering changes the result."""

    class P:
        n_spatial = 3
        out_shape = (0, 4, 5)

    def plan(self, meta):
        return self.P

    def encodeFilters(self, filters, meta):
        assert filters.shape[0] == meta.n_filters
        return filters

    def conv2DSSFromTensors(self, s0, share, f, meta, rand, out=None):
        assert f.shape[0] == meta.n_filters
        m = f.shape[0]
        key = f.reshape(m, -1).sum(1)
        cts = (key[:, None, None] * 7 + torch.arange(self.P.n_spatial)[None, :, None] + rand[0]).expand(m, 3, 8)
        cts = cts.reshape(m * 3, 2, 1, 4).contiguous()
        sh = (key[:, None, None] + torch.arange(20).reshape(1, 4, 5)).contiguous()
        if out is not None:
            assert tuple(out[0].shape) == tuple(cts.shape) and tuple(out[1].shape) == tuple(sh.shape)
            out[0].copy_(cts), out[1].copy_(sh)
            return out
        return cts, sh


def _worker(rank, world, port, n_filters, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openmm_amd.hom_conv2d_ss import Meta
    from openmm_amd.sharding import ShardedConv, shard_range, sharded_elementwise
    try:
        filters = torch.arange(n_filters * 2 * 3 * 3, dtype=torch.int64).reshape(n_filters, 2, 3, 3)
        meta = Meta((2, 8, 8), (2, 3, 3), n_filters, 1, 1, True)
        srv = FakeServer()
        s0 = torch.zeros((2, 2, 2, 4), dtype=torch.int64)
        ref_cts, ref_share = srv.conv2DSSFromTensors(s0, s0, filters, meta, (5,))
        ok = []
        for mode in ("gather", "d2h"):
            sc = ShardedConv(srv, meta, mode=mode)
            lo, hi = shard_range(n_filters, world, rank)
            assert (sc.lo, sc.hi) == (lo, hi)
            for rep in range(2):          # the assembly buffers are made once and written again by every call
                cts, share = sc.run(s0, s0, filters, lambda n: (5,))
                if rank == 0:
                    ok.append(torch.equal(cts, ref_cts) and torch.equal(share, ref_share))
                else:
                    assert cts is None and share is None
        x = torch.arange(1000, dtype=torch.int64)
        y = sharded_elementwise(lambda t: t * 3 + 1, x)
        if rank == 0:
            q.put((all(ok), len(ok) == 4, torch.equal(y, x * 3 + 1)))
        else:
            assert y is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_filters", [(2, 6), (2, 5), (3, 2), (3, 7)])
def test_sharded_conv_gloo(world, n_filters):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_filters, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, True)


class FailingServer(FakeServer):
    """raises the reference's zero-filter error (conv2DOneFilter :573-580) for a slice that holds an all-zero filter"""

    def conv2DSSFromTensors(self, s0, share, f, meta, rand, out=None):
        if bool((f.reshape(f.shape[0], -1) == 0).all(1).any()):
            raise ValueError("conv2DSS: a filter is all zero")
        return super().conv2DSSFromTensors(s0, share, f, meta, rand, out=out)


def _worker_zero_filter(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openmm_amd.hom_conv2d_ss import Meta
    from openmm_amd.sharding import ShardedConv
    try:
        n_filters = 6
        filters = torch.arange(n_filters * 2 * 3 * 3, dtype=torch.int64).reshape(n_filters, 2, 3, 3) + 1
        filters[n_filters - 1] = 0           # lives in the LAST rank's slice only
        meta = Meta((2, 8, 8), (2, 3, 3), n_filters, 1, 1, True)
        s0 = torch.zeros((2, 2, 2, 4), dtype=torch.int64)
        got = []
        for mode in ("gather", "d2h"):
            sc = ShardedConv(FailingServer(), meta, mode=mode)
            try:
                sc.run(s0, s0, filters, lambda n: (5,))
                got.append("no error")
            except ValueError:
                got.append("own")
            except RuntimeError:
                got.append("peer")
            filters_ok = filters.clone()
            filters_ok[n_filters - 1] = 1
            cts, share = sc.run(s0, s0, filters_ok, lambda n: (5,))   # and the object still works afterwards
            got.append(cts is not None if rank == 0 else cts is None)
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_error_on_one_rank_is_raised_on_every_rank():
    """a rank whose filter slice fails must not leave the others inside the collective"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_zero_filter, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res[2] == ["own", True, "own", True]
    assert res[0] == ["peer", True, "peer", True] and res[1] == ["peer", True, "peer", True]


def test_shard_range_matches_launchworks():
    from openmm_amd.sharding import shard_range
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            per = (n + w - 1) // w
            assert all(hi - lo <= per for lo, hi in parts)
