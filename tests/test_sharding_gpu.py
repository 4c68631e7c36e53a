"""The sharded forms with the REAL kernels under world_size 2: two processes, each with its own HIP context on
the box's GPU (rank r uses device r % device_count), process group over gloo (RCCL refuses two ranks on one device;
on a multi-GPU node the same code runs over nccl, see bench.py).  Rank 0's assembled outputs must equal the
unsharded call bit for bit, in both assembly modes (collective gather / per-rank copies into the shared page-locked
host buffer), for ragged filter counts and for more ranks than filters; BOLE sharded by ciphertext index likewise."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu
T, N, P32 = 1 << 37, 4096, 4293918721


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import openmm_amd
        from openmm_amd.hom_bn_ss import HomBNSS, Meta as BnMeta
        from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
        from openmm_amd.sharding import ShardedConv, sharded_bole
        g = openmm_amd.Context(N, (60, 49), T, device=dev)
        smp = openmm_amd.Sampler(g, key=bytes(range(32)))     # same key on every rank: replicated inputs
        _, pk = smp.keygen()
        srv = HomConv2DSS()
        srv.setUp(g, pk_ntt=pk)
        results = []
        for (ishape, fhw, M, same, stride) in [((16, 14, 14), (3, 3), 7, 1, 1), ((12, 28, 28), (1, 1), 6, 0, 1),
                                                ((3, 70, 70), (3, 3), 1, 1, 1), ((64, 14, 14), (1, 1), 130, 0, 2)]:
            meta = Meta(ishape, (ishape[0],) + fhw, M, same, stride, True)
            plan = srv.plan(meta)
            gen = torch.Generator(device="cuda").manual_seed(0x4D264D)
            cts = smp.uniform(2 * plan.n_in).view(plan.n_in, 2, 2, N)
            share = torch.randint(0, T, ishape, dtype=torch.int64, device="cuda", generator=gen)
            filters = torch.randint(1, 1 << 12, (M,) + meta.fshape, dtype=torch.int64, device="cuda", generator=gen)
            rand = (smp.ternary(plan.n_out), smp.noise(plan.n_out), smp.uniform(plan.n_out, lv=1)[:, 0].contiguous())
            ref = srv.conv2DSSFromTensors(cts, share, filters, meta, rand)
            for mode in ("gather", "d2h"):
                sc = ShardedConv(srv, meta, mode=mode)
                lo = sc.lo * plan.n_spatial
                for rep in range(2):
                    got = sc.run(cts, share, filters, lambda n: tuple(r[lo:lo + n] for r in rand))
                    if rank == 0:
                        results.append(bool(torch.equal(got[0].to(ref[0].device), ref[0])
                                            and torch.equal(got[1].to(ref[1].device), ref[1])))
                    else:
                        assert got == (None, None)
        # BOLE by ciphertext index, ragged last ciphertext
        gf = openmm_amd.Context(N, (60, 49), P32, device=dev)
        bn = HomBNSS()
        bn.setUp_Field(P32, gf)
        n = 3 * N + 100
        meta = BnMeta((n,), P32, True)
        fs = openmm_amd.Sampler(gf, key=bytes(range(32)))
        ex, ey = (fs.uniform(2 * 4).view(4, 2, 2, N) for _ in range(2))
        gen = torch.Generator(device="cuda").manual_seed(7)
        x1, y1, rr = (torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device="cuda", generator=gen) for _ in range(3))
        px, py = bn.encodeVector(x1, meta), bn.encodeVector(y1, meta)
        ref = bn.bole(ex, ey, px, py, meta, rr)
        got = sharded_bole(bn, ex, ey, px, py, meta, rr)
        if rank == 0:
            results.append(bool(torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])))
            q.put(results)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_layers_real_kernels_two_ranks(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = q.get(timeout=5)
    assert len(res) == 4 * 2 * 2 + 1 and all(res), res
