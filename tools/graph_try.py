import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, openmm_amd
from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
t, N = 1 << 37, 4096
ctx = openmm_amd.Context(N, (60, 49), t, device=0)
gen = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)
qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda")
srv = HomConv2DSS(); srv.setUp(ctx, pk_ntt=rnd(0, 1 << 62, 2, 2, N) % qs.view(1, 2, 1))
for shape in [(16,56,56,1,1,64,0,1), (64,56,56,3,3,64,1,1), (256,14,14,1,1,1024,0,1)]:
    c,h,w,fh,fw,m,same,stride = shape
    meta = Meta((c,h,w),(c,fh,fw),m,same,stride,True)
    plan = srv.plan(meta)
    cts = rnd(0, 1 << 62, plan.n_in, 2, 2, N) % qs.view(1,1,2,1)
    share = rnd(0, t, c, h, w)
    filters = (rnd(-(1<<12)+1, 1<<12, m, c, fh, fw) | 1) & (t-1)
    rand = (rnd(-1,2,plan.n_out,N), rnd(-8,9,plan.n_out,N), rnd(0,1<<62,plan.n_out,N) % ctx.q[0])
    prep = srv.prepareFilters(filters, meta)
    layer = lambda: srv.conv2DSSPrepared(cts, srv.encodeImage(share, meta), prep, meta, rand)
    ref = layer(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): layer()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = layer()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    def timeit(fn, reps=50):
        fn(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
    print(shape, "graph ok" if ok else "MISMATCH", "eager %.3f ms, graph %.3f ms" % (timeit(layer), timeit(g.replay)))
