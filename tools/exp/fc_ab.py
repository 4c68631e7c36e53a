"""A/B of the FC layer with two builds of libmm_hip: python tools/exp/fc_ab.py <lib path>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd.lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import openmm_amd
from openmm_amd.hom_fc_ss import HomFCSS, Meta as FcMeta
t, N = 1 << 37, 4096
ctx = openmm_amd.Context(N, (60, 49), t)
gen = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda lo, hi, *s: torch.randint(lo, hi, s, dtype=torch.int64, device="cuda", generator=gen)
qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda")
pk = rnd(0, 1 << 62, 2, 2, N) % qs.view(1, 2, 1)
fc = HomFCSS(); fc.setUp(ctx, pk_ntt=pk)
fmeta = FcMeta((1001, 2048), (2048,), True)
fplan = fc.plan(fmeta)
vec_ct = rnd(0, 1 << 62, fplan.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
vec, wmat = rnd(0, t, 2048), (rnd(-(1 << 12) + 1, 1 << 12, 1001, 2048) | 1) & (t - 1)
frand = (rnd(-1, 2, fplan.n_out, N), rnd(-8, 9, fplan.n_out, N), rnd(0, 1 << 62, fplan.n_out, N) % ctx.q[0])
wp, ip = fc.encodeWeightMatrix(wmat, fmeta), fc.encodeInputVector(vec, fmeta)
def run_all(): return fc.matVecMul(fc.encodeWeightMatrix(wmat, fmeta), vec_ct, fc.encodeInputVector(vec, fmeta), fmeta, frand)
def run_mv(): return fc.matVecMul(wp, vec_ct, ip, fmeta, frand)
for name, fn in (("encode+matVecMul", run_all), ("matVecMul", run_mv)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print(L.LIB_PATH.split("/")[-1], name, f"{(time.perf_counter() - t0) / 50 * 1e3:.4f} ms")
