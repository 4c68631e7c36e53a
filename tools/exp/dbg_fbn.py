import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import openmm_amd
from openmm_amd import lib as L
from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
from tests.layer_fixtures import keypair, noise, rns_uniform, ternary, SEED
from tests.oracle_binding import OracleCtx, uniform_mod, A, load
from tests.oracle_layers import ConvOracle
from tests.test_fused_bn import _bind, make_case
olib = _bind(load())
bl, k, N = 37, 12, 4096
T = 1 << bl
c, h, w, fh, fw, M = 128, 28, 28, 3, 3, 8
f, s = make_case(bl, fh, fw, c, M, SEED + 77)
scaled = np.zeros_like(f)
olib.mmo_fused_bn_filters(A(f), A(s), fh, fw, c, M, bl, k, T, A(scaled))
filters = np.ascontiguousarray(scaled.reshape(fh, fw, c, M).transpose(3, 2, 0, 1))
sv = np.where(filters >= T // 2, filters.astype(np.int64) - T, filters.astype(np.int64))
print("weights range", sv.min(), sv.max(), "zeros", (sv == 0).mean())
o, g = OracleCtx(N, (60, 49), T), openmm_amd.Context(N, (60, 49), T)
cv, meta = ConvOracle(o, (c, h, w), (c, fh, fw), M, 1, 1, True), Meta((c, h, w), (c, fh, fw), M, 1, 1, True)
sk, pk = keypair(o)
share0, share1 = uniform_mod(SEED + 1, (c, h, w), T), uniform_mod(SEED + 2, (c, h, w), T)
cts = cv.encrypt_image(sk, share0, rns_uniform(o, SEED + 4, cv.n_in, 2), noise(SEED + 5, (cv.n_in, N)))
u, e1 = ternary(SEED + 6, (cv.n_out, N)), noise(SEED + 7, (cv.n_out, N))
mask = rns_uniform(o, SEED + 8, cv.n_out, 1)[:, 0]
fp = cv.encode_filters(filters)
print("zero filter polys:", [(m, cc) for m in range(M) for cc in range(cv.n_cs) if not fp[m, cc].any()])
ref_out, ref_share = cv.server(cts, cv.encode_image(share1), fp, np.ascontiguousarray(pk[:, :1]), u, e1, mask)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64) if x.dtype == np.uint64 else x).cuda()
for route in ("model", "0", "1"):
    if route != "model": os.environ["MMH_CONV_POINTWISE"] = route
    srv = HomConv2DSS(); srv.setUp(g, pk_ntt=dev(pk))
    d_f = dev(filters)
    o1 = srv.conv2DSSFromTensors(dev(cts), dev(share1), d_f, meta, (dev(u), dev(e1), dev(mask)))
    prep = srv.prepareFilters(d_f, meta)
    o2 = srv.conv2DSSPrepared(dev(cts), srv.encodeImage(dev(share1), meta), prep, meta, (dev(u), dev(e1), dev(mask)))
    for name, oo in (("tensors", o1), ("prepared", o2)):
        a = oo[0].cpu().numpy().view(np.uint64)
        bad = np.argwhere(a != ref_out)
        print(route, name, "cts equal", len(bad) == 0, "share equal", np.array_equal(oo[1].cpu().numpy().view(np.uint64), ref_share), bad[:3].tolist(), len(bad))
lib = L.load_library()
d_scaled = torch.empty((M, c, fh, fw), dtype=torch.int64, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
L.check(lib.mmh_fused_bn_filters(L.dptr(dev(f)), L.dptr(dev(s)), fh, fw, c, M, bl, k, T, 1, L.dptr(d_scaled), L.dptr(flag), L._stream_ptr()))
torch.cuda.synchronize()
got = d_scaled.cpu().numpy().view(np.uint64)
bad = np.argwhere(got != filters)
print("scaled equal", len(bad) == 0, len(bad), bad[:5].tolist())
for b in bad[:5]:
    b = tuple(b); print(got[b], filters[b])
df, ds = dev(f), dev(s)
L.check(lib.mmh_fused_bn_filters(L.dptr(df), L.dptr(ds), fh, fw, c, M, bl, k, T, 1, L.dptr(d_scaled), L.dptr(flag), L._stream_ptr()))
torch.cuda.synchronize()
print("kept alive equal", np.array_equal(d_scaled.cpu().numpy().view(np.uint64), filters))
