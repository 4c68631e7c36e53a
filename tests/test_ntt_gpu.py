"""GPU parity: batched NTT/INTT through the C ABI vs the oracle, bit for bit."""
import numpy as np
import pytest

from tests.oracle_binding import OracleCtx, uniform_mod, splitmix64

pytestmark = pytest.mark.gpu
SEED = 0x4D264D
CONFIGS = [(4096, (60, 49)), (8192, (60, 60, 60, 49))]


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def _polys(o, n_polys, seed):
    return np.stack([np.stack([uniform_mod(seed + 7 * k + l, o.N, q) for l, q in enumerate(o.q)])
                     for k in range(n_polys)])


@pytest.mark.parametrize("N,bits", CONFIGS)
def test_parameters_match_oracle(mm, N, bits):
    o, g = OracleCtx(N, bits, 0), mm.Context(N, bits, 0)
    assert g.q == o.q and g.psi == o.psi


@pytest.mark.parametrize("N,bits", CONFIGS)
@pytest.mark.parametrize("n_polys", [1, 3, 17])
def test_forward_inverse_bit_exact(mm, N, bits, n_polys):
    from openmm_amd.lib import to_device, to_host
    o, g = OracleCtx(N, bits, 0), mm.Context(N, bits, 0)
    a = _polys(o, n_polys, SEED)
    d = to_device(a)
    g.ntt_forward(d)
    fwd = to_host(d)
    assert np.array_equal(fwd, o.ntt_forward(a))
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)


@pytest.mark.parametrize("threads512", ["0", "1"])
@pytest.mark.parametrize("f64", ["0", "1"])
def test_n8192_both_kernel_families(mm, monkeypatch, threads512, f64):
    """N = 8192 has 256-thread kernels (32 coefficients per lane) and 512-thread kernels (16 per lane, the last stage
    folded into the store / load); the dispatcher picks per launch (ntt.hip).  Both, forced, forward and inverse, on the
    integer and on the double-precision butterflies, against the oracle: random, all-(q-1) and half-zero polynomials."""
    from openmm_amd.lib import to_device, to_host
    monkeypatch.setenv("MMH_NTT_512", threads512)     # read on every launch
    monkeypatch.setenv("MMH_NTT_F64", f64)            # read when the context is made
    N, bits = 8192, (60, 60, 60, 49)
    o, g = OracleCtx(N, bits, 0), mm.Context(N, bits, 0)
    a = _polys(o, 5, SEED + 512)
    a[1] = np.array(o.q, dtype=np.uint64)[:, None] - np.uint64(1)
    a[2, :, ::2] = 0
    d = to_device(a)
    g.ntt_forward(d)
    assert np.array_equal(to_host(d), o.ntt_forward(a))
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)
    # single-limb launches on the 49-bit prime (the case the dispatcher sends to the 512-thread forward kernel)
    g1 = mm.Context(N, q=[o.q[3]])
    b = np.ascontiguousarray(a[:, 3:4])
    d = to_device(b)
    g1.ntt_forward(d)
    assert np.array_equal(to_host(d), o.ntt_forward(a)[:, 3:4])
    g1.ntt_inverse(d)
    assert np.array_equal(to_host(d), b)


def test_edge_values_and_single_limb(mm):
    """all-zero, all q-1, a single monomial; limbs=1 uses q[0] only; empty batch is a no-op"""
    from openmm_amd.lib import to_device, to_host
    import torch
    o, g = OracleCtx(4096, (60, 49), 0), mm.Context(4096, (60, 49), 0)
    N = 4096
    a = np.zeros((3, 1, N), np.uint64)
    a[1] = o.q[0] - 1
    a[2, 0, 4095] = 12345
    d = to_device(a)
    g.ntt_forward(d, limbs=1)
    assert np.array_equal(to_host(d), o.ntt_forward(a, limbs=1))
    g.ntt_inverse(d, limbs=1)
    assert np.array_equal(to_host(d), a)
    g.ntt_forward(torch.empty(0, dtype=torch.int64, device="cuda"), limbs=2)


def test_lazy_inputs_accepted(mm):
    """forward accepts inputs in [0,4q), inverse in [0,2q) (lazy SEAL callers, hom_conv2d_ss.cc:306)"""
    from openmm_amd.lib import to_device, to_host
    o, g = OracleCtx(4096, (60, 49), 0), mm.Context(4096, (60, 49), 0)
    a = _polys(o, 2, SEED + 99)
    qs = np.array(o.q, dtype=np.uint64)[None, :, None]
    lazy = a + qs * (splitmix64(SEED, a.size).reshape(a.shape) % np.uint64(4))
    d = to_device(lazy)
    g.ntt_forward(d)
    ref = o.ntt_forward(a)
    assert np.array_equal(to_host(d), ref)
    lazy2 = ref + qs * (splitmix64(SEED + 1, a.size).reshape(a.shape) % np.uint64(2))
    d = to_device(lazy2)
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)


def test_full_size_roundtrip_and_linearity(mm):
    """BASELINE-size batch (4096 polys x 2 limbs = 256 MiB): size-independent properties"""
    import torch
    from openmm_amd.lib import to_host
    g = mm.Context(4096, (60, 49), 0)
    n = 4096
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    qs = torch.tensor(g.q, dtype=torch.int64, device="cuda").view(1, 2, 1)
    a = torch.randint(0, 2 ** 62, (n, 2, 4096), dtype=torch.int64, device="cuda", generator=gen) % qs
    b = torch.randint(0, 2 ** 62, (n, 2, 4096), dtype=torch.int64, device="cuda", generator=gen) % qs
    s = (a + b) % qs
    fa, fb, fs = a.clone(), b.clone(), s.clone()
    g.ntt_forward(fa), g.ntt_forward(fb), g.ntt_forward(fs)
    assert torch.equal((fa + fb) % qs, fs)
    g.ntt_inverse(fa)
    assert torch.equal(fa, a)
    # spot-check 2 polys of the big batch against the oracle
    o = OracleCtx(4096, (60, 49), 0)
    for k in (0, n - 1):
        assert np.array_equal(to_host(fs[k]), o.ntt_forward(to_host(s[k])))


@pytest.mark.parametrize("N", [4096, 8192])
def test_generic_primes_take_the_conditional_subtract_path(mm, N):
    """primes that are not 2^b - c use the kernel variant without the fold reduction"""
    from openmm_amd.lib import to_device, to_host
    q = [844424930181121, 864691128455200769, 576461851815641089]
    o, g = OracleCtx(N, q=q), mm.Context(N, q=q)
    assert g.psi == o.psi
    a = _polys(o, 5, SEED + 3)
    a[0, :, :4] = np.array(q, dtype=np.uint64)[:, None] - 1
    d = to_device(a)
    g.ntt_forward(d)
    assert np.array_equal(to_host(d), o.ntt_forward(a))
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)


def test_worst_case_magnitudes(mm):
    """all coefficients q-1 with lazy offsets: the largest values the lazy ranges must carry"""
    from openmm_amd.lib import to_device, to_host
    o, g = OracleCtx(4096, (60, 49), 0), mm.Context(4096, (60, 49), 0)
    qs = np.array(o.q, dtype=np.uint64)[None, :, None]
    a = np.broadcast_to(qs - 1, (2, 2, 4096)).copy()
    ref = o.ntt_forward(a)
    d = to_device(a + 3 * qs)          # forward accepts < 4q
    g.ntt_forward(d)
    assert np.array_equal(to_host(d), ref)
    d = to_device(ref + qs)            # inverse accepts < 2q
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)


@pytest.mark.parametrize("N", [4096, 8192])
@pytest.mark.parametrize("f64", ["1", "0"])
def test_small_moduli_double_precision_and_integer_paths_agree(mm, N, f64, monkeypatch):
    """moduli below 2^49 run their butterflies on the FP64 pipe (exact integers in doubles);
    MMH_NTT_F64=0 keeps the integer butterflies: both must equal the oracle bit for bit, on
    random data, on the all-(q-1) vector and on lazy inputs (forward < 4q, inverse < 2q)"""
    from openmm_amd.lib import to_device, to_host
    monkeypatch.setenv("MMH_NTT_F64", f64)
    # SEAL's 49-bit primes of both degrees, the largest 48-bit and 33-bit NTT primes, the BOLE field prime of l=32
    q = [562949953216513, 562949952847873, 281474976694273, 8589852673, 4293918721]
    q = [m for m in q if m % (2 * N) == 1][:4]
    assert len(q) >= 2
    o, g = OracleCtx(N, q=q), mm.Context(N, q=q)
    assert g.psi == o.psi
    a = _polys(o, 6, SEED + 11)
    qs = np.array(q, dtype=np.uint64)[None, :, None]
    a[0] = np.broadcast_to(qs[0] - 1, a[0].shape)
    a[1] = 0
    a[1, :, N - 1] = 1
    ref = o.ntt_forward(a)
    d = to_device(a + qs * (splitmix64(SEED + 5, a.size).reshape(a.shape) % np.uint64(4)))
    g.ntt_forward(d)
    assert np.array_equal(to_host(d), ref)
    d = to_device(ref + qs * (splitmix64(SEED + 6, a.size).reshape(a.shape) % np.uint64(2)))
    g.ntt_inverse(d)
    assert np.array_equal(to_host(d), a)


def test_context_calls_leave_the_current_device_alone():
    """mmh_ctx_create / destroy and every entry point run on the context's device and put the caller's current
    device back (they used to leave hipSetDevice(device) behind).  With one GPU the switch cannot happen, so the
    check is that a context for a device that does not exist fails cleanly and changes nothing; with more than
    one, a context on the last device is used while device 0 is current."""
    import torch
    import openmm_amd
    from openmm_amd.lib import MMHError, to_device, to_host
    assert torch.cuda.is_available()
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(0)
    with pytest.raises(MMHError):
        openmm_amd.Context(4096, (60, 49), 0, device=n_dev)
    assert torch.cuda.current_device() == 0
    torch.zeros(1, device="cuda").add_(1)          # the process still works on device 0
    if n_dev > 1:
        o = OracleCtx(4096, (60, 49), 0)
        g = openmm_amd.Context(4096, (60, 49), 0, device=n_dev - 1)
        assert torch.cuda.current_device() == 0
        a = np.stack([np.stack([uniform_mod(SEED + l, 4096, q) for l, q in enumerate(o.q)])])
        d = to_device(a, device=f"cuda:{n_dev - 1}")
        with torch.cuda.device(n_dev - 1):
            s = torch.cuda.current_stream().cuda_stream
        import ctypes as C
        from openmm_amd.lib import check, dptr
        check(g.lib.mmh_ntt_forward(g.handle, dptr(d), 1, 2, C.c_void_p(s)))
        assert torch.cuda.current_device() == 0
        torch.cuda.synchronize(n_dev - 1)
        assert np.array_equal(to_host(d), o.ntt_forward(a))
        g.close()
        assert torch.cuda.current_device() == 0
