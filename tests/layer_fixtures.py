"""Seeded inputs shared by the layer tests (splitmix64 from 0x4D264D)."""
import numpy as np

from tests.oracle_binding import A, splitmix64, uniform_mod

SEED = 0x4D264D


def keypair(o, seed=SEED):
    """ternary secret in NTT form + public key from seeded randomness (oracle side)"""
    N, L = o.N, o.L
    s = (splitmix64(seed, N) % np.uint64(3)).astype(np.int64) - 1
    sk = np.stack([np.where(s < 0, np.uint64(q - 1), s.astype(np.uint64)) for q in o.q]).astype(np.uint64)
    sk = o.ntt_forward(sk[None])[0]
    a_ntt = np.stack([uniform_mod(seed + 1 + l, N, q) for l, q in enumerate(o.q)])
    e = noise(seed + 9, N)
    pk = np.zeros((2, L, N), np.uint64)
    o.lib.mmo_make_public_key(o.p, A(sk), A(a_ntt), A(e), A(pk))
    return sk, pk


def noise(seed, shape):
    r = splitmix64(seed, int(np.prod(shape)))
    return ((r & np.uint64(7)).astype(np.int64) - (r >> np.uint64(61)).astype(np.int64)).reshape(shape)


def ternary(seed, shape):
    return ((splitmix64(seed, int(np.prod(shape))) % np.uint64(3)).astype(np.int64) - 1).reshape(shape)


def rns_uniform(o, seed, n, limbs):
    N = o.N
    return np.stack([np.stack([uniform_mod(seed + 13 * k + l, N, o.q[l]) for l in range(limbs)])
                     for k in range(n)])


def signed_weights(seed, shape, bound, t):
    """|w| < bound, stored mod t (getRingElt, library_fixed_uniform_cheetah.cpp)"""
    w = (splitmix64(seed, int(np.prod(shape))) % np.uint64(2 * bound - 1)).astype(np.int64) - (bound - 1)
    return np.where(w < 0, np.uint64(t) - (-w).astype(np.uint64), w.astype(np.uint64)).astype(np.uint64).reshape(shape)
