"""ctypes binding of oracle/libmm_oracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libmm_oracle.so")

u64, sz, ci, vp = C.c_uint64, C.c_size_t, C.c_int, C.c_void_p
MAX_L = 4


class Ctx(C.Structure):
    _fields_ = [
        ("N", sz), ("logN", ci), ("L", sz), ("q", u64 * MAX_L), ("psi", u64 * MAX_L),
        ("psi_rev", vp * MAX_L), ("ipsi_rev", vp * MAX_L),
        ("psi_rev_sh", vp * MAX_L), ("ipsi_rev_sh", vp * MAX_L),
        ("t", u64), ("t_is_pow2", ci),
        ("delta_mod_q", (u64 * MAX_L) * 2), ("q_mod_t", u64 * 2), ("upper_half_threshold", u64),
        ("gamma", u64), ("prod_t_gamma_mod_q", (u64 * MAX_L) * 2),
        ("inv_punct_mod_q", (u64 * MAX_L) * 2), ("punct_mod_t", (u64 * MAX_L) * 2),
        ("punct_mod_gamma", (u64 * MAX_L) * 2),
        ("neg_inv_q_mod_t", u64 * 2), ("neg_inv_q_mod_gamma", u64 * 2), ("inv_gamma_mod_t", u64),
        ("inv_qlast_mod_q", u64 * MAX_L),
        ("batching", ci), ("t_psi", u64), ("t_psi_rev", vp), ("t_ipsi_rev", vp), ("slot_map", vp),
    ]


def A(x):
    """numpy array -> void* (keeps no reference: pass live arrays)."""
    return x.ctypes.data_as(vp)


def U(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(SO)
        for name in ["mmo_mulmod", "mmo_powmod", "mmo_invmod", "mmo_min_primitive_root", "mmo_shoup",
                     "mmo_modulo_mult"]:
            getattr(lib, name).restype = u64
        lib.mmo_mulmod.argtypes = [u64, u64, u64]
        lib.mmo_powmod.argtypes = [u64, u64, u64]
        lib.mmo_invmod.argtypes = [u64, u64]
        lib.mmo_min_primitive_root.argtypes = [u64, u64]
        lib.mmo_shoup.argtypes = [u64, u64]
        lib.mmo_modulo_mult.argtypes = [u64, u64, u64]
        lib.mmo_is_prime.argtypes = [u64]
        lib.mmo_get_primes.argtypes = [u64, ci, sz, vp]
        lib.mmo_coeff_modulus_create.argtypes = [sz, vp, sz, vp]
        lib.mmo_ntt_tables.argtypes = [sz, u64, u64, vp, vp]
        for n in ["mmo_ntt_forward", "mmo_ntt_inverse"]:
            getattr(lib, n).argtypes = [vp, sz, u64, vp]
            getattr(lib, n).restype = None
        for n in ["mmo_ntt_forward_fast", "mmo_ntt_inverse_fast"]:
            getattr(lib, n).argtypes = [vp, sz, u64, vp, vp]
            getattr(lib, n).restype = None
        lib.mmo_negacyclic_schoolbook.argtypes = [vp, vp, vp, sz, u64]
        for n in ["mmo_dyadic_mul", "mmo_add_mod", "mmo_sub_mod"]:
            getattr(lib, n).argtypes = [vp, vp, vp, sz, u64]
        lib.mmo_ctx_create.restype = C.POINTER(Ctx)
        lib.mmo_ctx_create.argtypes = [sz, vp, sz, u64]
        lib.mmo_ctx_create_q.restype = C.POINTER(Ctx)
        lib.mmo_ctx_create_q.argtypes = [sz, vp, sz, u64]
        lib.mmo_ctx_destroy.argtypes = [C.POINTER(Ctx)]
        P = C.POINTER(Ctx)
        lib.mmo_plain_lift.argtypes = [P, ci, vp, sz, vp]
        lib.mmo_multiply_plain.argtypes = [P, ci, vp, vp, sz, vp]
        lib.mmo_add_plain.argtypes = [P, ci, vp, vp, sz]
        lib.mmo_sub_plain.argtypes = [P, ci, vp, vp, sz]
        lib.mmo_ct_add.argtypes = [P, ci, vp, vp, sz]
        lib.mmo_mod_switch_to_last.argtypes = [P, vp, vp, sz]
        lib.mmo_decrypt_scale_and_round.argtypes = [P, ci, vp, sz, vp]
        lib.mmo_encrypt_symmetric.argtypes = [P, ci, vp, vp, vp, vp, sz, vp]
        lib.mmo_make_public_key.argtypes = [P, vp, vp, vp, vp]
        lib.mmo_encrypt_zero_asymmetric.argtypes = [P, ci, vp, vp, vp, vp, vp]
        lib.mmo_decrypt.argtypes = [P, ci, vp, vp, vp]
        lib.mmo_truncate_masks.argtypes = [P, vp, vp]
        lib.mmo_batch_encode.argtypes = [P, vp, sz, vp]
        lib.mmo_batch_decode.argtypes = [P, vp, vp]
        lib.mmo_r2f_pre.argtypes = [ci, vp, sz, ci, u64, vp, vp, vp]
        lib.mmo_r2f_post.argtypes = [ci, vp, vp, vp, sz, ci, u64, vp]
        lib.mmo_f2r_pre.argtypes = [ci, vp, sz, u64, ci, vp, vp, vp]
        lib.mmo_f2r_post.argtypes = [ci, vp, vp, vp, sz, u64, ci, vp]
        lib.mmo_f2r_trunc_post.argtypes = [ci, vp, vp, vp, sz, u64, ci, ci, vp]
        lib.mmo_r2r_pre.argtypes = [ci, vp, sz, ci, vp, vp, vp]
        lib.mmo_r2r_post.argtypes = [ci, vp, vp, vp, sz, ci, ci, vp]
        lib.mmo_r2r128_post.argtypes = [ci, vp, vp, vp, sz, ci, ci, vp]
        lib.mmo_trunc_pre.argtypes = [ci, vp, sz, ci, ci, ci, vp, vp]
        lib.mmo_trunc_post.argtypes = [ci, vp, vp, sz, ci, ci, ci, ci, vp]
        lib.mmo_trunc_reduce_pre.argtypes = [vp, sz, ci, vp]
        lib.mmo_trunc_reduce_post.argtypes = [vp, vp, sz, ci, ci, vp]
        lib.mmo_relu_pre.argtypes = [ci, vp, sz, ci, vp, vp]
        lib.mmo_relu_post.argtypes = [ci, vp, vp, sz, vp, vp]
        lib.mmo_mux_pre.argtypes = [vp, vp, sz, ci, vp]
        lib.mmo_mux_post.argtypes = [vp, vp, vp, vp, sz, ci, vp]
        lib.mmo_b2a_pre.argtypes = [vp, sz, ci, vp]
        lib.mmo_b2a_post.argtypes = [ci, vp, vp, sz, ci, vp]
        lib.mmo_ideal_cot.argtypes = [vp, vp, vp, sz, ci, u64, ci, vp]
        lib.mmo_bole_cross_term.argtypes = [vp, vp, vp, sz, u64]
        lib.mmo_aes128_expand.argtypes = [vp, vp]
        lib.mmo_aes128_encrypt.argtypes = [vp, vp, vp]
        lib.mmo_mitccrh_hash.argtypes = [vp, u64, vp, sz, ci, vp]
        lib.mmo_cot_gids.argtypes = [sz]
        lib.mmo_cot_gids.restype = u64
        lib.mmo_cot_wire_bytes.argtypes = [sz, ci, u64, ci]
        lib.mmo_cot_wire_bytes.restype = sz
        lib.mmo_cot_send.argtypes = [vp, vp, vp, u64, vp, sz, ci, u64, ci, vp, vp]
        lib.mmo_cot_recv.argtypes = [vp, vp, vp, u64, vp, sz, ci, u64, ci, vp]
        lib.mmo_have_avx512.restype = ci
        lib.mmo_set_vector_isa.argtypes = [ci]
        lib.mmo_set_vector_isa.restype = ci
        lib.mmo_ntt_forward_batch.argtypes = [P, vp, sz, sz, ci]
        lib.mmo_ntt_inverse_batch.argtypes = [P, vp, sz, sz, ci]
        lib.mmo_conv_mac_reference.argtypes = [P, vp, vp, sz, sz, sz, vp, ci]
        _lib = lib
    return _lib


class OracleCtx:
    """Convenience wrapper around mmo_ctx."""

    def __init__(self, N=4096, bits=(60, 49), t=0, q=None):
        self.lib = load()
        if q is not None:
            arr = U(q)
            self.p = self.lib.mmo_ctx_create_q(N, A(arr), len(q), t)
        else:
            arr = np.asarray(bits, dtype=np.int32)
            self.p = self.lib.mmo_ctx_create(N, A(arr), len(bits), t)
        assert self.p, "mmo_ctx_create failed"
        c = self.p.contents
        self.N, self.L, self.t = c.N, c.L, c.t
        self.q = [int(c.q[i]) for i in range(self.L)]
        self.psi = [int(c.psi[i]) for i in range(self.L)]
        self.gamma = int(c.gamma)

    def table(self, name, l=None):
        c = self.p.contents
        ptr = getattr(c, name) if l is None else getattr(c, name)[l]
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(u64)), shape=(self.N,))

    def limbs(self, lv):
        return self.L if lv == 0 else 1

    def ntt_forward(self, polys, limbs=None):
        """polys: uint64 array [..., limbs, N], returns a transformed copy."""
        limbs = self.L if limbs is None else limbs
        out = U(polys).copy().reshape(-1, limbs, self.N)
        for k in range(out.shape[0]):
            for l in range(limbs):
                self.lib.mmo_ntt_forward_fast(A(out[k, l]), self.N, self.q[l],
                                              A(self.table("psi_rev", l)), A(self.table("psi_rev_sh", l)))
        return out.reshape(np.shape(polys))

    def ntt_inverse(self, polys, limbs=None):
        limbs = self.L if limbs is None else limbs
        out = U(polys).copy().reshape(-1, limbs, self.N)
        for k in range(out.shape[0]):
            for l in range(limbs):
                self.lib.mmo_ntt_inverse_fast(A(out[k, l]), self.N, self.q[l],
                                              A(self.table("ipsi_rev", l)), A(self.table("ipsi_rev_sh", l)))
        return out.reshape(np.shape(polys))

    def __del__(self):
        try:
            self.lib.mmo_ctx_destroy(self.p)
        except Exception:
            pass


def splitmix64(seed, n):
    """Deterministic uint64 stream (seed 0x4D264D)."""
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64(seed)
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = x + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out[:] = z ^ (z >> np.uint64(31))
    return out


def uniform_mod(seed, shape, q):
    """Deterministic values in [0,q) from splitmix64."""
    n = int(np.prod(shape))
    return (splitmix64(seed, n) % np.uint64(q)).reshape(shape)
