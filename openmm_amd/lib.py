"""ctypes binding of libmm_hip.so (C ABI: include/mm_hip.h).  No fallback path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmm_hip.so")

CODE_NAMES = ["OK", "ERR_CONFIG", "ERR_NULL_POINTER", "ERR_DIM_MISMATCH", "ERR_SEAL_MEMORY",
              "ERR_KEY_MISSING", "ERR_OUT_BOUND", "ERR_INVALID_ARG", "ERR_INTERNAL"]


class MMHError(RuntimeError):
    """Non-OK gemini::Code from the C ABI (include/gemini/core/types.h:30-40)."""

    def __init__(self, code, text):
        self.code = code
        name = CODE_NAMES[code] if 0 <= code < len(CODE_NAMES) else str(code)
        super().__init__(f"{name}: {text}")


_lib = None
_vp, _sz, _u64, _int = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int

# name -> argtypes; every function returns int (gemini::Code) unless listed in _RESTYPE
SIGNATURES = {
    "mmh_device_count": [C.POINTER(_int)],
    "mmh_conv_plan_work": [_vp, _vp],
    "mmh_sample_reshare": [_vp, C.POINTER(C.c_uint32), _u64, _sz, _vp, _vp, _vp, _vp],
    "mmh_sample_reshare_range": [_vp, C.POINTER(C.c_uint32), _u64, _sz, _sz, _vp, _vp, _vp, _vp],
    "mmh_ctx_create": [C.POINTER(_vp), _sz, C.POINTER(_u64), _sz, _u64, _int],
    "mmh_ctx_create_bits": [C.POINTER(_vp), _sz, C.POINTER(_int), _sz, _u64, _int],
    "mmh_ctx_destroy": [_vp],
    "mmh_ctx_info": [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_u64), C.POINTER(_u64),
                     C.POINTER(_u64), C.POINTER(_u64)],
    "mmh_malloc": [C.POINTER(_vp), _sz],
    "mmh_free": [_vp],
    "mmh_host_alloc": [C.POINTER(_vp), _sz],
    "mmh_host_free": [_vp],
    "mmh_get_device": [C.POINTER(_int)],
    "mmh_set_device": [_int],
    "mmh_stream_create": [C.POINTER(_vp)],
    "mmh_stream_destroy": [_vp],
    "mmh_repack_nhwc_to_chw": [_vp, _sz, _sz, _sz, _u64, _vp, _vp],
    "mmh_repack_chw_to_nhwc": [_vp, _sz, _sz, _sz, _vp, _vp],
    "mmh_repack_hwio_to_oihw": [_vp, _sz, _sz, _sz, _sz, _u64, _vp, _vp],
    "mmh_host_register": [_vp, _sz],
    "mmh_host_unregister": [_vp],
    "mmh_scratch_release": [],
    "mmh_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "mmh_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "mmh_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "mmh_memset": [_vp, _int, _sz, _vp],
    "mmh_stream_sync": [_vp],
    "mmh_ntt_forward": [_vp, _vp, _sz, _sz, _vp],
    "mmh_ntt_inverse": [_vp, _vp, _sz, _sz, _vp],
    "mmh_ntt_forward_t": [_vp, _vp, _sz, _vp],
    "mmh_ntt_inverse_t": [_vp, _vp, _sz, _vp],
    "mmh_ntt_forward_limb": [_vp, _vp, _sz, _sz, _vp],
    "mmh_ntt_inverse_limb": [_vp, _vp, _sz, _sz, _vp],
    "mmh_poly_op": [_int, _vp, _vp, _u64, _sz, _u64, _vp, _vp],
    "mmh_add_plain": [_vp, _int, _vp, _vp, _sz, _int, _vp, _vp],
    "mmh_plain_lift_ntt": [_vp, _int, _vp, _sz, _vp, _vp],
    "mmh_dyadic_mac": [_vp, _int, _vp, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _vp],
    "mmh_mul_plain_acc": [_vp, _int, _vp, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _vp],
    "mmh_ct_add": [_vp, _int, _vp, _vp, _sz, _vp],
    "mmh_mod_switch_to_last": [_vp, _vp, _sz, _vp, _vp],
    "mmh_encrypt_zero_asym": [_vp, _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp],
    "mmh_encrypt_symmetric": [_vp, _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp],
    "mmh_make_public_key": [_vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_decrypt": [_vp, _int, _vp, _vp, _sz, _vp, _vp],
    "mmh_scale_and_round": [_vp, _int, _vp, _sz, _vp, _sz, _vp, _vp],
    "mmh_sub_rns_poly": [_vp, _int, _vp, _vp, _sz, _vp],
    "mmh_truncate_masks": [_vp, C.POINTER(_u64), C.POINTER(_u64)],
    "mmh_truncate_and_zero": [_vp, _vp, _sz, _vp, _sz, _vp],
    "mmh_batch_encode": [_vp, _vp, _sz, _vp, _vp],
    "mmh_batch_decode": [_vp, _vp, _sz, _vp, _vp],
    "mmh_batch_decode_noalloc": [_vp, _vp, _sz, _vp, _vp],
    "mmh_batch_decode_noalloc_max": [_vp, C.POINTER(_sz)],
    "mmh_chacha20_words": [C.POINTER(C.c_uint32), _u64, _u64, _vp, _sz, _vp],
    "mmh_sample_ternary": [C.POINTER(C.c_uint32), _u64, _vp, _sz, _vp],
    "mmh_sample_noise": [C.POINTER(C.c_uint32), _u64, _vp, _sz, _vp],
    "mmh_sample_uniform": [_vp, _int, C.POINTER(C.c_uint32), _u64, _vp, _sz, _vp],
    "mmh_signed_to_ntt": [_vp, _int, _vp, _sz, _vp, _vp],
    "mmh_conv_plan_create": [_vp, _vp, C.POINTER(_vp)],
    "mmh_conv_plan_destroy": [_vp],
    "mmh_conv_plan_info": [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz),
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "mmh_conv_encode_image": [_vp, _vp, _vp, _vp],
    "mmh_conv_encode_filters": [_vp, _vp, _vp, _vp],
    "mmh_conv_encrypt_image": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_bn_tensor_count": [_vp, _vp, _vp, _vp],
    "mmh_bn_encode_tensor": [_vp, _vp, _vp, _vp, _vp],
    "mmh_bn_encrypt_tensor": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_bn_direct_server": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_bn_decrypt_tensor": [_vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server_exact": [_vp] * 13,
    "mmh_fc_server_exact": [_vp] * 13,
    "mmh_flood_add": [_vp, _vp, _vp, _sz, _vp],
    "mmh_reduce_mod_t": [_vp, _vp, _vp, _sz, _vp],
    "mmh_conv_server_weights": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server_tensors": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server_tensors_ev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server_tensors_sampled": [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32), _u64, _sz, _vp, _vp, _vp, _vp, _vp],
    "mmh_event_create": [C.POINTER(_vp)],
    "mmh_event_destroy": [_vp],
    "mmh_event_record": [_vp, _vp],
    "mmh_event_sync": [_vp],
    "mmh_conv_check_filters": [_vp, _vp, _vp, _vp],
    "mmh_conv_filters_prepare": [_vp, _vp, C.POINTER(_vp), _vp],
    "mmh_conv_filters_destroy": [_vp],
    "mmh_conv_server_prepared": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_server_prepared_tensors": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_bn_encode_vector_r2f": [_vp, C.c_int, _vp, _vp, _vp, _sz, C.c_int, _vp, _vp],
    "mmh_bn_encrypt_vector_r2f": [_vp, _vp, C.c_int, _vp, _vp, _vp, _sz, C.c_int, _vp, _vp, _vp, _vp],
    "mmh_bn_decrypt_vector_f2r": [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _vp, _vp],
    "mmh_reshare_zero_prepare": [_vp, _vp, _vp, _sz, _vp, _vp],
    "mmh_conv_server_online": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_conv_decrypt": [_vp, _vp, _vp, _vp, _vp],
    "mmh_ct_split_c0": [_vp, C.c_int, _vp, _sz, _vp, _vp],
    "mmh_ct_join": [_vp, C.c_int, _vp, _vp, _sz, _vp, _vp],
    "mmh_conv_result_packed_words": [_vp, C.POINTER(_sz)],
    "mmh_conv_result_pack": [_vp, _vp, _vp, _vp],
    "mmh_conv_result_unpack": [_vp, _vp, _vp, _vp],
    "mmh_fc_plan_create": [_vp, _vp, C.POINTER(_vp)],
    "mmh_fc_plan_create_split": [_vp, _vp, C.POINTER(C.c_int64), C.POINTER(_vp)],
    "mmh_fc_plan_destroy": [_vp],
    "mmh_fc_split": [C.c_int64, C.c_int64, _sz, C.POINTER(C.c_int64)],
    "mmh_fc_plan_info": [_vp, C.POINTER(C.c_int64), C.POINTER(_sz), C.POINTER(_sz)],
    "mmh_fc_encode_input": [_vp, _vp, _vp, _vp],
    "mmh_fc_encode_weights": [_vp, _vp, _vp, _vp],
    "mmh_fc_encrypt_input": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_fc_server": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_fc_decrypt": [_vp, _vp, _vp, _vp, _vp],
    "mmh_bn_encode_vector": [_vp, _vp, _sz, _vp, _vp],
    "mmh_bn_encrypt_vector": [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp],
    "mmh_bn_server": [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp],
    "mmh_bole_server": [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp],
    "mmh_bn_decrypt_vector": [_vp, _vp, _vp, _sz, _vp, _vp],
    "mmh_r2f_pre": [_int, _vp, _sz, _int, _u64, _vp, _vp, _vp, _vp],
    "mmh_r2f_post": [_int, _vp, _vp, _vp, _sz, _int, _u64, _vp, _vp],
    "mmh_f2r_pre": [_int, _vp, _sz, _u64, _int, _vp, _vp, _vp, _vp],
    "mmh_f2r_post": [_int, _vp, _vp, _vp, _sz, _u64, _int, _vp, _vp],
    "mmh_f2r_trunc_post": [_int, _vp, _vp, _vp, _sz, _u64, _int, _int, _vp, _vp],
    "mmh_r2r_pre": [_int, _vp, _sz, _int, _vp, _vp, _vp, _vp],
    "mmh_r2r_post": [_int, _vp, _vp, _vp, _sz, _int, _int, _vp, _vp],
    "mmh_r2r128_post": [_int, _vp, _vp, _vp, _sz, _int, _int, _vp, _vp],
    "mmh_relu_pre": [_int, _vp, _sz, _int, _vp, _vp, _vp],
    "mmh_relu_post": [_int, _vp, _vp, _sz, _vp, _vp, _vp],
    "mmh_mux_pre": [_vp, _vp, _sz, _int, _vp, _vp],
    "mmh_mux_post": [_vp, _vp, _vp, _vp, _sz, _int, _vp, _vp],
    "mmh_b2a_pre": [_vp, _sz, _int, _vp, _vp],
    "mmh_b2a_post": [_int, _vp, _vp, _sz, _int, _vp, _vp],
    "mmh_trunc_pre": [_int, _vp, _sz, _int, _int, _int, _vp, _vp, _vp],
    "mmh_trunc_post": [_int, _vp, _vp, _sz, _int, _int, _int, _int, _vp, _vp],
    "mmh_trunc_reduce_pre": [_vp, _sz, _int, _vp, _vp],
    "mmh_trunc_reduce_post": [_vp, _vp, _sz, _int, _int, _vp, _vp],
    "mmh_fused_bn_filters": [_vp, _vp, _int, _int, _int, _int, _int, _int, _u64, _int, _vp, _vp, _vp],
    "mmh_fused_bn_bias": [_vp, _vp, _sz, _int, _int, _vp],
    "mmh_bole_cross_term": [_vp, _vp, _vp, _sz, _u64, _vp],
    "mmh_cot_wire_bytes": [_sz, _int, _u64, _int, C.POINTER(_sz)],
    "mmh_cot_gids": [_sz, C.POINTER(_u64)],
    "mmh_cot_send": [_vp, C.POINTER(_u64), C.POINTER(_u64), _u64, _vp, _sz, _int, _u64, _int, _vp, _vp, _vp],
    "mmh_cot_send_f2r": [_vp, C.POINTER(_u64), C.POINTER(_u64), _u64, _vp, _sz, _u64, _int, _int, _vp, _vp, _vp],
    "mmh_cot_send_r2f_encode": [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64), _u64, _vp, _sz, _int, _vp, _vp, _vp],
    "mmh_cot_recv": [_vp, _vp, C.POINTER(_u64), _u64, _vp, _sz, _int, _u64, _int, _vp, _vp],
    "mmh_bn_alice_prepare": [_vp, _vp, _vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), _u64, _vp, _sz, _int, _int,
                             _vp, C.POINTER(C.c_uint32), _u64, _vp, _vp, _vp, _vp, _vp, _vp],
    "mmh_bn_alice_finish": [_vp, _vp, _vp, _sz, _vp, _vp],
}


def load_library():
    """Load libmm_hip.so; raise loudly when it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C openmm_amd/csrc` "
            "(or __graft_entry__.build()). openmm_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _int
    lib.mmh_last_error.restype = C.c_char_p
    lib.mmh_last_error.argtypes = []
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise MMHError(code, load_library().mmh_last_error().decode())


def _stream_ptr():
    import torch
    return _vp(torch.cuda.current_stream().cuda_stream)


class DeferredFlag:
    """A boolean verdict computed on the device and read back without draining the launch queue:
    the value is copied to pinned host memory right away, the caller enqueues its kernels, and only
    then waits for that early copy (not for the kernels behind it).  The reference decides the same
    conditions on host data before it starts (hom_conv2d_ss.cc:573-580, hom_fc_ss.cc:414-422)."""

    def __init__(self, flag_dev, negate=False):
        import torch
        self.negate = negate
        self.host = torch.empty((), dtype=flag_dev.dtype, pin_memory=True)
        self.host.copy_(flag_dev.reshape(()), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def __bool__(self):
        self.event.synchronize()
        return bool(self.host) != self.negate


class VerdictRing:
    """Page-locked words the kernels store early verdicts into, each with an event recorded right behind the launch
    that settles it (mmh_conv_server_tensors_ev): the host waits for that event alone.  A ring, so that a slot is
    not reused while a call that writes it may still be in flight (callers read their verdict before they return)."""

    def __init__(self, slots=32):
        import torch
        self.lib = load_library()
        self.words = torch.zeros(slots, dtype=torch.int32, pin_memory=True)
        self.base = self.words.data_ptr()
        self.events = []
        for _ in range(slots):
            e = _vp()
            check(self.lib.mmh_event_create(C.byref(e)))
            self.events.append(e)
        self.next = 0

    def take(self):
        i = self.next
        self.next = (i + 1) % len(self.events)
        return i, _vp(self.base + 4 * i), self.events[i]

    def read(self, i):
        check(self.lib.mmh_event_sync(self.events[i]))
        return int(self.words[i])

    def __del__(self):
        try:
            for e in self.events:
                self.lib.mmh_event_destroy(e)
        except Exception:
            pass


def dptr(t):
    """Device pointer of a contiguous torch CUDA tensor (None -> NULL)."""
    if t is None:
        return _vp(0)
    if not (t.is_cuda and t.is_contiguous()):   # a host or strided pointer would fault inside a kernel
        raise MMHError(7, "dptr: expected a contiguous CUDA tensor")
    return _vp(t.data_ptr())


def to_device(arr, device="cuda:0"):
    """numpy uint64 array -> torch int64 CUDA tensor with the same bits."""
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    """torch int64 tensor -> numpy uint64 array with the same bits."""
    return t.detach().cpu().numpy().view(np.uint64)


class Context:
    """seal::SEALContext stand-in for the hot path (cheetah-api.cpp:95-105,:151-162)."""

    def __init__(self, N=4096, bits=(60, 49), t=0, q=None, device=0):
        self.lib = load_library()
        self.handle = _vp()
        if q is not None:
            arr = (_u64 * len(q))(*q)
            check(self.lib.mmh_ctx_create(C.byref(self.handle), N, arr, len(q), t, device))
        else:
            arr = (_int * len(bits))(*bits)
            check(self.lib.mmh_ctx_create_bits(C.byref(self.handle), N, arr, len(bits), t, device))
        n, l, tt, g = _sz(), _sz(), _u64(), _u64()
        qs, psis = (_u64 * 4)(), (_u64 * 4)()
        check(self.lib.mmh_ctx_info(self.handle, C.byref(n), C.byref(l), qs, psis, C.byref(tt),
                                    C.byref(g)))
        self.N, self.L, self.t, self.gamma = n.value, l.value, tt.value, g.value
        self.q = [int(qs[i]) for i in range(self.L)]
        self.psi = [int(psis[i]) for i in range(self.L)]
        self.device = device

    def close(self):
        if self.handle:
            self.lib.mmh_ctx_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # seal::util::ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey, batched, in place
    def ntt_forward(self, polys, limbs=None):
        limbs = self.L if limbs is None else limbs
        assert polys.numel() % (limbs * self.N) == 0
        check(self.lib.mmh_ntt_forward(self.handle, dptr(polys), polys.numel() // (limbs * self.N),
                                       limbs, _stream_ptr()))
        return polys

    def ntt_inverse(self, polys, limbs=None):
        limbs = self.L if limbs is None else limbs
        assert polys.numel() % (limbs * self.N) == 0
        check(self.lib.mmh_ntt_inverse(self.handle, dptr(polys), polys.numel() // (limbs * self.N),
                                       limbs, _stream_ptr()))
        return polys

    def ntt_forward_t(self, polys):
        check(self.lib.mmh_ntt_forward_t(self.handle, dptr(polys), polys.numel() // self.N,
                                         _stream_ptr()))
        return polys

    def ntt_inverse_t(self, polys):
        check(self.lib.mmh_ntt_inverse_t(self.handle, dptr(polys), polys.numel() // self.N,
                                         _stream_ptr()))
        return polys

    # ------------------------------------------------------------------ BFV batches
    def limbs(self, lv):
        return self.L if lv == 0 else 1

    def _empty(self, *shape):
        import torch
        return torch.empty(shape, dtype=torch.int64, device=f"cuda:{self.device}")

    def add_plain(self, cts, pts, lv=0, sub=False):
        """seal::Evaluator::add_plain / sub_plain on [n,2,limbs,N] and [n,N]; returns new cts"""
        out = self._empty(*cts.shape)
        check(self.lib.mmh_add_plain(self.handle, lv, dptr(cts), dptr(pts), cts.shape[0], int(sub),
                                     dptr(out), _stream_ptr()))
        return out

    def plain_lift_ntt(self, pts, lv=0):
        out = self._empty(pts.shape[0], self.limbs(lv), self.N)
        check(self.lib.mmh_plain_lift_ntt(self.handle, lv, dptr(pts), pts.shape[0], dptr(out), _stream_ptr()))
        return out

    def mul_plain_acc(self, cts, pts, n_o1, n_o2, K, ct_sk, ct_s2, pt_s1, pt_sk, pt_s2, lv=0):
        out = self._empty(n_o1 * n_o2, 2, self.limbs(lv), self.N)
        check(self.lib.mmh_mul_plain_acc(self.handle, lv, dptr(cts), cts.shape[0], dptr(pts), pts.shape[0],
                                         n_o1, n_o2, K, ct_sk, ct_s2, pt_s1, pt_sk, pt_s2, dptr(out),
                                         _stream_ptr()))
        return out

    def ct_add(self, a, b, lv=0):
        check(self.lib.mmh_ct_add(self.handle, lv, dptr(a), dptr(b), a.shape[0], _stream_ptr()))
        return a

    def mod_switch_to_last(self, cts):
        out = self._empty(cts.shape[0], 2, 1, self.N)
        check(self.lib.mmh_mod_switch_to_last(self.handle, dptr(cts), cts.shape[0], dptr(out), _stream_ptr()))
        return out

    def encrypt_zero_asym(self, pk_ntt, u, e0, e1, lv=0):
        out = self._empty(u.shape[0], 2, self.limbs(lv), self.N)
        check(self.lib.mmh_encrypt_zero_asym(self.handle, lv, dptr(pk_ntt), dptr(u), dptr(e0), dptr(e1),
                                             u.shape[0], dptr(out), _stream_ptr()))
        return out

    def encrypt_symmetric(self, sk_ntt, a, e, pts, lv=0):
        out = self._empty(a.shape[0], 2, self.limbs(lv), self.N)
        check(self.lib.mmh_encrypt_symmetric(self.handle, lv, dptr(sk_ntt), dptr(a), dptr(e), dptr(pts),
                                             a.shape[0], dptr(out), _stream_ptr()))
        return out

    def make_public_key(self, sk_ntt, a_ntt, e):
        out = self._empty(2, self.L, self.N)
        check(self.lib.mmh_make_public_key(self.handle, dptr(sk_ntt), dptr(a_ntt), dptr(e), dptr(out),
                                           _stream_ptr()))
        return out

    def decrypt(self, sk_ntt, cts, lv=0):
        out = self._empty(cts.shape[0], self.N)
        check(self.lib.mmh_decrypt(self.handle, lv, dptr(sk_ntt), dptr(cts), cts.shape[0], dptr(out),
                                   _stream_ptr()))
        return out

    def scale_and_round(self, polys, targets=None, lv=0):
        nt = self.N if targets is None else targets.numel()
        out = self._empty(polys.shape[0], nt)
        check(self.lib.mmh_scale_and_round(self.handle, lv, dptr(polys), polys.shape[0], dptr(targets), nt,
                                           dptr(out), _stream_ptr()))
        return out

    def sub_rns_poly(self, cts, r, lv=0):
        check(self.lib.mmh_sub_rns_poly(self.handle, lv, dptr(cts), dptr(r), cts.shape[0], _stream_ptr()))
        return cts

    def truncate_masks(self):
        m0, m1 = _u64(), _u64()
        check(self.lib.mmh_truncate_masks(self.handle, C.byref(m0), C.byref(m1)))
        return m0.value, m1.value

    def truncate_and_zero(self, cts, keep=None, keep_period=1):
        check(self.lib.mmh_truncate_and_zero(self.handle, dptr(cts), cts.shape[0], dptr(keep), keep_period,
                                             _stream_ptr()))
        return cts

    def batch_encode(self, vals):
        out = self._empty(*vals.shape)
        check(self.lib.mmh_batch_encode(self.handle, dptr(vals), vals.numel() // self.N, dptr(out),
                                        _stream_ptr()))
        return out

    def batch_decode(self, pts):
        out = self._empty(*pts.shape)
        check(self.lib.mmh_batch_decode(self.handle, dptr(pts), pts.numel() // self.N, dptr(out),
                                        _stream_ptr()))
        return out


class Sampler:
    """ChaCha20-keyed device sampler: every call consumes a fresh stream id."""

    def __init__(self, ctx, key=None):
        self.ctx = ctx
        key = os.urandom(32) if key is None else key
        assert len(key) == 32
        self.key = (C.c_uint32 * 8).from_buffer_copy(key)
        self.next_stream = 1

    def _sid(self):
        s = self.next_stream
        self.next_stream += 1
        return s

    def words(self, n):
        out = self.ctx._empty(n)
        check(self.ctx.lib.mmh_chacha20_words(self.key, self._sid(), 0, dptr(out), n, _stream_ptr()))
        return out

    def ternary(self, n_polys):
        out = self.ctx._empty(n_polys, self.ctx.N)
        check(self.ctx.lib.mmh_sample_ternary(self.key, self._sid(), dptr(out), out.numel(), _stream_ptr()))
        return out

    def noise(self, n_polys):
        out = self.ctx._empty(n_polys, self.ctx.N)
        check(self.ctx.lib.mmh_sample_noise(self.key, self._sid(), dptr(out), out.numel(), _stream_ptr()))
        return out

    def uniform(self, n_polys, lv=0):
        out = self.ctx._empty(n_polys, self.ctx.limbs(lv), self.ctx.N)
        check(self.ctx.lib.mmh_sample_uniform(self.ctx.handle, lv, self.key, self._sid(), dptr(out), n_polys,
                                              _stream_ptr()))
        return out

    def reshare(self, n_polys):
        """(u, e1, mask) of n re-sharings in one launch = ternary, noise, uniform(lv=1) on three consecutive streams"""
        u, e1, mask = (self.ctx._empty(n_polys, self.ctx.N) for _ in range(3))
        sid = self.next_stream
        self.next_stream += 3
        check(self.ctx.lib.mmh_sample_reshare(self.ctx.handle, self.key, sid, n_polys, dptr(u), dptr(e1), dptr(mask),
                                              _stream_ptr()))
        return u, e1, mask

    def reshare_streams(self):
        """the stream id a call that draws a re-sharing's randomness itself should use (three consecutive streams)"""
        sid = self.next_stream
        self.next_stream += 3
        return sid

    def reshare_range(self, sid, first_poly, n_polys):
        """polynomials [first_poly, first_poly + n_polys) of the draw reshare() makes on streams sid .. sid + 2"""
        u, e1, mask = (self.ctx._empty(n_polys, self.ctx.N) for _ in range(3))
        check(self.ctx.lib.mmh_sample_reshare_range(self.ctx.handle, self.key, sid, first_poly, n_polys, dptr(u),
                                                    dptr(e1), dptr(mask), _stream_ptr()))
        return u, e1, mask

    def keygen(self):
        """KeyGenerator: ternary secret (NTT form) and pk = (-(a s + e), a) (cheetah-api.cpp:107-114)"""
        s = self.ternary(1)
        sk = self.ctx._empty(self.ctx.L, self.ctx.N)
        check(self.ctx.lib.mmh_signed_to_ntt(self.ctx.handle, 0, dptr(s), 1, dptr(sk), _stream_ptr()))
        pk = self.ctx.make_public_key(sk, self.uniform(1)[0].contiguous(), self.noise(1)[0].contiguous())
        return sk, pk
