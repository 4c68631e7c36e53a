"""Multi-GPU sharding of the path's independent units.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests and in single-GPU rehearsals).  Units never interact: output filters of a HomConv layer
(hom_conv2d_ss.cc:679-690), row blocks of HomFC (hom_fc_ss.cc:443-457), ciphertext indices of
BN/BOLE (hom_bn_ss.cc:261-271, :302-309), element ranges of the conversions
(library_fixed_uniform_cheetah.cpp:494-511).  Each rank computes a contiguous range of units on
replicated inputs with no data-path collective; the only exchange is the reassembly of the per-rank
results on the rank that serialises them for the network, in one of two ways:

* ``gather``: ONE RCCL gather per layer straight into a pre-sized buffer on rank 0 (every rank computes into one
  flat block [ciphertexts | share] of the common shard size, so nothing is padded, listed or concatenated afterwards);
* ``d2h``: no collective at all: every rank copies its slice device-to-host, asynchronously, to its offset
  in one page-locked host buffer that all ranks map (the serialiser reads from host memory anyway).
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_units, world, rank):
    """Contiguous range of rank `rank`, split like the reference's LaunchWorks
    (hom_conv2d_ss.cc:41-45): ceil(n/world) units per worker, trailing workers may be empty."""
    per = (n_units + world - 1) // world
    start = min(rank * per, n_units)
    return start, min(start + per, n_units)


def _collective_device(t, group):
    """gloo moves host memory: a CUDA tensor is staged through the host for it (single-GPU rehearsals and
    CPU tests); nccl takes the device tensor as it is"""
    return t.cpu() if t.is_cuda and dist.get_backend(group) == "gloo" else t


def gather_into(local, out, dst=0, group=None):
    """Gather equal-sized per-rank blocks onto rank `dst`, written in place into `out`
    ([world * local.shape[0], ...], pre-sized by the caller; None on the other ranks)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        out.copy_(local)
        return out
    staged = _collective_device(local, group)
    if rank == dst:
        if staged is local:
            dist.gather(local, list(out.chunk(world)), dst=dst, group=group)   # views of `out`: no copy after
        else:
            host = torch.empty((world,) + tuple(local.shape), dtype=local.dtype)
            dist.gather(staged, list(host.unbind(0)), dst=dst, group=group)
            out.copy_(host.reshape(out.shape))
        return out
    dist.gather(staged, None, dst=dst, group=group)
    return None


def gather_units(local, n_units, dst=0, group=None):
    """Gather per-rank unit slices (dim 0) onto rank `dst`; returns the full tensor there, None
    elsewhere.  Ragged / empty last shards are brought to the common size for the collective."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return local
    per = (n_units + world - 1) // world
    if local.shape[0] != per:
        padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
        local = padded
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) \
        if rank == dst else None
    out = gather_into(local.contiguous(), out, dst, group)
    return None if out is None else out[:n_units]


class SharedPinned:
    """One host buffer of int64 words mapped by every rank of the group (a file in /dev/shm that rank 0 creates
    and unlinks once everyone has it open) and page-locked in each process, so that device-to-host copies into
    it run asynchronously at the link's rate."""
    _seq = 0

    def __init__(self, words, group=None):
        rank = dist.get_rank(group)
        SharedPinned._seq += 1
        words = (words + 511) // 512 * 512          # whole pages
        name = [f"/dev/shm/mmh_{os.getpid()}_{SharedPinned._seq}" if rank == 0 else None]
        if rank == 0:
            with open(name[0], "wb") as f:
                f.truncate(words * 8)
        dist.broadcast_object_list(name, src=0, group=group)
        self.t = torch.from_file(name[0], shared=True, size=words, dtype=torch.int64)
        self.registered = False
        if torch.cuda.is_available():
            from . import lib as L
            self.t.zero_()                          # touch the pages before locking them
            self.registered = L.load_library().mmh_host_register(L._vp(self.t.data_ptr()), words * 8) == 0
        dist.barrier(group)
        if rank == 0:
            os.unlink(name[0])

    def close(self):
        if self.registered:
            from . import lib as L
            L.load_library().mmh_host_unregister(L._vp(self.t.data_ptr()))
            self.registered = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedConv:
    """HomConv2DSS::conv2DSS with the output filters split over the ranks: every rank runs the one-call server
    form (encodeImage + encodeFilters + conv2DSS, cheetah-api.cpp:401-445) on its filter slice.

    mode "gather": results are assembled on rank 0 in device memory by one collective into a pre-sized buffer;
    mode "d2h":    every rank copies its slice into a host buffer shared by all ranks (no collective)."""

    def __init__(self, server, meta, group=None, mode="gather"):
        import dataclasses
        assert mode in ("gather", "d2h")
        self.server, self.meta, self.group, self.mode = server, meta, group, mode
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lo, self.hi = shard_range(meta.n_filters, self.world, self.rank)
        self.per = (meta.n_filters + self.world - 1) // self.world
        self.local_meta = dataclasses.replace(meta, n_filters=max(self.hi - self.lo, 1))
        self._bufs = None
        self._host = None

    def _buffers(self, like, n_sp, out_shape, ct_tail):
        """One flat per-rank result block [ciphertexts | share] of the common shard size, so that ONE collective (or one
        host copy) moves both; on rank 0 the assembly block [world][ciphertexts | share] (gather) or the shared host
        buffer (d2h).  Made once per layer object.  Returns (block, cts view, share view, assembly)."""
        if self._bufs is None:
            n_c = self.per * n_sp * int(torch.Size(ct_tail).numel())
            n_s = self.per * int(torch.Size(out_shape[1:]).numel())
            block = torch.zeros(n_c + n_s, dtype=like.dtype, device=like.device)
            cts = block[:n_c].view(self.per * n_sp, *ct_tail)
            share = block[n_c:].view(self.per, *out_shape[1:])
            assembly = None
            if self.mode == "gather" and self.rank == 0:
                assembly = torch.zeros(self.world, n_c + n_s, dtype=like.dtype, device=like.device)
            if self.mode == "d2h" and self.world > 1:
                self._host = SharedPinned(self.world * (n_c + n_s), self.group)
            self._bufs = (block, cts, share, assembly)
        return self._bufs

    def _agree(self, err):
        """Every rank learns whether ANY rank failed before the exchange, and all raise: a rank that raised alone
        (e.g. the reference's error for an all-zero filter in its slice, conv2DOneFilter :573-580) would leave the
        others waiting in the collective until its time-out."""
        if self.world == 1:
            if err is not None:
                raise err
            return
        dev = "cpu" if dist.get_backend(self.group) == "gloo" or not torch.cuda.is_available() else "cuda"
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if err is not None:
            raise err
        if int(flag.item()):
            raise RuntimeError("ShardedConv: another rank's filter slice failed (its error is raised there)")

    def run(self, img_share0, share_tensor, filters, rand_for):
        """img_share0: the client's ciphertexts (replicated); share_tensor: the server's share [C,H,W] (None when
        the input is not shared); filters [M,C,fh,fw] (this rank reads filters[lo:hi]); rand_for(n_out_local)
        supplies (u, e1, mask).  Returns (out_cts, out_share) on rank 0 -- device tensors in mode "gather", views of
        the shared host buffer in mode "d2h" -- and (None, None) elsewhere.  An error on any rank is raised on every
        rank, before the exchange.  In mode "d2h" the views rank 0 gets stay valid until rank 0 enters its next run()
        (the call begins with a barrier, so no rank overwrites the host buffer before that)."""
        n_local = self.hi - self.lo
        full = self.server.plan(self.meta)
        n_sp, N = full.n_spatial, img_share0.shape[-1]
        block, cts, share, assembly = self._buffers(img_share0, n_sp, full.out_shape, (2, 1, N))
        if self.mode == "d2h" and self.world > 1:
            dist.barrier(self.group)   # rank 0 has consumed the previous result by the time it gets here
        err = None
        if n_local > 0:   # more ranks than filters leaves trailing ranks idle: they contribute their (zero) buffers
            try:
                self.server.conv2DSSFromTensors(img_share0, share_tensor, filters[self.lo:self.hi].contiguous(),
                                                self.local_meta, rand_for(n_local * n_sp),
                                                out=(cts[:n_local * n_sp], share[:n_local]))
            except Exception as e:   # agreed on below
                err = e
        self._agree(err)
        M = self.meta.n_filters
        if self.world == 1:
            return cts[:M * n_sp], share[:M]
        n_c, n_s = cts.numel(), share.numel()

        if self.mode == "gather":
            gather_into(block, None if assembly is None else assembly.view(-1), 0, self.group)   # ONE collective
            if self.rank != 0:
                return None, None
            # rank r's block is [ciphertexts | share]: the two parts of all ranks are brought together here (one small
            # device copy each on rank 0; the collective is what costs)
            c = assembly[:, :n_c].reshape(self.world * self.per * n_sp, 2, 1, N)
            sh = assembly[:, n_c:].reshape(self.world * self.per, *share.shape[1:])
            return c[:M * n_sp], sh[:M]
        host = self._host.t
        h_cts = host[:self.world * n_c].view(self.world * self.per * n_sp, 2, 1, N)
        h_share = host[self.world * n_c:self.world * (n_c + n_s)].view(self.world * self.per, *share.shape[1:])
        h_cts[self.rank * self.per * n_sp:(self.rank + 1) * self.per * n_sp].copy_(cts, non_blocking=True)
        h_share[self.rank * self.per:(self.rank + 1) * self.per].copy_(share, non_blocking=True)
        if block.is_cuda:
            torch.cuda.current_stream().synchronize()
        dist.barrier(self.group)       # every slice has landed
        return (h_cts[:M * n_sp], h_share[:M]) if self.rank == 0 else (None, None)


def sharded_elementwise(fn, x, group=None):
    """Apply a per-element conversion kernel wrapper `fn` to this rank's element range of x and
    gather the pieces on rank 0 (Ring2Field / Field2Ring threads in the reference)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(x.shape[0], world, rank)
    local = fn(x[lo:hi].contiguous()) if hi > lo else x[:0]
    return gather_units(local, x.shape[0], 0, group)


def sharded_bole(bn, enc_x, enc_y, pt_x, pt_y, meta, random, group=None):
    """HomBNSS::bole + addMask (hom_bn_ss.cc:280-362) with the ciphertext index split over the ranks: rank r
    multiplies ciphertexts [lo, hi) (N slots each) and the pieces are gathered on rank 0."""
    import dataclasses
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_ct, N = enc_x.shape[0], enc_x.shape[-1]
    n = meta.vec_shape[0]
    lo, hi = shard_range(n_ct, world, rank)
    if hi > lo:
        e_lo, e_hi = lo * N, min(hi * N, n)
        local = dataclasses.replace(meta, vec_shape=(e_hi - e_lo,))
        cts, share = bn.bole(enc_x[lo:hi].contiguous(), enc_y[lo:hi].contiguous(), pt_x[lo:hi].contiguous(),
                             pt_y[lo:hi].contiguous(), local, random[e_lo:e_hi].contiguous())
        if share.shape[0] < (hi - lo) * N:     # ragged last ciphertext: bring the share slice to whole ciphertexts
            share = torch.cat([share, share.new_zeros((hi - lo) * N - share.shape[0])])
        share = share.view(hi - lo, N)
    else:
        cts = enc_x.new_zeros((0, 2, 1, N))
        share = enc_x.new_zeros((0, N))
    out_cts = gather_units(cts, n_ct, 0, group)
    out_share = gather_units(share, n_ct, 0, group)
    if out_cts is None:
        return None, None
    return out_cts, out_share.reshape(-1)[:n]
