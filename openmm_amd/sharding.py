"""Multi-GPU sharding of the path's independent units.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  Units never interact: output filters of a HomConv layer
(hom_conv2d_ss.cc:679-690), row blocks of HomFC (hom_fc_ss.cc:443-457), ciphertext indices of
BN/BOLE (hom_bn_ss.cc:261-271, :302-309), element ranges of the conversions
(library_fixed_uniform_cheetah.cpp:494-511).  Each rank computes a contiguous range of units on
replicated inputs with no data-path collective; the only exchange is the gather of the per-rank
results onto the rank that serialises them for the network.
"""
import torch
import torch.distributed as dist


def shard_range(n_units, world, rank):
    """Contiguous range of rank `rank`, split like the reference's LaunchWorks
    (hom_conv2d_ss.cc:41-45): ceil(n/world) units per worker, trailing workers may be empty."""
    per = (n_units + world - 1) // world
    start = min(rank * per, n_units)
    return start, min(start + per, n_units)


def gather_units(local, n_units, dst=0, group=None):
    """Gather per-rank unit slices (dim 0) onto rank `dst`; returns the full tensor there, None
    elsewhere.  Ragged / empty last shards are padded to the common size for the collective."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return local
    per = (n_units + world - 1) // world
    shape = (per,) + tuple(local.shape[1:])
    padded = torch.zeros(shape, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(bufs)[:n_units]


class ShardedConv:
    """HomConv2DSS::conv2DSS with the output filters split over the ranks."""

    def __init__(self, server, meta, group=None):
        import dataclasses
        self.server, self.meta, self.group = server, meta, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lo, self.hi = shard_range(meta.n_filters, self.world, self.rank)
        self.local_meta = dataclasses.replace(meta, n_filters=max(self.hi - self.lo, 1))

    def run(self, img_share0, img_share1, filters, rand_for):
        """filters: the rank's own slice is filters[lo:hi] ([M,C,fh,fw], full tensor replicated or
        only the slice valid); rand_for(n_out_local) supplies (u, e1, mask).  Returns
        (out_cts, out_share) gathered on rank 0 (None elsewhere)."""
        n_local = self.hi - self.lo
        full = self.server.plan(self.meta)
        n_sp = full.n_spatial
        if n_local > 0:
            cts, share = self.server.conv2DSSFromWeights(img_share0, img_share1,
                                                         filters[self.lo:self.hi].contiguous(), self.local_meta,
                                                         rand_for(n_local * n_sp))
            cts = cts.reshape(n_local, n_sp, *cts.shape[1:])
        else:  # more ranks than filters: nothing to do, contribute an empty slice
            ref = img_share0
            cts = torch.zeros((0, n_sp, 2, 1, ref.shape[-1]), dtype=ref.dtype, device=ref.device)
            share = torch.zeros((0,) + tuple(full.out_shape[1:]), dtype=ref.dtype, device=ref.device)
        out_cts = gather_units(cts, self.meta.n_filters, 0, self.group)
        out_share = gather_units(share, self.meta.n_filters, 0, self.group)
        if out_cts is not None:
            out_cts = out_cts.reshape(-1, *out_cts.shape[2:])
        return out_cts, out_share


def sharded_elementwise(fn, x, group=None):
    """Apply a per-element conversion kernel wrapper `fn` to this rank's element range of x and
    gather the pieces on rank 0 (Ring2Field / Field2Ring threads in the reference)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(x.shape[0], world, rank)
    local = fn(x[lo:hi].contiguous()) if hi > lo else x[:0]
    return gather_units(local, x.shape[0], 0, group)
