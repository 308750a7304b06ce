"""Multi-GPU MSM: one process per GPU, base-range sharding, RCCL all-gather of the per-rank partials.

The reference itself splits an MSM by base range and sums the chunk results
(ec/src/scalar_mul/variable_base/mod.rs:521-557); across GPUs the same split needs exactly one exchange:
every rank contributes one Projective point (3 field elements: 144 B for BLS12-381 G1).  The reduction
operator is elliptic-curve addition, which RCCL does not offer as a reduce op, so the collective is an
all-gather (latency-bound, a few microseconds over xGMI) followed by world_size-1 point additions on
the host.  No other data-path collective exists: bases and scalars never leave their GPU.
"""
import numpy as np

from . import curves as cv
from .msm import msm_bigint, msm_unchecked, sum_projective


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's contiguous base range; the first n % world ranks take one extra element."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def combine_partials(curve, partial, group=None):
    """All-gather every rank's partial Projective and return their sum (identical on all ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.ascontiguousarray(partial, dtype=np.uint64)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64)).to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    allp = torch.stack(bufs).cpu().numpy().view(np.uint64)
    return sum_projective(curve, allp)


def msm_bigint_sharded(curve, bases_shard, bigints_shard, group=None, local_msm=msm_bigint):
    """MSM over the union of all ranks' shards; each rank passes only its own (base, scalar) range."""
    return combine_partials(curve, local_msm(curve, bases_shard, bigints_shard), group)


def msm_sharded(curve, bases_shard, scalars_shard, group=None):
    """Same for Fr (Montgomery) scalars: the sharded form of VariableBaseMSM::msm_unchecked."""
    return combine_partials(curve, msm_unchecked(curve, bases_shard, scalars_shard), group)


# ---- radix-2 FFT sharded by coefficient range ---------------------------------------------------------------
# n = 2^k coefficients, G = world_size ranks (a power of two <= 16), rank r holds x[r*m .. (r+1)*m), m = n/G, and
# receives X[r*m .. (r+1)*m) (natural order, block layout in and out).  With i = i1*m + i2 and j = j1 + G*j2:
#     X[j1 + G j2] = sum_{i2 < m} w_m^(i2 j2) * w_n^(i2 j1) * ( sum_{i1 < G} w_G^(i1 j1) x[i1 m + i2] )
#   exchange 1  all-to-all: rank q collects, for its i2 range, the values of every i1
#   local       G-point transform over i1 (ark_hip_fft_axis_device)
#   exchange 2  all-to-all: rank j1 collects u[j1][all i2]
#   local       size-m FFT with the coset pre-scaling h = w_n^j1 fused into its first pass (the single-GPU kernel)
#   exchange 3  all-to-all back to block layout (a pipeline that multiplies pointwise and transforms back can skip it)
# Three all-to-alls of (G-1)/G of the data each; over xGMI (7 links x ~153 GB/s per GPU) a 2^26-point transform moves
# 3 x 224 MiB per GPU.  The butterflies themselves never cross GPUs except in the G-point stage.
def _all_to_all(t, group, backend):
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=group)
        return out
    src = t.cpu()
    out = torch.empty_like(src)
    dist.all_to_all_single(out, src, group=group)
    return out.to(t.device)


def fft_sharded(field, n, x_local, inverse=False, group=None):
    """Forward (or inverse) radix-2 FFT of size n over all ranks' shards; x_local: CUDA int64 tensor [n/G, 4]
    (Montgomery Fr).  Returns this rank's block of the result (new tensor)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import check, lib
    from .domain import Radix2EvaluationDomain
    fid = cv.field_id(field)
    L = lib()
    dom_n = Radix2EvaluationDomain.new(fid, n)
    if dom_n is None or dom_n.size() != n:
        raise ValueError("n must be a power of two within the field's two-adicity")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dom_n.ifft(x_local) if inverse else dom_n.fft(x_local)
    G = dist.get_world_size(group)
    r = dist.get_rank(group)
    backend = dist.get_backend(group)
    if G & (G - 1) or G > 16 or n % (G * G):
        raise ValueError("world size must be a power of two <= 16 with G^2 dividing n")
    m = n // G
    sub = m // G
    assert x_local.is_cuda and x_local.numel() == 4 * m
    w = dom_n.group_gen_inv() if inverse else dom_n.group_gen()

    def fr_pow(base, e):
        out = np.zeros(4, dtype=np.uint64)
        b = np.ascontiguousarray(base, dtype=np.uint64)
        check(L.ark_hip_fr_pow(fid, b.ctypes.data_as(C.c_void_p), e, out.ctypes.data_as(C.c_void_p)), "fr_pow")
        return out

    # exchange 1: sub-block q of my block (i2 in [q*sub, (q+1)*sub)) goes to rank q
    recv = _all_to_all(x_local.reshape(G * sub, 4).contiguous(), group, backend)        # [i1][t]
    torch.cuda.current_stream().synchronize()
    root_g = fr_pow(w, n // G)                                                          # primitive G-th root
    check(L.ark_hip_fft_axis_device(fid, recv.data_ptr(), G, sub, root_g.ctypes.data_as(C.c_void_p)), "fft_axis")
    # exchange 2: row j1 goes to rank j1; what arrives is u[r][q*sub + t], i.e. i2 order
    col = _all_to_all(recv, group, backend)
    torch.cuda.current_stream().synchronize()
    # local size-m transform with the pre-scaling (w_n^r)^i2 fused in: a forward "coset FFT" whose generator is
    # w_m (or w_m^-1) and whose offset is w_n^r (or w_n^-r)
    dom_m = Radix2EvaluationDomain.new(fid, m)
    s = dom_m._s
    if inverse:
        for k in range(4):
            s.group_gen[k] = s.group_gen_inv[k]
    h = fr_pow(w, r)
    for k in range(4):
        s.offset[k] = int(h[k])
    check(L.ark_hip_fft_in_place_device(fid, C.byref(s), col.data_ptr()), "local fft")
    if inverse:  # the 1/n of the inverse transform
        sinv = torch.from_numpy(np.tile(dom_n.size_inv(), (m, 1)).view(np.int64)).cuda()
        torch.cuda.current_stream().synchronize()
        check(L.ark_hip_fr_mul_device(fid, col.data_ptr(), sinv.data_ptr(), col.data_ptr(), m), "scale")
    check(L.ark_hip_synchronize(), "sync")
    # exchange 3: I hold X[r + G*j2]; block q of j2 goes to rank q, which interleaves the G residues
    z = _all_to_all(col, group, backend)                                                # [j1][t] = X[j1 + G (r*sub + t)]
    return z.reshape(G, sub, 4).permute(1, 0, 2).contiguous().reshape(m, 4)
