"""Multi-GPU MSM and FFT: one process per GPU; the exchanges run on RCCL inside libark_hip.so (comm_init) or, without
it, on torch.distributed.

The reference itself splits an MSM by base range and sums the chunk results
(ec/src/scalar_mul/variable_base/mod.rs:521-557); across GPUs the same split needs exactly one exchange.  The
reduction operator is elliptic-curve addition, which RCCL does not offer as a reduce op, so the collective is an
all-gather (latency-bound, a few microseconds over xGMI):
  * inside the library (comm_init done, device-resident shards): every rank contributes its per-window PART SUMS
    (a 64-byte header + at most 1024 XYZZ points, ~40 KB) straight from device memory, one kernel adds the copies of
    every part and ONE host tail runs on the sums of the whole job (csrc/capi_comm.hip: ark_hip_msm_sw_device_sharded);
  * on torch.distributed (gloo on CPU, or no library communicator): every rank contributes one finished Projective point
    (3 field elements: 144 B for BLS12-381 G1) and world_size - 1 point additions follow on the host.
No other data-path collective exists: bases and scalars never leave their GPU.
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


def _is_cuda(t):
    return hasattr(t, "is_cuda") and t.is_cuda


def _sharded_in_library(curve, bases_shard, scalars_shard, mont):
    """local MSM + all-gather + sum, all inside libark_hip.so (ark_hip_msm_sw_device_sharded: RCCL)"""
    import ctypes as C
    from ._lib import check, lib
    import torch
    from .msm import _rows
    cid = cv.curve_id(curve)
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    # the library reads raw device memory on its own non-blocking stream: the shards must be dense, and whatever produced
    # them on torch's stream (a slice + .contiguous(), a scalar computation) must have finished
    if not (bases_shard.is_contiguous() and scalars_shard.is_contiguous()):
        raise ValueError("sharded MSM needs contiguous shards (a strided view would be read as dense memory)")
    n = min(_rows(bases_shard, cv.affine_words(cid)), _rows(scalars_shard, cv.SCALAR_WORDS))
    torch.cuda.current_stream().synchronize()
    check(lib().ark_hip_msm_sw_device_sharded(cid, bases_shard.data_ptr(), scalars_shard.data_ptr(), n, mont,
                                              out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw_device_sharded")
    return out


def msm_bigint_sharded(curve, bases_shard, bigints_shard, group=None, local_msm=msm_bigint):
    """MSM over the union of all ranks' shards; each rank passes only its own (base, scalar) range.  With the library's
    RCCL communicator up (comm_init) and device-resident shards the whole call is one C entry point."""
    if library_comm_active() and local_msm is msm_bigint and _is_cuda(bases_shard) and _is_cuda(bigints_shard):
        return _sharded_in_library(curve, bases_shard, bigints_shard, 0)
    return combine_partials(curve, local_msm(curve, bases_shard, bigints_shard), group)


def msm_sharded(curve, bases_shard, scalars_shard, group=None):
    """Same for Fr (Montgomery) scalars: the sharded form of VariableBaseMSM::msm_unchecked."""
    if library_comm_active() and _is_cuda(bases_shard) and _is_cuda(scalars_shard):
        return _sharded_in_library(curve, bases_shard, scalars_shard, 1)
    return combine_partials(curve, msm_unchecked(curve, bases_shard, scalars_shard), group)


# ---- the library's own RCCL communicator ------------------------------------------------------------------------------
# The data-path collectives run INSIDE libark_hip.so (csrc/capi_comm.hip): torch.distributed is only the
# control plane that carries rank 0's RCCL unique id to the other ranks -- exactly what an MPI or Rust host would do with
# its own broadcast.  With the gloo backend (CPU tests, ranks sharing one GPU) there is no RCCL: the same algorithms run
# with torch.distributed as the transport.
_LIB_COMM = {"world": 0}


def comm_init(group=None):
    """Collective: create this rank's RCCL communicator inside the library.  Returns True when the library owns the
    exchange from now on (backend nccl, world > 1), False otherwise."""
    import ctypes as C
    import torch.distributed as dist
    from ._lib import check, lib
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or dist.get_backend(group) != "nccl":
        return False
    if _LIB_COMM["world"] == dist.get_world_size(group):
        return True
    L = lib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    buf = (C.c_ubyte * 128)()
    rc = L.ark_hip_comm_unique_id(buf) if rank == 0 else 0
    # rank 0's failure travels with the broadcast: every rank raises together instead of the others waiting for an id
    box = [(rc, bytes(buf))]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    check(box[0][0], "ark_hip_comm_unique_id (on rank 0)")
    ident = (C.c_ubyte * 128).from_buffer_copy(box[0][1])
    check(L.ark_hip_comm_init(ident, rank, world), "ark_hip_comm_init")
    _LIB_COMM["world"] = world
    return True


def comm_destroy():
    from ._lib import check, lib
    if _LIB_COMM["world"]:
        check(lib().ark_hip_comm_destroy(), "ark_hip_comm_destroy")
        _LIB_COMM["world"] = 0


def library_comm_active():
    return _LIB_COMM["world"] > 1


# ---- radix-2 FFT sharded by coefficient range: ONE exchange -----------------------------------------------------------
# n = 2^k coefficients over G ranks (a power of two <= 16, G^2 | n), m = n/G per rank, sub = m/G.  With i = i1 + G i2 and
# j = j1 m + j2:
#     X[j1 m + j2] = sum_{i1 < G} w_G^(i1 j1) * w_n^(i1 j2) * ( sum_{i2 < m} w_m^(i2 j2) x[i1 + G i2] )
#   forward  in:  x_local[i2] = x[rank + G i2]                      (cyclic by coefficient index)
#            out: y_local[j1 sub + t] = X[j1 m + rank sub + t]      (G rows of the rank's j2 range)
#            = local size-m transform (twiddle w_n^rank fused into its last pass), all-to-all, G-point transform
#   inverse  takes the forward's output layout back to the forward's input layout.
# (ark_hip_fft_sharded_device, include/ark_hip.h.)  One all-to-all of (G-1)/G of the data per transform: 224 MiB per GPU
# at 2^26 over 8 GPUs, ~0.2 ms over 7 xGMI links; the reference's CPU counterpart of the split is the recursion of
# radix2/fft.rs:190-349 cut after log2 G stages.
def shard_input(x_full, rank, world):
    """rank's slice of a full coefficient vector [n, 4] in the forward transform's input layout"""
    return x_full[rank::world]


def unshard_output(y_locals):
    """full evaluation vector [n, 4] from the ranks' forward outputs (list of [m, 4] arrays, rank order)"""
    G = len(y_locals)
    m = y_locals[0].shape[0]
    sub = m // G
    out = np.empty((G * m,) + tuple(y_locals[0].shape[1:]), dtype=y_locals[0].dtype)
    for r, y in enumerate(y_locals):
        for j1 in range(G):
            out[j1 * m + r * sub: j1 * m + (r + 1) * sub] = y[j1 * sub:(j1 + 1) * sub]
    return out


def fft_sharded(field, n, x_local, inverse=False, group=None, offset=None):
    """Forward (or inverse) radix-2 FFT of size n over all ranks' shards; x_local: CUDA int64 tensor [n/G, 4] (Montgomery
    Fr) in the layouts above.  offset: coset offset (4 limbs) or None.  Returns this rank's part of the result (new tensor)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import check, lib
    from .domain import Radix2EvaluationDomain
    fid = cv.field_id(field)
    L = lib()
    dom = Radix2EvaluationDomain.new(fid, n)
    if dom is None or dom.size() != n:
        raise ValueError("n must be a power of two within the field's two-adicity")
    if offset is not None:
        dom = dom.get_coset(offset)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dom.ifft(x_local) if inverse else dom.fft(x_local)
    G = dist.get_world_size(group)
    r = dist.get_rank(group)
    if G & (G - 1) or G > 16 or n % (G * G):
        raise ValueError("world size must be a power of two <= 16 with G^2 dividing n")
    m = n // G
    assert x_local.is_cuda and x_local.numel() == 4 * m
    y = x_local.clone().contiguous()
    torch.cuda.current_stream().synchronize()
    sref = C.byref(dom._s)
    inv = 1 if inverse else 0
    if library_comm_active():
        check(L.ark_hip_fft_sharded_device(fid, sref, y.data_ptr(), inv), "ark_hip_fft_sharded_device")
        check(L.ark_hip_synchronize(), "sync")
        return y

    def exchange(t):  # block q of every rank to rank q, carried by torch.distributed (gloo: through host memory)
        check(L.ark_hip_synchronize(), "sync")
        if dist.get_backend(group) == "nccl":
            out = torch.empty_like(t)
            dist.all_to_all_single(out, t, group=group)
            torch.cuda.current_stream().synchronize()
            return out
        src = t.cpu()
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src, group=group)
        return out.to(t.device)

    if not inverse:
        check(L.ark_hip_fft_shard_local_device(fid, sref, r, G, y.data_ptr(), 0), "shard_local")
        z = exchange(y)
        check(L.ark_hip_fft_shard_cross_device(fid, sref, G, z.data_ptr(), z.data_ptr(), 0), "shard_cross")
    else:
        check(L.ark_hip_fft_shard_cross_device(fid, sref, G, y.data_ptr(), y.data_ptr(), 1), "shard_cross")
        z = exchange(y)
        check(L.ark_hip_fft_shard_local_device(fid, sref, r, G, z.data_ptr(), 1), "shard_local")
    check(L.ark_hip_synchronize(), "sync")
    return z
