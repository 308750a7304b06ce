"""At-size parity for the BASELINE.json configurations and the corner cases round 1 left uncovered
(VERDICT r1 "What's weak" #1): G2 at full size, 2^26 and the 2^23 shard plan, inverse transforms at 2^22 against the
oracle, unreduced scalars, concurrent host threads, short (degree-aware) FFT inputs, and the sharded MSM with the
product's own kernels under more than one rank.  Everything goes through the C ABI."""
import ctypes as C
import os
import socket
import sys
import threading

import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
import hip_lib as H
import pyref as P

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth as S  # noqa: E402

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def sf(cid):
    return O.curve_info(cid)[1]


def rmod(cid):
    return S.R[O.FIELDS[sf(cid)]]


def dlog_case(cname, logn, seed, prepared=False):
    """MSM of P_i = (a + i b)G at 2^logn == k*G, k = sum s_i (a + i b): exact at any size."""
    import torch
    cid = O.CID[cname]
    r = rmod(cid)
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, seed, r)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    if prepared:
        pb = A.PreparedBases(cid, bases)
        got = pb.msm_bigint(d_sc)
        pb.free()
    else:
        got = A.msm_bigint(cid, bases, d_sc)
    k = S.dlog_of_msm(sc, S.A0, S.B0, r)
    # k*G from the ORACLE (independent of the device arithmetic)
    kg = O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k)))
    assert np.array_equal(A.into_affine(cid, got), kg), (cname, logn, prepared)
    del bases, d_sc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("cname,logn", [("BLS12_377_G2", 22), ("BLS12_381_G2", 20)])
def test_msm_g2_at_size_dlog(cname, logn):
    # BASELINE config 5 (BLS12-377 G2, 2^22) and its BLS12-381 sibling: the full-size plan (window split, sort
    # partitioning, level-0 chunking) on the Fp2 path
    dlog_case(cname, logn, 4000 + logn)


@pytest.mark.parametrize("logn", [23, 25, 26, 27])
def test_msm_bls12_381_g1_2_26_and_shard_plan_dlog(logn):
    # BASELINE config 4: the 2^26 total and the 2^23-per-GPU shard plan of its 8-way split; 2^25 and 2^27: the window
    # groups (from 2^25) on either side of the sort's tile-size switch (16384-key tiles at 2^26 only: 2^11 super-buckets
    # fit the scatter kernel's LDS beside them, 2^12 do not -- a 2^27 job once asked for 160 KiB and aborted the queue)
    dlog_case("BLS12_381_G1", logn, 5000 + logn)


def test_two_jobs_in_flight_at_2_22_dlog():
    # the two MSM lanes at a throughput-bound size: two asynchronous jobs over one prepared base set (different scalar
    # vectors), both in flight, each == its own k*G; then a plain synchronous job while a prepared one is still in flight
    import torch
    cname, logn = "BLS12_381_G1", 22
    cid = O.CID[cname]
    r = rmod(cid)
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    pb = A.PreparedBases(cid, bases)
    scs = [S.gen_scalars(n, 9100 + i, r) for i in range(2)]
    dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in scs]
    kgs = [O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(S.dlog_of_msm(x, S.A0, S.B0, r)))) for x in scs]
    j0 = pb.msm_bigint_async(dev[0])
    j1 = pb.msm_bigint_async(dev[1])
    r1, r0 = j1.wait(), j0.wait()
    assert np.array_equal(A.into_affine(cid, r0), kgs[0]) and np.array_equal(A.into_affine(cid, r1), kgs[1])
    j0 = pb.msm_bigint_async(dev[0])
    plain = A.msm_bigint(cid, bases, dev[1])          # enqueued behind / beside the prepared job: second lane
    assert np.array_equal(A.into_affine(cid, plain), kgs[1])
    assert np.array_equal(A.into_affine(cid, j0.wait()), kgs[0])
    pb.free()
    del bases, dev
    torch.cuda.empty_cache()


def test_fft_batch_2_22_vs_oracle_and_single():
    # BASELINE config 3 through the batched entry: three polynomials at 2^22 (three streams), one checked limb for limb
    # against the oracle, all three against the single-transform entry
    import torch
    fname, log_n = "BLS12_381_FR", 22
    fid = O.FID[fname]
    n = 1 << log_n
    d = A.Radix2EvaluationDomain.new(fname, n)
    xs = [O.gen_scalars(fid, 3300 + i, n, montgomery=True) for i in range(3)]
    dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in xs]
    single = [d.fft(t) for t in dev]                   # copies
    d.fft_batch_in_place(dev)
    for i in range(3):
        assert torch.equal(dev[i].view(torch.int64).reshape(-1), single[i].view(torch.int64).reshape(-1)), i
    threads = min(os.cpu_count() or 8, 64)
    assert np.array_equal(dev[1].cpu().numpy().view(np.uint64).reshape(-1), O.fft(fid, xs[1], log_n, None, False, threads))
    del dev, single
    torch.cuda.empty_cache()


def test_msm_g2_2_18_vs_oracle_wnaf():
    import torch
    cname, logn = "BLS12_377_G2", 18
    cid = O.CID[cname]
    n = 1 << logn
    seed = O.gen_bases(cid, A4, B4, 1 << 10)
    d = H.gpu_extend_bases(cid, seed, n, lambda m: _delta(cid, m))
    bases = d.cpu().numpy().view(np.uint64).reshape(n, -1)
    scalars = O.gen_scalars(sf(cid), 1818, n)
    got = A.msm_bigint(cid, d, torch.from_numpy(scalars.view(np.int64)).cuda())
    exp = O.msm(cid, bases, scalars, O.WNAF, os.cpu_count() or 8)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp))


def _delta(cid, m):
    r = rmod(cid)
    k = (m * P.from_limbs(B4)) % r
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), np.array(P.to_limbs(k, 4), dtype=np.uint64)))


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_msm_unreduced_scalars(cname):
    # msm_bigint takes any BigInt<4>: values in [r, 2^bits) give the reference's result (make_digits, mod.rs:754-794,
    # is exact below 2^bits); values with higher bits are rejected with ARK_HIP_ERR_SCALAR_RANGE
    cid = O.CID[cname]
    r = rmod(cid)
    bits = r.bit_length()
    n = 96
    bases = O.gen_bases(cid, A4, B4, n)
    rng = np.random.default_rng(77)
    vals = [r, r + 1, (1 << bits) - 1, r + (1 << 64), (1 << bits) - (1 << 70), r - 1, 0, 1]
    vals += [r + int.from_bytes(rng.bytes(40), "little") % ((1 << bits) - r) for _ in range(n - len(vals))]
    scalars = np.array([P.to_limbs(v, 4) for v in vals], dtype=np.uint64)
    got = A.msm_bigint(cid, bases, scalars)
    exp = O.msm(cid, bases, scalars, O.SIGNED, 4)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp))
    # and the result is the reduced scalars' MSM
    red = np.array([P.to_limbs(v % r, 4) for v in vals], dtype=np.uint64)
    assert np.array_equal(A.into_affine(cid, got), A.into_affine(cid, A.msm_bigint(cid, bases, red)))
    # a scalar >= 2^bits: error code, no result
    for bad in ((1 << bits), (1 << 256) - 1):
        s2 = scalars.copy()
        s2[5] = P.to_limbs(bad, 4)
        with pytest.raises(A.ArkHipError) as ei:
            A.msm_bigint(cid, bases, s2)
        assert ei.value.code == -4
    # the library is still usable afterwards
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, bases, scalars)), O.to_affine(cid, exp))


def test_ifft_and_coset_ifft_2_22_vs_oracle():
    # BASELINE config 3, inverse direction: limb for limb against the oracle (round 1 only had the round trip)
    import torch
    fname, log_n = "BLS12_381_FR", 22
    fid = O.FID[fname]
    n = 1 << log_n
    x = O.gen_scalars(fid, 2244, n, montgomery=True)
    d = A.Radix2EvaluationDomain.new(fname, n)
    dx = torch.from_numpy(x.view(np.int64)).cuda()
    threads = min(os.cpu_count() or 8, 64)
    got = d.ifft(dx).cpu().numpy().view(np.uint64).reshape(-1)
    assert np.array_equal(got, O.fft(fid, x, log_n, None, True, threads))
    gen = O.field_const(fid, 3)
    dc = d.get_coset(gen)
    got = dc.ifft(dx).cpu().numpy().view(np.uint64).reshape(-1)
    assert np.array_equal(got, O.fft(fid, x, log_n, gen, True, threads))
    got = dc.fft(dx).cpu().numpy().view(np.uint64).reshape(-1)
    assert np.array_equal(got, O.fft(fid, x, log_n, gen, False, threads))


@pytest.mark.parametrize("fname", ["BLS12_381_FR", "BN254_FR", "BLS12_377_FR"])
def test_fft_degree_aware_short_inputs(fname):
    # fft_in_place with coeffs.len() * 4 <= size takes the degree-aware path (radix2/mod.rs:141, fft.rs:29-71): same
    # output as the zero-padded transform.  Power-of-two and ragged lengths, subgroup and coset, host and device.
    import torch
    fid = O.FID[fname]
    gen = O.field_const(fid, 3)
    for log_n, ln in [(4, 4), (6, 9), (10, 256), (11, 512), (12, 1), (12, 3), (12, 1000), (13, 2048), (16, 777),
                      (16, 1 << 14), (20, 1 << 18), (20, (1 << 17) + 5), (22, 1 << 20)]:
        n = 1 << log_n
        x = O.gen_scalars(fid, 100 * log_n + ln % 97, ln, montgomery=True)
        full = np.zeros((n, 4), dtype=np.uint64)
        full[:ln] = x
        d = A.Radix2EvaluationDomain.new(fname, n)
        exp = O.fft(fid, full, log_n, None, False, 8)
        assert np.array_equal(d.fft(x).reshape(-1), exp), (fname, log_n, ln, "host")
        y = d.fft(torch.from_numpy(x.view(np.int64)).cuda())
        assert np.array_equal(y.cpu().numpy().view(np.uint64).reshape(-1), exp), (fname, log_n, ln, "device")
        if log_n <= 16:
            dc = d.get_coset(gen)
            assert np.array_equal(dc.fft(x).reshape(-1), O.fft(fid, full, log_n, gen, False, 8)), (fname, log_n, ln, "coset")


@pytest.mark.parametrize("fname", ["BLS12_381_FR", "BN254_FR", "BLS12_377_FR"])
def test_fft_batch_matches_single_transforms(fname):
    # ark_hip_fft_batch_in_place_device: several transforms over one domain, up to three in flight on separate streams --
    # each result limb for limb what the oracle gives for that polynomial (forward / inverse, subgroup / coset), batch
    # sizes around the lane count, sizes from the single-workgroup kernel to three passes
    import torch
    fid = O.FID[fname]
    gen = O.field_const(fid, 3)
    for log_n, count in [(3, 5), (10, 4), (11, 1), (13, 2), (16, 7), (20, 3)]:
        n = 1 << log_n
        d = A.Radix2EvaluationDomain.new(fname, n)
        for dom, off in ((d, None), (d.get_coset(gen), gen)):
            xs = [O.gen_scalars(fid, 7000 + 31 * log_n + i, n, montgomery=True) for i in range(count)]
            for inverse in (False, True):
                dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in xs]
                dom.fft_batch_in_place(dev, inverse=inverse)
                for i in range(count):
                    exp = O.fft(fid, xs[i], log_n, off, inverse, 8)
                    assert np.array_equal(dev[i].cpu().numpy().view(np.uint64).reshape(-1), exp), (fname, log_n, count, i, inverse)
    d = A.Radix2EvaluationDomain.new(fname, 1 << 8)
    assert d.fft_batch_in_place([]) == []


def test_concurrent_host_threads_msm_and_fft():
    # SURVEY 8(b) threading row: trait functions may be called from many rayon threads at once.  Four host threads
    # hammer the host-pointer MSM (shared staging buffers), the device MSM and the FFT; every result must be right.
    cid = O.CID["BLS12_381_G1"]
    fid = O.FID["BLS12_381_FR"]
    n = 3000
    bases = O.gen_bases(cid, A4, B4, n)
    cases = []
    for t in range(4):
        sc = O.gen_scalars(sf(cid), 900 + t, n - 100 * t)
        cases.append((bases[: n - 100 * t], sc, O.to_affine(cid, O.msm(cid, bases[: n - 100 * t], sc, O.SIGNED, 2))))
    x = O.gen_scalars(fid, 5, 1 << 12, montgomery=True)
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", 1 << 12)
    fexp = O.fft(fid, x, 12, None, False, 2)
    errors = []

    def worker(t):
        try:
            for it in range(12):
                b, s, e = cases[(t + it) % 4]
                if not np.array_equal(A.into_affine(cid, A.msm_bigint(cid, b, s)), e):
                    errors.append(("msm", t, it))
                if not np.array_equal(dom.fft(x).reshape(-1), fexp):
                    errors.append(("fft", t, it))
        except Exception as ex:  # noqa: BLE001
            errors.append(("exc", t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


# ---- sharded MSM with the product's kernels under 2 and 4 ranks (gloo, ranks share the box's one GPU) -----------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _msm_worker(rank, world, port, cname, n, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from algebra_amd import dist as D
    cid = O.CID[cname]
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 5, n)
    lo, hi = D.shard_bounds(n, rank, world)
    d_b = torch.from_numpy(bases[lo:hi].view(np.int64)).cuda()
    d_s = torch.from_numpy(scalars[lo:hi].view(np.int64)).cuda()
    total = D.msm_bigint_sharded(cid, d_b, d_s)                      # local MSM = the HIP pipeline
    full = O.msm(cid, bases, scalars, O.SIGNED, 2)
    ok = np.array_equal(A.into_affine(cid, total), O.to_affine(cid, full))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cname,n", [(2, "BLS12_381_G1", 20011), (4, "BLS12_381_G1", 4099), (2, "BLS12_377_G2", 1500)])
def test_sharded_msm_product_kernels_multi_rank(world, cname, n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_msm_worker, args=(r, world, port, cname, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
