//! MSM entry points for short-Weierstrass curves.
//!
//! Reference surface replaced (ec/src/scalar_mul/variable_base/mod.rs:59-150, short_weierstrass/mod.rs:112-119,
//! group.rs:650-657):  `msm` -> [`sw_msm`], `msm_bigint` (and through it `msm_unchecked`, `msm_chunks`,
//! `ChunkedPippenger`, `HashMapPippenger`) -> [`sw_msm_bigint`], `msm_u1` / `msm_u8` / `msm_u16` / `msm_u32` / `msm_u64`
//! -> [`sw_msm_small`].
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::scalar_mul::variable_base::VariableBaseMSM;
use ark_ff::PrimeField;
use ark_hip_sys as sys;
use ark_std::vec::Vec;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;
use core::mem::{offset_of, size_of, MaybeUninit};

type BigIntOf<P> = <<P as ark_ec::CurveConfig>::ScalarField as PrimeField>::BigInt;

/// Marker of the curve configs libark_hip.so serves: `CURVE` is the library's id of THIS curve.
///
/// Every typed entry point of this module is bounded by it, so a config that has not declared itself served does not
/// compile against them, and the id a call passes is checked against the declaration (rounds 3-5 trusted the integer and, for
/// the transform over group elements, recognised `Projective<P>` by `core::any::type_name` strings).  Implemented by the
/// patched curve crates behind their `hip` feature (patches/0002: one `unsafe impl` per config) and by [`hip_sw_config!`]
/// for the wrapper configs of unmodified arkworks.
///
/// # Safety
/// The implementor asserts that `CURVE` names this very curve in include/ark_hip.h -- the same base field, equation,
/// scalar field and prime-order subgroup -- and that the config's `Affine` / `Projective` use `ZeroFlag = ()`: the library
/// computes in the arithmetic of the curve the id names and reads the points as raw limbs ([`layout_ok`] still checks sizes
/// and offsets at run time).
pub unsafe trait HipServed: SWCurveConfig {
    const CURVE: c_int;
    /// the library's id of `Self::ScalarField`
    const SCALAR_FIELD: c_int = match Self::CURVE {
        sys::BN254_G1 => sys::BN254_FR,
        sys::BLS12_381_G1 | sys::BLS12_381_G2 => sys::BLS12_381_FR,
        _ => sys::BLS12_377_FR,
    };
}

/// Makes `Projective<P>` known to the transform over group elements (`fft_in_place::<Projective<P>>` through ark-poly's
/// hook, `ark_hip_sys::radix2_fft_in_place`): ark-poly sits below ark-ec and cannot name the type, so the typed layer
/// registers its `TypeId`.  Every entry point of this module does it on the way in (a read-locked scan of at most five
/// entries); call it directly when a program transforms points before its first MSM.
pub fn serve_group_coefficients<P: HipServed>() {
    sys::register_group_type::<Projective<P>>(P::CURVE, P::SCALAR_FIELD);
}

/// the id the caller passed against the one the config declares (the declaration wins)
#[inline]
fn served_id<P: HipServed>(curve: c_int) -> c_int {
    debug_assert_eq!(curve, P::CURVE, "ark-hip: curve id passed does not match the config's HipServed::CURVE");
    serve_group_coefficients::<P>();
    P::CURVE
}

/// u64 words of one base-field element for a library curve id
pub const fn fe_words(curve: c_int) -> usize {
    match curve {
        0 => 4,
        1 | 2 => 6,
        _ => 12,
    }
}

/// Layout guard.  arkworks types are not `#[repr(C)]`; in practice `Fp(BigInt([u64; N]), PhantomData)`,
/// `Affine { x, y, infinity: () }` and `Projective { x, y, z }` are contiguous in declaration order, which is what the
/// C ABI assumes.  A curve whose `ZeroFlag` is `bool` (an extra byte in `Affine`) or any other surprise fails the check:
/// the call then takes the CPU path and says so once -- never silently.
///
/// Sizes alone would let a reordering of equally sized members through (rustc is free to reorder the fields of a
/// `repr(Rust)` struct), so the OFFSETS of the public coordinates are checked as well (SURVEY 8b "Memory layout"):
/// `x` at 0, `y` one field element in, `z` two (affine.rs:30-37, group.rs:34-41; `core::mem::offset_of!`, Rust 1.77).
/// `Affine::infinity` is `pub(super)`: it cannot be named here, and with `x`, `y` pinned and the total size equal to
/// two field elements it has no bytes left to occupy (`ZeroFlag = ()`).
fn layout_ok<P: SWCurveConfig, S>(curve: c_int) -> bool {
    let fe = fe_words(curve) * 8;
    let ok = size_of::<Affine<P>>() == 2 * fe
        && size_of::<Projective<P>>() == 3 * fe
        && size_of::<P::BaseField>() == fe
        && offset_of!(Affine<P>, x) == 0
        && offset_of!(Affine<P>, y) == fe
        && offset_of!(Projective<P>, x) == 0
        && offset_of!(Projective<P>, y) == fe
        && offset_of!(Projective<P>, z) == 2 * fe
        && size_of::<S>() == 32;
    if !ok {
        #[cfg(feature = "std")]
        {
            static WARNED: std::sync::Once = std::sync::Once::new();
            WARNED.call_once(|| {
                std::eprintln!(
                    "ark-hip: in-memory layout of curve id {curve} does not match libark_hip.so \
                     (Affine {} B, Projective {} B, scalar {} B): MSM stays on the CPU",
                    size_of::<Affine<P>>(),
                    size_of::<Projective<P>>(),
                    size_of::<S>()
                );
            });
        }
        debug_assert!(ok, "ark-hip layout mismatch");
    }
    ok
}

/// One device MSM over host slices; `None` on layout mismatch or device error (the caller falls back to the CPU).
fn device_msm<P: SWCurveConfig, S>(curve: c_int, bases: &[Affine<P>], scalars: &[S], montgomery: bool) -> Option<Projective<P>> {
    if !layout_ok::<P, S>(curve) {
        return None;
    }
    let n = bases.len().min(scalars.len());
    let mut out = MaybeUninit::<Projective<P>>::uninit();
    let rc = unsafe {
        sys::ark_hip_msm_sw(curve, bases.as_ptr() as *const u64, scalars.as_ptr() as *const u64, n,
                            montgomery as c_int, out.as_mut_ptr() as *mut u64)
    };
    (rc == 0).then(|| unsafe { out.assume_init() })
}

/// CPU fallback for canonical scalars.
fn cpu_msm_bigint<P: SWCurveConfig>(bases: &[Affine<P>], bigints: &[BigIntOf<P>]) -> Projective<P> {
    #[cfg(feature = "ec-hook")]
    {
        ark_ec::scalar_mul::variable_base::msm_bigint_default(bases, bigints) // patches/0001
    }
    #[cfg(not(feature = "ec-hook"))]
    {
        <Projective<P> as VariableBaseMSM>::msm_bigint(bases, bigints) // trait default: msm_signed on the CPU
    }
}

/// `SWCurveConfig::msm` (short_weierstrass/mod.rs:112-119): `Err(min_len)` when the lengths differ, else the sum
/// on the GPU -- the `into_bigint` pass of variable_base/mod.rs:60-62 included (scalars cross as Montgomery residues).
pub fn sw_msm<P: HipServed>(curve: c_int, bases: &[Affine<P>], scalars: &[P::ScalarField]) -> Result<Projective<P>, usize> {
    let curve = served_id::<P>(curve);
    if bases.len() != scalars.len() {
        return Err(bases.len().min(scalars.len()));
    }
    Ok(device_msm::<P, P::ScalarField>(curve, bases, scalars, true).unwrap_or_else(|| {
        let bigints = scalars.iter().map(|s| s.into_bigint()).collect::<Vec<_>>();
        cpu_msm_bigint::<P>(bases, &bigints)
    }))
}

/// `VariableBaseMSM::msm_bigint` (variable_base/mod.rs:80-85): truncates to the shorter input like the reference.
pub fn sw_msm_bigint<P: HipServed>(curve: c_int, bases: &[Affine<P>], bigints: &[BigIntOf<P>]) -> Projective<P> {
    let curve = served_id::<P>(curve);
    device_msm::<P, BigIntOf<P>>(curve, bases, bigints, false).unwrap_or_else(|| cpu_msm_bigint::<P>(bases, bigints))
}

/// `VariableBaseMSM::msm_u1` / `msm_u8` / `msm_u16` / `msm_u32` / `msm_u64` (variable_base/mod.rs:87-117) behind the
/// `SWCurveConfig::msm_small` hook of patches/0001: the scalars cross the boundary as they are (1 / 2 / 4 / 8 bytes each,
/// `bool` = one byte) and the device builds only the windows their bits reach; truncates to the shorter input like the
/// reference's `preamble` (:349-369).  Any device or layout problem falls back to the reference's CPU bodies.
#[cfg(feature = "ec-hook")]
pub fn sw_msm_small<P: HipServed>(
    curve: c_int,
    bases: &[Affine<P>],
    scalars: ark_ec::scalar_mul::variable_base::SmallScalars<'_>,
) -> Projective<P> {
    let curve = served_id::<P>(curve);
    use ark_ec::scalar_mul::variable_base::SmallScalars as S;
    let (ptr, len, bytes, bits): (*const c_void, usize, c_int, c_int) = match scalars {
        S::U1(s) => (s.as_ptr() as *const c_void, s.len(), 1, 1), // bool: guaranteed 0x00 / 0x01, one byte
        S::U8(s) => (s.as_ptr() as *const c_void, s.len(), 1, 0),
        S::U16(s) => (s.as_ptr() as *const c_void, s.len(), 2, 0),
        S::U32(s) => (s.as_ptr() as *const c_void, s.len(), 4, 0),
        S::U64(s) => (s.as_ptr() as *const c_void, s.len(), 8, 0),
    };
    let n = bases.len().min(len);
    if layout_ok::<P, BigIntOf<P>>(curve) {
        let mut out = MaybeUninit::<Projective<P>>::uninit();
        let rc = unsafe {
            sys::ark_hip_msm_sw_small(curve, bases.as_ptr() as *const u64, ptr, n, bytes, bits, out.as_mut_ptr() as *mut u64)
        };
        if rc == 0 {
            return unsafe { out.assume_init() };
        }
    }
    ark_ec::scalar_mul::variable_base::msm_small_default(bases, scalars)
}

/// `VariableBaseMSM::msm_chunks` (variable_base/mod.rs:119-150) over slices: streams aligned at their end, steps of
/// 2^20 pairs, the next step's upload overlapping the current step's kernels.  `None` on device error.
pub fn sw_msm_chunks<P: HipServed>(curve: c_int, bases: &[Affine<P>], scalars: &[P::ScalarField]) -> Option<Projective<P>> {
    let curve = served_id::<P>(curve);
    assert!(scalars.len() <= bases.len());
    if !layout_ok::<P, P::ScalarField>(curve) {
        return None;
    }
    let mut out = MaybeUninit::<Projective<P>>::uninit();
    let rc = unsafe {
        sys::ark_hip_msm_sw_chunks(curve, bases.as_ptr() as *const u64, bases.len(), scalars.as_ptr() as *const u64,
                                   scalars.len(), 0, out.as_mut_ptr() as *mut u64)
    };
    (rc == 0).then(|| unsafe { out.assume_init() })
}

/// A base slice pinned on the GPU for the guard's lifetime (`ark_hip_msm_bases_pin` / `_unpin`).
///
/// `sw_msm` / `sw_msm_bigint` / `sw_msm_small` are functions of their two slices.  By default the library keeps a device
/// copy of a base slice it has seen and re-validates it on every call with a hash of the slice's full content (one host
/// pass per call, see [`base_cache_config`]); a prover that calls `G::msm(&srs, ..)` again and again can spare even that
/// pass by pinning the SRS:
///
/// ```ignore
/// let _resident = ResidentBases::pin(ark_hip::BLS12_381_G1, &srs)?;   // uploads srs; borrows it until dropped
/// for w in witnesses { let c = G1Projective::msm(&srs, &w)?; }       // scalars only over PCIe
/// ```
///
/// The guard holds the SHARED BORROW of the slice, so the compiler rejects any `&mut` access to the bases while a device
/// copy of them exists -- the resident copy cannot go stale, which is exactly the guarantee the reference's
/// `bases: &[Affine]` gives for the duration of one call (variable_base/mod.rs:59-85), stretched over the guard's life.
/// Sub-slices (`&srs[..n]`, the steps of `msm_chunks`) hit the resident copy too.
pub struct ResidentBases<'a, P: SWCurveConfig> {
    curve: c_int,
    device: c_int,
    bases: &'a [Affine<P>],
}
impl<'a, P: HipServed> ResidentBases<'a, P> {
    /// `None` if the layout check fails, no device is present or the copy does not fit (callers then simply run unpinned).
    /// The pin lives in the context of the device that is current NOW; the guard remembers it and unpins there.
    pub fn pin(curve: c_int, bases: &'a [Affine<P>]) -> Option<Self> {
        let curve = served_id::<P>(curve);
        if bases.is_empty() || !layout_ok::<P, P::ScalarField>(curve) {
            return None;
        }
        let device = unsafe { sys::ark_hip_get_device() };
        let rc = unsafe { sys::ark_hip_msm_bases_pin(curve, bases.as_ptr() as *const u64, bases.len()) };
        (rc == 0).then_some(Self { curve, device, bases })
    }
}
impl<'a, P: SWCurveConfig> ResidentBases<'a, P> {
    pub fn bases(&self) -> &'a [Affine<P>] {
        self.bases
    }
}
impl<'a, P: SWCurveConfig> Drop for ResidentBases<'a, P> {
    fn drop(&mut self) {
        // unpin on the device that holds the pin (another device may be current by now: its context does not know the
        // range, the unpin would fail and the stale range would stay registered -- ADVICE r4), then restore
        unsafe {
            let cur = sys::ark_hip_get_device();
            if self.device >= 0 && cur != self.device {
                sys::ark_hip_set_device(self.device);
            }
            let rc = sys::ark_hip_msm_bases_unpin(self.curve, self.bases.as_ptr() as *const u64, self.bases.len());
            debug_assert_eq!(rc, 0, "ark-hip: unpin failed; the pinned range may still be registered");
            if self.device >= 0 && cur != self.device && cur >= 0 {
                sys::ark_hip_set_device(cur);
            }
        }
    }
}

/// The verified cache behind `sw_msm` / `sw_msm_bigint` (include/ark_hip.h, `ark_hip_msm_cache_*`; on by default with
/// 16 GiB -- at most a quarter of the device memory --, `ARK_HIP_BASE_CACHE_MB` overrides): device copies keyed by (curve, address, length),
/// validated on every call by a hash of the slice's FULL content computed on host threads while the device works from the
/// copy; a changed slice is refreshed and the MSM rerun, so a result never reflects stale bases.
/// `budget_bytes = Some(0)` turns it off (every call then streams its bases over PCIe).
pub fn base_cache_config(budget_bytes: Option<u64>, auto_prepare_after: Option<u32>) -> bool {
    let b = budget_bytes.map(|v| v as core::ffi::c_longlong).unwrap_or(-1);
    let a = auto_prepare_after.map(|v| v as c_int).unwrap_or(-1);
    unsafe { sys::ark_hip_msm_cache_config(b, a) == 0 }
}
pub fn base_cache_clear() -> bool {
    unsafe { sys::ark_hip_msm_cache_clear() == 0 }
}
/// `[cached sets, device bytes, hits, misses, refreshed, evicted, pinned sets, pinned hits]`
pub fn base_cache_stats() -> Option<[u64; 8]> {
    let mut out = [0u64; 8];
    (unsafe { sys::ark_hip_msm_cache_stats(out.as_mut_ptr()) } == 0).then_some(out)
}

/// One MSM over `n_gpus` GPUs of this node, driven from this process (base-range shards, variable_base/mod.rs:521-557).
pub fn msm_multi<P: HipServed>(curve: c_int, n_gpus: usize, bases: &[Affine<P>], bigints: &[BigIntOf<P>]) -> Option<Projective<P>> {
    let curve = served_id::<P>(curve);
    if !layout_ok::<P, BigIntOf<P>>(curve) {
        return None;
    }
    let n = bases.len().min(bigints.len());
    let mut out = MaybeUninit::<Projective<P>>::uninit();
    let rc = unsafe {
        sys::ark_hip_msm_sw_multi(curve, n_gpus as c_int, bases.as_ptr() as *const u64, bigints.as_ptr() as *const u64, n, 0,
                                  out.as_mut_ptr() as *mut u64)
    };
    (rc == 0).then(|| unsafe { out.assume_init() })
}

/// One process per GPU: the library's own RCCL communicator.  Rank 0 creates the id and the host ships it to the other
/// ranks by whatever it already has (MPI broadcast, a TCP socket); every rank then calls [`comm_init`] on its device
/// (collective).  From then on [`PreparedBases::msm_bigint_sharded`] and the `*_sharded` C entry points are collective
/// calls over all ranks; without a communicator they run the single-GPU path.
pub fn comm_unique_id() -> Option<[u8; sys::ARK_HIP_COMM_ID_BYTES]> {
    let mut id = [0u8; sys::ARK_HIP_COMM_ID_BYTES];
    (unsafe { sys::ark_hip_comm_unique_id(id.as_mut_ptr() as *mut c_void) } == 0).then_some(id)
}
pub fn comm_init(id: &[u8; sys::ARK_HIP_COMM_ID_BYTES], rank: usize, world: usize) -> bool {
    unsafe { sys::ark_hip_comm_init(id.as_ptr() as *const c_void, rank as c_int, world as c_int) == 0 }
}
pub fn comm_destroy() -> bool {
    unsafe { sys::ark_hip_comm_destroy() == 0 }
}

/// An MSM in flight on the device.  `wait` blocks until the result is there.
pub struct MsmJob<'a, P: SWCurveConfig> {
    job: *mut sys::ark_hip_msm_job,
    _inputs: PhantomData<&'a P>, // the scalars (and bases) must outlive the job
}
impl<P: SWCurveConfig> MsmJob<'_, P> {
    pub fn wait(mut self) -> Option<Projective<P>> {
        let mut out = MaybeUninit::<Projective<P>>::uninit();
        let rc = unsafe { sys::ark_hip_msm_wait(self.job, out.as_mut_ptr() as *mut u64) };
        self.job = core::ptr::null_mut();
        (rc == 0).then(|| unsafe { out.assume_init() })
    }
}
impl<P: SWCurveConfig> Drop for MsmJob<'_, P> {
    fn drop(&mut self) {
        if !self.job.is_null() {
            unsafe { sys::ark_hip_msm_wait(self.job, core::ptr::null_mut()) }; // never leak a device job slot
        }
    }
}

/// A base set (an SRS) kept in GPU memory across MSMs together with its per-window multiples (include/ark_hip.h,
/// "prepared base sets"): uploaded and precomputed once, then every call moves only the scalars.  What a prover's
/// commit loop holds in place of the `bases: &[Affine]` argument of `VariableBaseMSM::msm`.
pub struct PreparedBases<P: SWCurveConfig> {
    handle: *mut sys::ark_hip_msm_bases,
    n: usize,
    _p: PhantomData<P>,
}
unsafe impl<P: SWCurveConfig> Send for PreparedBases<P> {}
unsafe impl<P: SWCurveConfig> Sync for PreparedBases<P> {} // the library locks per device
impl<P: HipServed> PreparedBases<P> {
    pub fn new(curve: c_int, bases: &[Affine<P>]) -> Option<Self> {
        let curve = served_id::<P>(curve);
        if !layout_ok::<P, P::ScalarField>(curve) {
            return None;
        }
        let mut h = core::ptr::null_mut();
        let rc = unsafe { sys::ark_hip_msm_bases_prepare(curve, bases.as_ptr() as *const u64, bases.len(), &mut h) };
        (rc == 0).then(|| Self { handle: h, n: bases.len(), _p: PhantomData })
    }
}
impl<P: SWCurveConfig> PreparedBases<P> {
    pub fn len(&self) -> usize {
        self.n
    }
    pub fn is_empty(&self) -> bool {
        self.n == 0
    }
    fn run<S>(&self, scalars: &[S], montgomery: bool) -> Option<Projective<P>> {
        let n = self.n.min(scalars.len());
        let mut out = MaybeUninit::<Projective<P>>::uninit();
        let rc = unsafe {
            sys::ark_hip_msm_prepared(self.handle, scalars.as_ptr() as *const u64, n, montgomery as c_int,
                                      out.as_mut_ptr() as *mut u64)
        };
        (rc == 0).then(|| unsafe { out.assume_init() })
    }
    /// `msm`: `Err(min_len)` on a length mismatch (variable_base/mod.rs:73-77); `Ok(None)` on device error.
    pub fn msm(&self, scalars: &[P::ScalarField]) -> Result<Option<Projective<P>>, usize> {
        if scalars.len() != self.n {
            return Err(scalars.len().min(self.n));
        }
        Ok(self.run(scalars, true))
    }
    pub fn msm_unchecked(&self, scalars: &[P::ScalarField]) -> Option<Projective<P>> {
        self.run(scalars, true)
    }
    pub fn msm_bigint(&self, bigints: &[BigIntOf<P>]) -> Option<Projective<P>> {
        self.run(bigints, false)
    }
    /// This rank's shard of ONE MSM over all ranks (base-range shards, variable_base/mod.rs:521-557): the scalars are this
    /// rank's range, already in device memory (`d_scalars`: a device pointer to `n` canonical `BigInt`s); the partials
    /// are all-gathered over RCCL inside the library and every rank receives the same sum.  Collective.
    ///
    /// # Safety
    /// `d_scalars` must point to `n * 32` readable bytes of this GPU's memory.
    pub unsafe fn msm_bigint_sharded(&self, d_scalars: *const c_void, n: usize) -> Option<Projective<P>> {
        let mut out = MaybeUninit::<Projective<P>>::uninit();
        let rc = sys::ark_hip_msm_prepared_device_sharded(self.handle, d_scalars, n.min(self.n), 0, out.as_mut_ptr() as *mut u64);
        (rc == 0).then(|| out.assume_init())
    }
    /// Enqueue and return: the scalars upload on the copy stream while the previous MSM computes.
    pub fn msm_bigint_async<'a>(&'a self, bigints: &'a [BigIntOf<P>]) -> Option<MsmJob<'a, P>> {
        let n = self.n.min(bigints.len());
        let mut job = core::ptr::null_mut();
        let rc = unsafe { sys::ark_hip_msm_prepared_async(self.handle, bigints.as_ptr() as *const u64, n, 0, &mut job) };
        (rc == 0).then(|| MsmJob { job, _inputs: PhantomData })
    }
}
impl<P: SWCurveConfig> Drop for PreparedBases<P> {
    fn drop(&mut self) {
        unsafe { sys::ark_hip_msm_bases_free(self.handle) };
    }
}

/// `BatchMulPreprocessing` (ec/src/scalar_mul/mod.rs:156-251) on the device: the table of multiples of one group
/// element is built and kept in GPU memory; `batch_mul(v)` returns `v[i] * base` as affine points.
pub struct BatchMulTable<P: SWCurveConfig> {
    handle: *mut sys::ark_hip_batch_mul_table,
    _p: PhantomData<P>,
}
impl<P: HipServed> BatchMulTable<P> {
    /// `BatchMulPreprocessing::new(base, num_scalars)`
    pub fn new(curve: c_int, base: Projective<P>, num_scalars: usize) -> Option<Self> {
        let curve = served_id::<P>(curve);
        if !layout_ok::<P, P::ScalarField>(curve) {
            return None;
        }
        let mut h = core::ptr::null_mut();
        let rc = unsafe { sys::ark_hip_batch_mul_table_new(curve, &base as *const _ as *const u64, num_scalars, &mut h) };
        (rc == 0).then(|| Self { handle: h, _p: PhantomData })
    }
}
impl<P: SWCurveConfig> BatchMulTable<P> {
    /// `BatchMulPreprocessing::batch_mul`
    pub fn batch_mul(&self, v: &[P::ScalarField]) -> Option<Vec<Affine<P>>> {
        let mut out: Vec<Affine<P>> = Vec::with_capacity(v.len());
        let rc = unsafe { sys::ark_hip_batch_mul(self.handle, v.as_ptr() as *const u64, v.len(), 1, out.as_mut_ptr() as *mut u64) };
        if rc != 0 {
            return None;
        }
        unsafe { out.set_len(v.len()) }; // every element written by the library (x | y, identity = (0, 0))
        Some(out)
    }
}
impl<P: SWCurveConfig> Drop for BatchMulTable<P> {
    fn drop(&mut self) {
        unsafe { sys::ark_hip_batch_mul_table_free(self.handle) };
    }
}

/// `CurveGroup::normalize_batch` (ec/src/models/short_weierstrass/group.rs:302-319) on the device: one upload of the
/// Projective points, a lane-batched Montgomery inversion (ff/src/fields/mod.rs:358-385 per lane), one download of the
/// Affine points (identity -> `Affine::identity()`, the all-zero encoding of `ZeroFlag = ()`).  The hook patches/0004 adds
/// to `SWCurveConfig` calls this; `None` (the CPU path runs) below 2^12 points, on a layout mismatch or a device error.
#[cfg(feature = "ec-hook")]
pub fn sw_normalize_batch<P: HipServed>(curve: c_int, v: &[Projective<P>]) -> Option<Vec<Affine<P>>> {
    let curve = served_id::<P>(curve);
    const MIN_POINTS: usize = 1 << 12; // PCIe both ways: 240 B per BLS12-381 G1 point against ~0.2 us of CPU work
    if v.len() < MIN_POINTS || !layout_ok::<P, P::ScalarField>(curve) {
        return None;
    }
    let mut out: Vec<Affine<P>> = Vec::with_capacity(v.len());
    let rc = unsafe { sys::ark_hip_sw_normalize_batch(curve, v.as_ptr() as *const u64, v.len(), out.as_mut_ptr() as *mut u64) };
    if rc != 0 {
        return None;
    }
    unsafe { out.set_len(v.len()) }; // every element written by the library
    Some(out)
}

/// `ScalarMul::batch_mul` (ec/src/scalar_mul/mod.rs:106-109: `BatchMulPreprocessing::new(self, v.len())` +
/// `batch_mul_with_preprocessing`) on the device: the table of multiples is built in GPU memory (sized from `v.len()` by the
/// device's own cost rule), the batch is one mixed addition per table row and scalar, the affine results come back once.
/// The hook patches/0004 adds to `SWCurveConfig` calls this; `None` below 2^10 scalars, on a layout mismatch or an error.
#[cfg(feature = "ec-hook")]
pub fn sw_batch_mul<P: HipServed>(curve: c_int, base: &Projective<P>, v: &[P::ScalarField]) -> Option<Vec<Affine<P>>> {
    let curve = served_id::<P>(curve);
    const MIN_SCALARS: usize = 1 << 10;
    if v.len() < MIN_SCALARS {
        return None;
    }
    BatchMulTable::<P>::new(curve, *base, v.len())?.batch_mul(v)
}

/// Page-locked host buffer of scalars (ark_hip_host_alloc): uploads at full PCIe rate and truly asynchronously.
pub struct PinnedScalars<S: Copy> {
    ptr: *mut S,
    len: usize,
}
impl<S: Copy> PinnedScalars<S> {
    pub fn new(len: usize) -> Option<Self> {
        let mut p: *mut c_void = core::ptr::null_mut();
        let rc = unsafe { sys::ark_hip_host_alloc(len * size_of::<S>(), &mut p) };
        (rc == 0).then(|| Self { ptr: p as *mut S, len })
    }
    pub fn as_mut_slice(&mut self) -> &mut [MaybeUninit<S>] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr as *mut MaybeUninit<S>, self.len) }
    }
    /// # Safety
    /// every element must have been written
    pub unsafe fn assume_init(&self) -> &[S] {
        core::slice::from_raw_parts(self.ptr, self.len)
    }
}
impl<S: Copy> Drop for PinnedScalars<S> {
    fn drop(&mut self) {
        unsafe { sys::ark_hip_host_free(self.ptr as *mut c_void) };
    }
}

/// Declares `$name`, a drop-in `SWCurveConfig` equal to the upstream config `$up` -- EVERY item of the trait
/// (short_weierstrass/mod.rs:34-203: the three constants, `ZeroFlag`, `mul_by_a`, `add_b`, the subgroup check,
/// cofactor clearing, both scalar multiplications, serialisation) is delegated, so points serialise in the upstream
/// format and keep the upstream's endomorphism-based checks -- except that `msm` (and, with patches/0001,
/// `msm_bigint` and the narrow-scalar `msm_small`) run on the MI355X.  For unmodified arkworks; with patches/0002 the upstream configs do this
/// themselves and `G1Projective` stays the same type.
#[macro_export]
macro_rules! hip_sw_config {
    ($name:ident, $up:ty, $id:expr) => {
        #[derive(Clone, Default, PartialEq, Eq)]
        pub struct $name;
        impl $name {
            #[inline]
            fn to_up(p: &ark_ec::short_weierstrass::Affine<Self>) -> ark_ec::short_weierstrass::Affine<$up> {
                if ark_ec::AffineRepr::is_zero(p) {
                    <ark_ec::short_weierstrass::Affine<$up> as ark_ec::AffineRepr>::zero()
                } else {
                    ark_ec::short_weierstrass::Affine::<$up>::new_unchecked(p.x, p.y)
                }
            }
            #[inline]
            fn from_up(p: ark_ec::short_weierstrass::Affine<$up>) -> ark_ec::short_weierstrass::Affine<Self> {
                if ark_ec::AffineRepr::is_zero(&p) {
                    <ark_ec::short_weierstrass::Affine<Self> as ark_ec::AffineRepr>::zero()
                } else {
                    ark_ec::short_weierstrass::Affine::<Self>::new_unchecked(p.x, p.y)
                }
            }
            #[inline]
            fn proj_to_up(p: &ark_ec::short_weierstrass::Projective<Self>) -> ark_ec::short_weierstrass::Projective<$up> {
                ark_ec::short_weierstrass::Projective::<$up>::new_unchecked(p.x, p.y, p.z)
            }
            #[inline]
            fn proj_from_up(p: ark_ec::short_weierstrass::Projective<$up>) -> ark_ec::short_weierstrass::Projective<Self> {
                ark_ec::short_weierstrass::Projective::<Self>::new_unchecked(p.x, p.y, p.z)
            }
        }
        // the wrapper config IS the upstream curve (every item delegated below) under the library id the macro was given
        unsafe impl $crate::msm::HipServed for $name {
            const CURVE: core::ffi::c_int = $id;
        }
        impl ark_ec::CurveConfig for $name {
            type BaseField = <$up as ark_ec::CurveConfig>::BaseField;
            type ScalarField = <$up as ark_ec::CurveConfig>::ScalarField;
            const COFACTOR: &'static [u64] = <$up as ark_ec::CurveConfig>::COFACTOR;
            const COFACTOR_INV: Self::ScalarField = <$up as ark_ec::CurveConfig>::COFACTOR_INV;
        }
        impl ark_ec::short_weierstrass::SWCurveConfig for $name {
            const COEFF_A: Self::BaseField = <$up as ark_ec::short_weierstrass::SWCurveConfig>::COEFF_A;
            const COEFF_B: Self::BaseField = <$up as ark_ec::short_weierstrass::SWCurveConfig>::COEFF_B;
            const GENERATOR: ark_ec::short_weierstrass::Affine<Self> = ark_ec::short_weierstrass::Affine::new_unchecked(
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::GENERATOR.x,
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::GENERATOR.y,
            );
            type ZeroFlag = <$up as ark_ec::short_weierstrass::SWCurveConfig>::ZeroFlag;

            #[inline(always)]
            fn mul_by_a(elem: Self::BaseField) -> Self::BaseField {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::mul_by_a(elem)
            }
            #[inline(always)]
            fn add_b(elem: Self::BaseField) -> Self::BaseField {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::add_b(elem)
            }
            fn is_in_correct_subgroup_assuming_on_curve(item: &ark_ec::short_weierstrass::Affine<Self>) -> bool {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::is_in_correct_subgroup_assuming_on_curve(&Self::to_up(item))
            }
            fn clear_cofactor(item: &ark_ec::short_weierstrass::Affine<Self>) -> ark_ec::short_weierstrass::Affine<Self> {
                Self::from_up(<$up as ark_ec::short_weierstrass::SWCurveConfig>::clear_cofactor(&Self::to_up(item)))
            }
            fn mul_projective(base: &ark_ec::short_weierstrass::Projective<Self>, scalar: &[u64])
                              -> ark_ec::short_weierstrass::Projective<Self> {
                Self::proj_from_up(<$up as ark_ec::short_weierstrass::SWCurveConfig>::mul_projective(&Self::proj_to_up(base), scalar))
            }
            fn mul_affine(base: &ark_ec::short_weierstrass::Affine<Self>, scalar: &[u64])
                          -> ark_ec::short_weierstrass::Projective<Self> {
                Self::proj_from_up(<$up as ark_ec::short_weierstrass::SWCurveConfig>::mul_affine(&Self::to_up(base), scalar))
            }
            /// short_weierstrass/mod.rs:112-119: the length check, then the sum on the GPU; any device error falls back
            /// to the reference's CPU path.
            fn msm(bases: &[ark_ec::short_weierstrass::Affine<Self>], scalars: &[Self::ScalarField])
                   -> Result<ark_ec::short_weierstrass::Projective<Self>, usize> {
                $crate::msm::sw_msm::<Self>($id, bases, scalars)
            }
            #[cfg(feature = "ec-hook")]
            fn msm_bigint(bases: &[ark_ec::short_weierstrass::Affine<Self>],
                          bigints: &[<Self::ScalarField as ark_ff::PrimeField>::BigInt])
                          -> ark_ec::short_weierstrass::Projective<Self> {
                $crate::msm::sw_msm_bigint::<Self>($id, bases, bigints)
            }
            #[cfg(feature = "ec-hook")]
            fn msm_small(bases: &[ark_ec::short_weierstrass::Affine<Self>],
                         scalars: ark_ec::scalar_mul::variable_base::SmallScalars<'_>)
                         -> ark_ec::short_weierstrass::Projective<Self> {
                $crate::msm::sw_msm_small::<Self>($id, bases, scalars)
            }
            #[cfg(feature = "ec-hook")]
            fn normalize_batch(v: &[ark_ec::short_weierstrass::Projective<Self>])
                               -> Option<ark_std::vec::Vec<ark_ec::short_weierstrass::Affine<Self>>> {
                $crate::msm::sw_normalize_batch::<Self>($id, v)
            }
            #[cfg(feature = "ec-hook")]
            fn batch_mul(base: &ark_ec::short_weierstrass::Projective<Self>, v: &[Self::ScalarField])
                         -> Option<ark_std::vec::Vec<ark_ec::short_weierstrass::Affine<Self>>> {
                $crate::msm::sw_batch_mul::<Self>($id, base, v)
            }
            #[inline]
            fn serialize_with_mode<W: ark_serialize::Write>(item: &ark_ec::short_weierstrass::Affine<Self>, writer: W,
                                                            compress: ark_serialize::Compress)
                                                            -> Result<(), ark_serialize::SerializationError> {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::serialize_with_mode(&Self::to_up(item), writer, compress)
            }
            fn deserialize_with_mode<R: ark_serialize::Read>(reader: R, compress: ark_serialize::Compress,
                                                             validate: ark_serialize::Validate)
                                                             -> Result<ark_ec::short_weierstrass::Affine<Self>, ark_serialize::SerializationError> {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::deserialize_with_mode(reader, compress, validate).map(Self::from_up)
            }
            #[inline]
            fn serialized_size(compress: ark_serialize::Compress) -> usize {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::serialized_size(compress)
            }
        }
    };
}
