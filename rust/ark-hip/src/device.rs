//! Device-resident vectors of a scalar field: `DensePolynomial` coefficients / `Evaluations` that stay in HBM.
//!
//! The reference's FFT callers chain transforms: `DensePolynomial::evaluate_over_domain`
//! (poly/src/polynomial/univariate/mod.rs:305-360: zero-extend + `fft_in_place` -> `Evaluations`), the pointwise
//! `+`, `-`, `*` of `Evaluations` over one domain (poly/src/evaluations/univariate/mod.rs:104-180) and
//! `Evaluations::interpolate` (mod.rs:40-50: `ifft_in_place` -> coefficients).  Through the `EvaluationDomain` hook every
//! link of such a chain crosses PCIe twice (2 x 128 MiB at 2^22: 5.3 ms around a 0.5 ms transform -- upload, transform and
//! download are serial by data dependence, so no pipelining removes them).  [`DeviceVec`] owns `ark_hip_malloc` memory;
//! with [`DeviceVec::evaluate_over_domain`], the `*Assign` operators of [`DeviceEvaluations`] and
//! [`DeviceEvaluations::interpolate`] the chain costs ONE upload per input and ONE download.
//!
//! Every operation is asynchronous on the library's stream of the current device and ordered with the others;
//! [`DeviceVec::to_vec`] waits.  A vector belongs to the device that was current when it was created: it is freed there, and
//! an operation attempted while another device is current returns `DeviceError::WrongDevice` instead of touching it.
//! Mirrors: `ark_hip::DeviceVec` in include/ark_hip.hpp (compiled and run by tests/test_gpu_cpp_mirror.py against the
//! oracle at 2^20), `algebra_amd.DeviceVec` in Python.
use ark_ff::FftField;
use ark_hip_sys as sys;
use ark_poly::domain::{EvaluationDomain, Radix2EvaluationDomain};
use ark_std::vec::Vec;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;
use core::ops::{AddAssign, DivAssign, MulAssign, SubAssign};

/// Why a device operation did not happen; the data a `DeviceVec` holds is unchanged when an `Err` comes back.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum DeviceError {
    /// `F` is not one of the scalar fields the library serves, or its in-memory size is not 32 bytes.
    UnsupportedField,
    /// lengths / domains of the operands differ (the reference asserts `self.domain == other.domain`)
    Mismatch,
    /// the vector lives on another device than the calling thread's current one (`ark_hip_set_device`)
    WrongDevice,
    /// the library's return code (include/ark_hip.h `ARK_HIP_ERR_*`)
    Library(c_int),
}
fn rc(code: c_int) -> Result<(), DeviceError> {
    if code == 0 { Ok(()) } else { Err(DeviceError::Library(code)) }
}

pub struct DeviceVec<F: FftField> {
    ptr: *mut c_void,
    len: usize,
    cap: usize,
    device: c_int,
    field: c_int,
    _f: PhantomData<F>,
}
// the pointer is device memory owned by this value; every access goes through the library, which locks its context
unsafe impl<F: FftField> Send for DeviceVec<F> {}

impl<F: FftField> DeviceVec<F> {
    fn field_id() -> Result<c_int, DeviceError> {
        if core::mem::size_of::<F>() != 32 || core::mem::align_of::<F>() != core::mem::align_of::<u64>() {
            return Err(DeviceError::UnsupportedField);
        }
        sys::fr_field_id::<F>().ok_or(DeviceError::UnsupportedField)
    }
    fn alloc(len: usize) -> Result<Self, DeviceError> {
        let field = Self::field_id()?;
        let mut ptr: *mut c_void = core::ptr::null_mut();
        if len != 0 {
            rc(unsafe { sys::ark_hip_malloc(len * 32, &mut ptr) })?;
        }
        let device = unsafe { sys::ark_hip_get_device() };
        Ok(Self { ptr, len, cap: len, device, field, _f: PhantomData })
    }
    /// `vec![F::zero(); len]` on the device.
    pub fn zeros(len: usize) -> Result<Self, DeviceError> {
        let v = Self::alloc(len)?;
        rc(unsafe { sys::ark_hip_memset_device(v.ptr, 0, len * 32) })?;
        Ok(v)
    }
    /// One upload (synchronous: `x` may be dropped when this returns).
    pub fn from_slice(x: &[F]) -> Result<Self, DeviceError> {
        let v = Self::alloc(x.len())?;
        rc(unsafe { sys::ark_hip_memcpy_h2d(v.ptr, x.as_ptr() as *const c_void, x.len() * 32) })?;
        Ok(v)
    }
    /// One download; waits for everything queued on the vector.
    pub fn to_vec(&self) -> Result<Vec<F>, DeviceError> {
        self.here()?;
        let mut out: Vec<F> = Vec::with_capacity(self.len);
        rc(unsafe { sys::ark_hip_memcpy_d2h(out.as_mut_ptr() as *mut c_void, self.ptr, self.len * 32) })?;
        unsafe { out.set_len(self.len) }; // canonical Montgomery residues, the reference's own representation
        Ok(out)
    }
    pub fn try_clone(&self) -> Result<Self, DeviceError> {
        self.here()?;
        let v = Self::alloc(self.len)?;
        rc(unsafe { sys::ark_hip_memcpy_d2d(v.ptr, self.ptr, self.len * 32) })?;
        Ok(v)
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    /// For the `_device` entry points of `ark_hip_sys` (e.g. the scalars of an MSM straight from an inverse transform).
    pub fn as_device_ptr(&self) -> *const c_void {
        self.ptr
    }
    pub fn as_device_mut_ptr(&mut self) -> *mut c_void {
        self.ptr
    }
    /// `Vec::resize(new_len, F::zero())` (also truncates).
    pub fn resize_zeroed(&mut self, new_len: usize) -> Result<(), DeviceError> {
        self.here()?;
        if new_len > self.cap {
            let mut v = Self::alloc(new_len)?;
            rc(unsafe { sys::ark_hip_memcpy_d2d(v.ptr, self.ptr, self.len * 32) })?;
            v.len = self.len;
            core::mem::swap(self, &mut v); // the old allocation is freed by v's drop
        }
        let old = self.len;
        self.len = new_len;
        if new_len > old {
            rc(unsafe { sys::ark_hip_memset_device((self.ptr as *mut u8).add(old * 32) as *mut c_void, 0, (new_len - old) * 32) })?;
        }
        Ok(())
    }
    /// every operation runs on the stream of the thread's CURRENT device: refuse a vector that lives elsewhere
    fn here(&self) -> Result<(), DeviceError> {
        if self.ptr.is_null() || unsafe { sys::ark_hip_get_device() } == self.device { Ok(()) } else { Err(DeviceError::WrongDevice) }
    }
    fn same_len(&self, other: &Self) -> Result<(), DeviceError> {
        self.here()?;
        other.here()?;
        if self.len == other.len && self.field == other.field { Ok(()) } else { Err(DeviceError::Mismatch) }
    }
    pub fn add_assign_pointwise(&mut self, other: &Self) -> Result<(), DeviceError> {
        self.same_len(other)?;
        rc(unsafe { sys::ark_hip_fr_add_device(self.field, self.ptr, other.ptr, self.ptr, self.len) })
    }
    pub fn sub_assign_pointwise(&mut self, other: &Self) -> Result<(), DeviceError> {
        self.same_len(other)?;
        rc(unsafe { sys::ark_hip_fr_sub_device(self.field, self.ptr, other.ptr, self.ptr, self.len) })
    }
    pub fn mul_assign_pointwise(&mut self, other: &Self) -> Result<(), DeviceError> {
        self.same_len(other)?;
        rc(unsafe { sys::ark_hip_fr_mul_device(self.field, self.ptr, other.ptr, self.ptr, self.len) })
    }
    /// `Evaluations /= &Evaluations` (mod.rs:153-163): a zero divisor gives zero, as the reference's `batch_inversion` leaves
    /// zeros in place and `div_assign` multiplies by them
    pub fn div_assign_pointwise(&mut self, other: &Self) -> Result<(), DeviceError> {
        self.same_len(other)?;
        rc(unsafe { sys::ark_hip_fr_div_device(self.field, self.ptr, other.ptr, self.ptr, self.len) })
    }
    /// `ark_ff::batch_inversion` on the device (zeros stay zero)
    pub fn batch_inverse(&mut self) -> Result<(), DeviceError> {
        self.here()?;
        rc(unsafe { sys::ark_hip_fr_inverse_device(self.field, self.ptr, self.ptr, self.len) })
    }
    /// every element times `k` (`&DensePolynomial * F`, dense.rs:604-622)
    pub fn scale(&mut self, k: &F) -> Result<(), DeviceError> {
        self.here()?;
        let kl = sys::limbs(k);
        rc(unsafe { sys::ark_hip_fr_scale_device(self.field, self.ptr, kl.as_ptr(), self.ptr, self.len) })
    }
    pub fn negate(&mut self) -> Result<(), DeviceError> {
        self.here()?;
        rc(unsafe { sys::ark_hip_fr_neg_device(self.field, self.ptr, self.ptr, self.len) })
    }
    /// `DensePolynomial::evaluate_over_domain` (polynomial/univariate/mod.rs:305-360) for coefficients already on the
    /// device: zero-extension and transform in place; at most size/4 coefficients take the degree-aware path
    /// (radix2/mod.rs:141).  More coefficients than the domain holds is the reference's folding case (mod.rs:330-352):
    /// `Err(Mismatch)`, fold on the host first.
    pub fn evaluate_over_domain(mut self, domain: Radix2EvaluationDomain<F>) -> Result<DeviceEvaluations<F>, DeviceError> {
        let have = self.len;
        if have > domain.size() {
            return Err(DeviceError::Mismatch);
        }
        self.resize_zeroed(domain.size())?;
        let d = raw_domain(&domain);
        rc(unsafe { sys::ark_hip_fft_in_place_degree_aware_device(self.field, &d, self.ptr, have) })?;
        Ok(DeviceEvaluations { evals: self, domain })
    }
}
impl<F: FftField> Drop for DeviceVec<F> {
    fn drop(&mut self) {
        if self.ptr.is_null() {
            return;
        }
        unsafe {
            let cur = sys::ark_hip_get_device();
            if self.device >= 0 && cur != self.device {
                sys::ark_hip_set_device(self.device);
            }
            let r = sys::ark_hip_free(self.ptr); // waits for the work queued on it
            debug_assert_eq!(r, 0);
            if self.device >= 0 && cur != self.device && cur >= 0 {
                sys::ark_hip_set_device(cur);
            }
        }
    }
}

/// The C mirror of the domain (include/ark_hip.h `ark_hip_radix2_domain`) from the reference's public fields.
fn raw_domain<F: FftField>(d: &Radix2EvaluationDomain<F>) -> sys::ark_hip_radix2_domain {
    sys::ark_hip_radix2_domain {
        size: d.size,
        log_size_of_group: d.log_size_of_group,
        _pad: 0,
        size_as_field_element: sys::limbs(&d.size_as_field_element),
        size_inv: sys::limbs(&d.size_inv),
        group_gen: sys::limbs(&d.group_gen),
        group_gen_inv: sys::limbs(&d.group_gen_inv),
        offset: sys::limbs(&d.offset),
        offset_inv: sys::limbs(&d.offset_inv),
        offset_pow_size: sys::limbs(&d.offset_pow_size),
    }
}

/// `Evaluations<F, Radix2EvaluationDomain<F>>` resident on the device (evaluations/univariate/mod.rs:18-29).
pub struct DeviceEvaluations<F: FftField> {
    pub evals: DeviceVec<F>,
    pub domain: Radix2EvaluationDomain<F>,
}
impl<F: FftField> DeviceEvaluations<F> {
    /// `Evaluations::from_vec_and_domain` for values already on the device.
    pub fn from_device_vec_and_domain(evals: DeviceVec<F>, domain: Radix2EvaluationDomain<F>) -> Result<Self, DeviceError> {
        if evals.len() != domain.size() { Err(DeviceError::Mismatch) } else { Ok(Self { evals, domain }) }
    }
    /// `Evaluations::interpolate` (mod.rs:47-50): the `domain.size()` coefficients, still on the device.  The reference
    /// then drops leading zeros (`DensePolynomial::from_coefficients_vec`): do that on the host after `to_vec()`.
    pub fn interpolate(mut self) -> Result<DeviceVec<F>, DeviceError> {
        self.evals.here()?;
        let d = raw_domain(&self.domain);
        rc(unsafe { sys::ark_hip_ifft_in_place_device(self.evals.field, &d, self.evals.ptr) })?;
        Ok(self.evals)
    }
    fn same_domain(&self, other: &Self) {
        assert_eq!(self.domain, other.domain, "domains are unequal"); // the reference's assertion, same message
    }
}
// `Evaluations op= &Evaluations` (mod.rs:104-180).  The reference's operators cannot fail; a device error here panics
// with the library's code (the fallible forms are the `*_assign_pointwise` methods of `DeviceVec`).
impl<'a, F: FftField> AddAssign<&'a DeviceEvaluations<F>> for DeviceEvaluations<F> {
    fn add_assign(&mut self, other: &'a DeviceEvaluations<F>) {
        self.same_domain(other);
        self.evals.add_assign_pointwise(&other.evals).expect("ark-hip: pointwise add on the device");
    }
}
impl<'a, F: FftField> SubAssign<&'a DeviceEvaluations<F>> for DeviceEvaluations<F> {
    fn sub_assign(&mut self, other: &'a DeviceEvaluations<F>) {
        self.same_domain(other);
        self.evals.sub_assign_pointwise(&other.evals).expect("ark-hip: pointwise sub on the device");
    }
}
impl<'a, F: FftField> MulAssign<&'a DeviceEvaluations<F>> for DeviceEvaluations<F> {
    fn mul_assign(&mut self, other: &'a DeviceEvaluations<F>) {
        self.same_domain(other);
        self.evals.mul_assign_pointwise(&other.evals).expect("ark-hip: pointwise mul on the device");
    }
}
impl<'a, F: FftField> DivAssign<&'a DeviceEvaluations<F>> for DeviceEvaluations<F> {
    fn div_assign(&mut self, other: &'a DeviceEvaluations<F>) {
        self.same_domain(other);
        self.evals.div_assign_pointwise(&other.evals).expect("ark-hip: pointwise division on the device");
    }
}
