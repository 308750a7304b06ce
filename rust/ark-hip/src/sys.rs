//! Raw FFI: one declaration per entry point of include/ark_hip.h (the only `unsafe` surface; ark-ec and
//! ark-poly `#![forbid(unsafe_code)]`, ec/src/lib.rs:10, poly/src/lib.rs:4, so this lives in its own crate).
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

pub const BN254_FR: c_int = 1;
pub const BLS12_381_FR: c_int = 3;
pub const BLS12_377_FR: c_int = 5;
pub const BN254_G1: c_int = 0;
pub const BLS12_381_G1: c_int = 1;
pub const BLS12_377_G1: c_int = 2;
pub const BLS12_377_G2: c_int = 3;
pub const BLS12_381_G2: c_int = 4;

/// Mirror of `Radix2EvaluationDomain<F>` (poly/src/domain/radix2/mod.rs:22-42) for a 4-limb F.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ark_hip_radix2_domain {
    pub size: u64,
    pub log_size_of_group: u32,
    pub _pad: u32,
    pub size_as_field_element: [u64; 4],
    pub size_inv: [u64; 4],
    pub group_gen: [u64; 4],
    pub group_gen_inv: [u64; 4],
    pub offset: [u64; 4],
    pub offset_inv: [u64; 4],
    pub offset_pow_size: [u64; 4],
}

extern "C" {
    pub fn ark_hip_device_count() -> c_int;
    pub fn ark_hip_init(device: c_int) -> c_int;
    pub fn ark_hip_shutdown();
    pub fn ark_hip_synchronize() -> c_int;
    pub fn ark_hip_version() -> *const c_char;
    pub fn ark_hip_msm_sw(curve: c_int, bases: *const u64, scalars: *const u64, n: usize,
                          scalars_are_montgomery: c_int, out_xyz: *mut u64) -> c_int;
    pub fn ark_hip_msm_sw_device(curve: c_int, d_bases: *const c_void, d_scalars: *const c_void, n: usize,
                                 scalars_are_montgomery: c_int, out_xyz: *mut u64) -> c_int;
    pub fn ark_hip_malloc(bytes: usize, out_dptr: *mut *mut c_void) -> c_int;
    pub fn ark_hip_free(dptr: *mut c_void) -> c_int;
    pub fn ark_hip_memcpy_h2d(dst_dptr: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn ark_hip_memcpy_d2h(dst_host: *mut c_void, src_dptr: *const c_void, bytes: usize) -> c_int;
    pub fn ark_hip_sw_sum(curve: c_int, jac_points: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    pub fn ark_hip_fft_in_place(field: c_int, dom: *const ark_hip_radix2_domain, data: *mut u64) -> c_int;
    pub fn ark_hip_ifft_in_place(field: c_int, dom: *const ark_hip_radix2_domain, data: *mut u64) -> c_int;
    pub fn ark_hip_fft_in_place_device(field: c_int, dom: *const ark_hip_radix2_domain, d_data: *mut c_void) -> c_int;
    pub fn ark_hip_ifft_in_place_device(field: c_int, dom: *const ark_hip_radix2_domain, d_data: *mut c_void) -> c_int;
}
