//! nova-mi355x-sys: the crate INTEGRATION.md section 1 describes, as files.  `ffi` is generated from include/nova_mi355x.h
//! (scripts/gen_rust_sys.py); the functions below are the thin unsafe layer the provider override in
//! src/provider/bn256_grumpkin.rs:43-78 / src/provider/pasta.rs:33-47 calls, shaped like the
//! blitzar wiring at src/provider/blitzar.rs:7-40.
//!
//! NEVER COMPILED in the image this repository is built in (no cargo / rustc there): tests/test_integration_shim.py checks
//! every declaration and every call below against the C header mechanically -- names, arity, ABI class of each parameter.
pub mod ffi;
pub use ffi::*;

use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::sync::OnceLock;

/// GPU present and initialised?  `nmx_init` is idempotent; a box without a GPU keeps the CPU path for good -- the library
/// itself never computes on the CPU (every entry point returns NMX_E_NO_DEVICE there).
pub fn available() -> bool {
    static OK: OnceLock<bool> = OnceLock::new();
    // one process, all GPUs of the node: large keys are sharded by the library, every MSM stays one synchronous call
    *OK.get_or_init(|| unsafe { nmx_init(-1) == 0 && nmx_init_devices(0, 0) == 0 })
}

/// The calling thread's last error message (empty when none).
pub fn last_error() -> String {
    unsafe {
        let p = nmx_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// Below this many pairs the CPU `msm()` is at least as fast as a call into the library (per curve).
pub fn min_gpu_n(curve: c_int) -> usize {
    unsafe { nmx_min_gpu_n(curve) }
}

/// One-time self-check of the zero-copy layout for one curve: the raw bytes of the standard generator and of
/// `Scalar::from(7)` must be x * 2^256 mod p limbs (halo2curves does not promise `repr(C)`).
pub unsafe fn layout_ok(curve: c_int, generator_raw: *const c_void, seven_raw: *const c_void) -> bool {
    nmx_check_layout(curve, generator_raw, seven_raw, 7) == 0
}

/// Zero-copy form of `vartime_multiscalar_mul` (src/provider/traits.rs:79): `scalars` / `bases` are the slices' own memory --
/// halo2curves' 4 x u64 Montgomery limbs, 32 bytes per field element, 64 per affine point (x || y, identity = all zero).
/// `None` on any error: the caller falls back to the CPU `msm()`, so `commit` stays infallible and never returns a wrong point.
pub unsafe fn msm_raw(curve: c_int, scalars: *const c_void, bases: *const c_void, n: usize) -> Option<([u8; 64], bool)> {
    let (mut out, mut inf) = ([0u8; 64], 0u8);
    let rc = nmx_msm(curve, scalars, bases, n, NMX_SCALARS_MONT | NMX_BASES_MONT, out.as_mut_ptr(), &mut inf);
    if rc == 0 { Some((out, inf != 0)) } else { None }
}

/// `msm_small_with_max_num_bits` (src/provider/msm.rs:478-503) on u64 scalars.
pub unsafe fn msm_u64_raw(curve: c_int, scalars: &[u64], bases: *const c_void, max_bits: u32) -> Option<([u8; 64], bool)> {
    let (mut out, mut inf) = ([0u8; 64], 0u8);
    let rc = nmx_msm_u64(curve, scalars.as_ptr(), bases, scalars.len(), max_bits, NMX_BASES_MONT, out.as_mut_ptr(), &mut inf);
    if rc == 0 { Some((out, inf != 0)) } else { None }
}

/// `batch_vartime_multiscalar_mul` (src/provider/traits.rs:82-90): every vector against a prefix of one base array.
pub unsafe fn msm_batch_raw(curve: c_int, vecs: &[(*const c_void, usize)], bases: *const c_void, n_bases: usize)
    -> Option<Vec<([u8; 64], bool)>> {
    let ptrs: Vec<*const c_void> = vecs.iter().map(|v| v.0).collect();
    let lens: Vec<usize> = vecs.iter().map(|v| v.1).collect();
    let (mut out, mut inf) = (vec![0u8; 64 * vecs.len()], vec![0u8; vecs.len()]);
    let rc = nmx_msm_batch(curve, ptrs.as_ptr(), lens.as_ptr(), vecs.len(), bases, n_bases,
                           NMX_SCALARS_MONT | NMX_BASES_MONT, out.as_mut_ptr(), inf.as_mut_ptr());
    if rc != 0 { return None; }
    Some((0..vecs.len()).map(|j| (out[64 * j..64 * j + 64].try_into().unwrap(), inf[j] != 0)).collect())
}

/// A commitment begun with `nmx_commit_begin`: the MSM runs while the caller goes on;
/// `finish` returns the point.  The witness vector must outlive the ticket.  Dropping an unfinished ticket finishes it.
pub struct PendingCommit { ticket: u64 }

impl PendingCommit {
    pub unsafe fn begin(ck_handle: u64, v: *const c_void, n: usize, h_xy64: *const c_void, r: *const c_void, flags: u32) -> Option<Self> {
        let mut ticket = 0u64;
        let rc = nmx_commit_begin(ck_handle, v, n, h_xy64, r, flags, &mut ticket);
        if rc == 0 { Some(PendingCommit { ticket }) } else { None }
    }
    pub fn finish(mut self) -> Option<([u8; 64], bool)> {
        let (mut out, mut inf) = ([0u8; 64], 0u8);
        let t = std::mem::replace(&mut self.ticket, 0);
        let rc = unsafe { nmx_commit_finish(t, out.as_mut_ptr(), &mut inf) };
        if rc == 0 { Some((out, inf != 0)) } else { None }
    }
}

impl Drop for PendingCommit {
    fn drop(&mut self) {
        if self.ticket != 0 {
            let (mut out, mut inf) = ([0u8; 64], 0u8);
            unsafe { nmx_commit_finish(self.ticket, out.as_mut_ptr(), &mut inf) };
        }
    }
}
