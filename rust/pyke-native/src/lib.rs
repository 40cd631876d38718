//! UNCOMPILED (no Rust toolchain in the build image) — the safe layer a maintainer would put between pyke Diffusers and
//! `sdengine-sys`.  Reference paths are relative to the pyke Diffusers repository.
//!
//! * `NativeEngine` replaces the `unet: Session` / `vae_decoder: Session` fields (src/pipelines/stable_diffusion/impl_main.rs:59,64).
//! * `NativeEngine::unet` is what `session.unet.run(..)` becomes (impl_txt2img.rs:283), `decode_latents` the body of
//!   impl_main.rs:398-413, `txt2img` the fused loop for schedulers that describe themselves (`NativeSchedulerKind`).
use std::ffi::{c_void, CStr, CString};
use std::os::raw::{c_float, c_int};
use std::path::Path;
use std::ptr;

use anyhow::{anyhow, bail, Result};
use ndarray::{Array4, ArrayView3, ArrayView4};
use sdengine_sys as sys;

fn check(status: c_int) -> Result<()> {
    if status == sys::SDE_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::sde_last_error()) }.to_string_lossy().into_owned();
    if status == sys::SDE_ERR_SCHEDULER {
        // the reference panics on scheduler misuse (src/schedulers/euler_discrete.rs:140,181): stay compatible
        panic!("{msg}");
    }
    Err(anyhow!("sdengine status {status}: {msg}"))
}

/// Owns a `*mut sde_engine`.  The engine serialises calls internally (`Session` is Send + Sync in the reference too).
pub struct NativeEngine(*mut sys::sde_engine);
unsafe impl Send for NativeEngine {}
unsafe impl Sync for NativeEngine {}

impl Drop for NativeEngine {
    fn drop(&mut self) {
        unsafe { sys::sde_engine_destroy(self.0) }
    }
}

impl NativeEngine {
    /// `StableDiffusionPipeline::new` for the `DiffusionDevice::NativeMI355X(device_id)` choice (src/lib.rs:81-107, impl_main.rs:109-181).
    pub fn new(root: &Path, device_id: usize) -> Result<Self> {
        let root = CString::new(root.to_string_lossy().as_bytes())?;
        let mut o = unsafe { std::mem::zeroed::<sys::sde_engine_opts>() };
        let mut e = ptr::null_mut();
        unsafe {
            sys::sde_engine_opts_default(&mut o);
            o.device_id = device_id as c_int;
            check(sys::sde_engine_create(root.as_ptr(), &o, &mut e))?;
        }
        Ok(Self(e))
    }

    /// impl_txt2img.rs:283 — `sample` [N,4,h,w], `ehs` [N,S,D] -> noise prediction [N,4,h,w].
    pub fn unet(&self, sample: ArrayView4<'_, f32>, timestep: f32, ehs: ArrayView3<'_, f32>) -> Result<Array4<f32>> {
        let (n, c, h, w) = sample.dim();
        let (n2, s, d) = ehs.dim();
        if n != n2 {
            bail!("sample and encoder_hidden_states disagree on the batch size");
        }
        let (sample, ehs) = (sample.as_standard_layout(), ehs.as_standard_layout());
        let mut out = Array4::<f32>::zeros((n, c, h, w));
        unsafe {
            check(sys::sde_unet_forward(self.0, sample.as_ptr(), n as c_int, h as c_int, w as c_int, timestep as c_float, ehs.as_ptr(),
                                        s as c_int, d as c_int, out.as_mut_ptr()))?;
        }
        Ok(out)
    }

    /// impl_main.rs:398-413 — latents [B,4,h,w] -> images [B,8h,8w,3] in [0,1] (`to_image` wraps each slice in an Rgb32FImage).
    pub fn decode_latents(&self, latents: ArrayView4<'_, f32>) -> Result<Array4<f32>> {
        let (b, _, h, w) = latents.dim();
        let latents = latents.as_standard_layout();
        let mut out = Array4::<f32>::zeros((b, 8 * h, 8 * w, 3));
        unsafe { check(sys::sde_vae_decode(self.0, latents.as_ptr(), b as c_int, h as c_int, w as c_int, out.as_mut_ptr()))? };
        Ok(out)
    }
}

/// What a built-in scheduler returns from the provided trait method `fn native_descriptor(&self) -> Option<NativeSchedulerKind>`
/// (src/schedulers/mod.rs:154); user schedulers return `None` and keep the per-call path (`NativeEngine::unet` + their own `step`).
pub struct NativeSchedulerKind { pub kind: c_int, pub cfg: sys::sde_sched_cfg }

pub struct NativeScheduler(*mut sys::sde_scheduler);
impl Drop for NativeScheduler {
    fn drop(&mut self) {
        unsafe { sys::sde_scheduler_destroy(self.0) }
    }
}
impl NativeScheduler {
    pub fn new(k: &NativeSchedulerKind) -> Result<Self> {
        let mut s = ptr::null_mut();
        unsafe { check(sys::sde_scheduler_create(k.kind, &k.cfg, &mut s))? };
        Ok(Self(s))
    }
}

/// The closure kinds of `StableDiffusionCallback` (src/pipelines/stable_diffusion/mod.rs:41-86), type-erased.
pub type StepCallback<'a> = dyn FnMut(usize, f32, &[f32]) -> bool + 'a;

unsafe extern "C" fn trampoline(step: c_int, t: c_float, payload: *const c_float, len: usize, user: *mut c_void) -> c_int {
    let cb = &mut *(user as *mut &mut StepCallback<'_>);
    let data: &[f32] = if payload.is_null() { &[] } else { std::slice::from_raw_parts(payload, len) };
    // a panic must not unwind into C: treat it as "stop"
    std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| cb(step as usize, t, data))).map(|k| k as c_int).unwrap_or(0)
}

pub struct Txt2Img<'a> {
    pub height: u32, pub width: u32, pub guidance_scale: f32, pub rescale_cfg: Option<f32>, pub steps: usize, pub seed: Option<u64>,
    /// `encode_prompt` output (impl_main.rs:339-375): [2B,S,D] with the unconditional half first when guidance_scale > 1
    pub text_embeddings: ArrayView3<'a, f32>,
    pub batch_size: usize,
    pub callback: Option<(c_int, usize, &'a mut StepCallback<'a>)>,   // (sde_callback_kind, frequency, closure)
}

impl NativeEngine {
    /// The whole of `StableDiffusionTxt2ImgOptions::run` (impl_txt2img.rs:246-327) on the device.
    pub fn txt2img(&self, sched: &mut NativeScheduler, o: Txt2Img<'_>) -> Result<Array4<f32>> {
        let ehs = o.text_embeddings.as_standard_layout();
        let (_, s, d) = ehs.dim();
        let mut opts = unsafe { std::mem::zeroed::<sys::sde_txt2img_opts>() };
        let mut images = Array4::<f32>::zeros((o.batch_size, o.height as usize, o.width as usize, 3));
        let mut cb_holder;
        unsafe {
            sys::sde_txt2img_opts_default(&mut opts);
            opts.height = o.height as c_int; opts.width = o.width as c_int; opts.guidance_scale = o.guidance_scale;
            if let Some(m) = o.rescale_cfg { opts.has_rescale_cfg = 1; opts.rescale_cfg = m; }
            opts.steps = o.steps as c_int;
            if let Some(seed) = o.seed { opts.has_seed = 1; opts.seed = seed; }
            opts.batch_size = o.batch_size as c_int;
            opts.ehs = ehs.as_ptr(); opts.s = s as c_int; opts.d = d as c_int;
            if let Some((kind, freq, cb)) = o.callback {
                cb_holder = cb;
                opts.callback_kind = kind; opts.callback_frequency = freq as c_int;
                opts.callback = Some(trampoline);
                opts.callback_user = &mut cb_holder as *mut &mut StepCallback<'_> as *mut c_void;
            }
            check(sys::sde_txt2img_run(self.0, sched.0, &opts, images.as_mut_ptr(), ptr::null_mut()))?;
        }
        Ok(images)
    }
}
