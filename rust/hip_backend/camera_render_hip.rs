// The body of `Camera::render` (src/raytracer/camera.rs:208-236) when the crate is built with the feature `hip`:
// same signature, same buffers, same result type; the per-pixel loop runs on the GPU through libdray_hip.so.
// Paste into `impl Camera` beside the existing `render`, which gets `#[cfg(not(feature = "hip"))]`.
//
// What stays as in the reference: `initialize()` first (:215); the three buffers are added to, never cleared
// (colors.rs:38-39), entries past image_width * image_height are left alone (the caller sizes them W x W,
// main.rs:69-73); `Ok(())` on success.  What changes: the window is presented once per pass instead of once per
// pixel sample (colors.rs:44), the per-row progress lines (:220-222) go, and an object the GPU has no form for
// makes the call fall back to the CPU loop instead of failing.

#[cfg(feature = "hip")]
pub fn render(
    &mut self, world: &dyn Hittable,
    window: &mut Window,
    color_buffer: &mut Vec<u32>,
    raw_buffer: &mut Vec<Vec3>,
    count_buffer: &mut Vec<i32>
) -> Result<()> {
    use crate::raytracer::hip_flatten::{flatten, FlattenError};
    use dray_hip_sys as sys;
    use std::io::{Error, ErrorKind};

    /// Samples per pixel between two window refreshes; 0 = one pass of all samples.  (On one MI355X the headline frame
    /// takes 27.5 ms in one pass, 35 ms in ten passes of 50, some 75 ms in fifty passes of 10.)
    const SAMPLES_PER_PASS: i32 = 50;

    self.initialize();

    let spheres = match flatten(world) {
        Ok(list) => list,
        Err(FlattenError::Unsupported(_)) => {
            return self.render_cpu(world, window, color_buffer, raw_buffer, count_buffer); // the reference's loop
        }
        Err(e) => return Err(Error::new(ErrorKind::InvalidData, e.to_string())),
    };

    let width = self.image_width as usize;
    let height = self.image_height as usize;
    let used = width * height;
    let len = color_buffer.len().min(raw_buffer.len()).min(count_buffer.len());
    if len < used {
        return Err(Error::new(ErrorKind::InvalidInput, "buffers shorter than image_width * image_height"));
    }

    let cam = sys::dray_camera {
        aspect_ratio: self.aspect_ratio,
        image_width: self.image_width,
        samples_per_pixel: self.samples_per_pixel,
        max_depth: self.max_depth,
        _pad: 0,
        vfov: self.vfov,
        lookfrom: [self.lookfrom.x(), self.lookfrom.y(), self.lookfrom.z()],
        lookat: [self.lookat.x(), self.lookat.y(), self.lookat.z()],
        vup: [self.vup.x(), self.vup.y(), self.vup.z()],
        defocus_angle: self.defocus_angle,
        focus_dist: self.focus_dist,
    };
    let mut opts = sys::dray_opts::default();
    unsafe { sys::dray_opts_default(&mut opts) };

    // `Vec3` has no #[repr(C)] (vec3.rs:7-10), so the running sums cross the boundary through a plain f64 buffer.
    // (With #[repr(transparent)] on Vec3 the two copies go and `raw_buffer.as_mut_ptr() as *mut f64` is passed.)
    let mut raw: Vec<f64> = Vec::with_capacity(3 * used);
    for v in raw_buffer.iter().take(used) {
        raw.extend_from_slice(&[v.x(), v.y(), v.z()]);
    }

    // A render context over every GPU of the node (interleaved scanlines, whole scene replicated), made once per
    // process: the sphere tables and the grid are rebuilt only when the world changes, and between the passes of one
    // render the running sums stay in GPU memory -- they go up with the first pass and come back with the last, in
    // between only the 4 bytes per pixel of the colour buffer move.
    static CONTEXT: std::sync::OnceLock<usize> = std::sync::OnceLock::new();
    let ctx = *CONTEXT.get_or_init(|| {
        let mut c: *mut sys::dray_context = std::ptr::null_mut();
        let rc = unsafe { sys::dray_context_create(std::ptr::null(), 0, &mut c) };
        if rc == sys::DRAY_OK { c as usize } else { 0 }
    }) as *mut sys::dray_context;
    let fail = |rc: i32| {
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::dray_strerror(rc)) }.to_string_lossy().into_owned();
        Error::new(ErrorKind::Other, msg)
    };
    if ctx.is_null() {
        return Err(fail(sys::DRAY_ERR_NO_DEVICE));
    }
    let rc = unsafe { sys::dray_context_set_scene(ctx, spheres.as_ptr(), spheres.len()) };
    if rc != sys::DRAY_OK {
        return Err(fail(rc));
    }

    // Each call continues the Philox streams of the samples the buffers already hold.
    opts.sample_offset = count_buffer.first().copied().unwrap_or(0).max(0) as u32;
    let total = self.samples_per_pixel;
    let per_pass = if SAMPLES_PER_PASS > 0 && SAMPLES_PER_PASS < total { SAMPLES_PER_PASS } else { total.max(1) };

    // The window is presented after every pass (colors.rs:44 presents after every pixel sample) from a callback the
    // library makes on this thread once the pass's colours are in `color_buffer`; the next pass is rendering by then.
    // The running sums go up before the first pass and come back after the last (dray_context_render with
    // DRAY_SYNC_UPLOAD_SUMS | DRAY_SYNC_DOWNLOAD_SUMS is the same frame in one pass, without the callback).
    struct Show<'a> {
        window: &'a mut Window,
        color: *const u32,
        len: usize,
        width: usize,
        height: usize,
    }
    unsafe extern "C" fn show(user: *mut std::ffi::c_void, _pass: i32, _samples_done: i32) {
        let s = &mut *(user as *mut Show);
        let _ = s.window.update_with_buffer(std::slice::from_raw_parts(s.color, s.len), s.width, s.height);
    }
    let mut user = Show { window, color: color_buffer.as_ptr(), len: color_buffer.len(), width, height };
    let rc = unsafe {
        sys::dray_context_render_progressive(ctx, &cam, color_buffer.as_mut_ptr(), raw.as_mut_ptr(), count_buffer.as_mut_ptr(),
                                             used, &opts, per_pass, Some(show), &mut user as *mut Show as *mut std::ffi::c_void,
                                             std::ptr::null_mut())
    };
    if rc != sys::DRAY_OK {
        return Err(fail(rc));
    }

    for (v, c) in raw_buffer.iter_mut().zip(raw.chunks_exact(3)) {
        *v = Vec3::new([c[0], c[1], c[2]]);
    }
    Ok(())
}
