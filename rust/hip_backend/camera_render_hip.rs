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

    /// Samples per pixel between two window refreshes; 0 = one pass of all samples.
    const SAMPLES_PER_PASS: i32 = 10;

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

    let mut cam = sys::dray_camera {
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

    // Each call continues the Philox streams of the samples the buffers already hold.
    let first_sample = count_buffer.first().copied().unwrap_or(0).max(0) as u32;
    let total = self.samples_per_pixel;
    let per_pass = if SAMPLES_PER_PASS > 0 && SAMPLES_PER_PASS < total { SAMPLES_PER_PASS } else { total };
    let mut done = 0;
    loop {
        cam.samples_per_pixel = if total > 0 { per_pass.min(total - done) } else { total };
        opts.sample_offset = first_sample + done as u32;
        // devices = null, n_devices = 0: every GPU of the node, interleaved scanlines, whole scene replicated
        let rc = unsafe {
            sys::dray_render_multi(&cam, spheres.as_ptr(), spheres.len(), color_buffer.as_mut_ptr(), raw.as_mut_ptr(),
                                   count_buffer.as_mut_ptr(), used, &opts, std::ptr::null(), 0, std::ptr::null_mut())
        };
        if rc != sys::DRAY_OK {
            let msg = unsafe { std::ffi::CStr::from_ptr(sys::dray_strerror(rc)) }.to_string_lossy().into_owned();
            return Err(Error::new(ErrorKind::Other, msg));
        }
        let _ = window.update_with_buffer(&color_buffer, width, height); // colors.rs:44, once per pass
        done += cam.samples_per_pixel;
        if done >= total {
            break;
        }
    }

    for (v, c) in raw_buffer.iter_mut().zip(raw.chunks_exact(3)) {
        *v = Vec3::new([c[0], c[1], c[2]]);
    }
    Ok(())
}
