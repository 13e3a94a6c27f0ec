//! Raw declarations of the C ABI of `libdray_hip.so` (`include/dray.h`), one for one: every `#[repr(C)]` struct
//! has the header's fields in the header's order, every function the header's signature.  A test of the
//! library's repository parses this file and the header and compares them, so keep the two in step.
//!
//! The library replaces the body of `Camera::render` (src/raytracer/camera.rs:208-236 of the path tracer).
#![allow(non_camel_case_types)]
#![no_std]

use core::ffi::{c_char, c_int, c_void};

// ---- dray_status ------------------------------------------------------------------------------------------
pub const DRAY_OK: c_int = 0;
pub const DRAY_ERR_INVALID_ARG: c_int = -1;
pub const DRAY_ERR_BUFFER_TOO_SMALL: c_int = -2;
pub const DRAY_ERR_UNSUPPORTED: c_int = -3;
pub const DRAY_ERR_NO_DEVICE: c_int = -4;
pub const DRAY_ERR_HIP: c_int = -5;
pub const DRAY_ERR_INTERNAL: c_int = -6;
pub const DRAY_ERR_TOO_LARGE: c_int = -7;

// ---- material kinds, shapes, modes, flags, scans ----------------------------------------------------------
pub const DRAY_MAT_DEFAULT: u32 = 0;
pub const DRAY_MAT_LAMBERTIAN: u32 = 1;
pub const DRAY_MAT_METAL: u32 = 2;
pub const DRAY_MAT_DIELECTRIC: u32 = 3;
pub const DRAY_MAT_TRANSPARENT: u32 = 4;

pub const DRAY_SHAPE_SPHERE: u32 = 0;
pub const DRAY_SHAPE_BOX: u32 = 1;

pub const DRAY_MODE_F64: u32 = 0;
pub const DRAY_MODE_F32: u32 = 1;
pub const DRAY_MODE_F32_PLAIN: u32 = 2;

pub const DRAY_FLAG_ITERATIVE_PRODUCT: u32 = 1;
pub const DRAY_FLAG_COMPACT_ROWS: u32 = 2;
pub const DRAY_FLAG_NO_REFILL: u32 = 4;

pub const DRAY_SCAN_AUTO: i32 = 0;
pub const DRAY_SCAN_SCALAR: i32 = 1;
pub const DRAY_SCAN_HYBRID: i32 = 2;
pub const DRAY_SCAN_GRID: i32 = 3;

pub const DRAY_NOALLOC_MAX_PIXELS: usize = 1048576;

pub const DRAY_SYNC_UPLOAD_SUMS: u32 = 1;
pub const DRAY_SYNC_DOWNLOAD_SUMS: u32 = 2;

// ---- structs ----------------------------------------------------------------------------------------------
/// The ten public camera fields, camera.rs:42-53, same order.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_camera {
    pub aspect_ratio: f64,
    pub image_width: i32,
    pub samples_per_pixel: i32,
    pub max_depth: i32,
    pub _pad: i32,
    pub vfov: f64,
    pub lookfrom: [f64; 3],
    pub lookat: [f64; 3],
    pub vup: [f64; 3],
    pub defocus_angle: f64,
    pub focus_dist: f64,
}

/// The private camera fields computed by Camera::initialize, camera.rs:55-65.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_camera_derived {
    pub image_height: i32,
    pub _pad: i32,
    pub pixel_samples_scale: f64,
    pub center: [f64; 3],
    pub pixel00_loc: [f64; 3],
    pub pixel_delta_u: [f64; 3],
    pub pixel_delta_v: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub w: [f64; 3],
    pub defocus_disk_u: [f64; 3],
    pub defocus_disk_v: [f64; 3],
}

/// One Sphere (sphere.rs:6-10) with its material flattened in.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_sphere {
    pub center: [f64; 3],
    pub radius: f64,
    pub kind: u32,
    pub shape: u32,
    pub albedo: [f64; 3],
    pub fuzz_or_ior: f64,
}

/// A BoundingBox (bounding_box.rs:6-9) in a dray_sphere's place in the list.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_box {
    pub min: [f64; 3],
    pub _unused0: f64,
    pub _unused_kind: u32,
    pub shape: u32,
    pub max: [f64; 3],
    pub _unused1: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_opts {
    pub seed: u64,
    pub mode: u32,
    pub flags: u32,
    pub sample_offset: u32,
    pub device: i32,
    pub row_begin: i32,
    pub row_step: i32,
    pub pixels_per_pool: i32,
    pub variant: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_stats {
    pub segments: u64,
    pub samples: u64,
    pub kernel_ms: f64,
    pub resolve_ms: f64,
    pub total_ms: f64,
    pub image_height: i32,
    pub rows_rendered: i32,
    pub n_spheres: i32,
    pub n_wide_spheres: i32,
    pub walk_steps: u64,
}

/// Device-resident sphere tables and workspace; only ever handled through a pointer.
#[repr(C)]
pub struct dray_scene {
    _private: [u8; 0],
}

/// A render context (scenes, device buffers, staging kept between calls); only ever handled through a pointer.
#[repr(C)]
pub struct dray_context {
    _private: [u8; 0],
}

/// PixelIndexEntry, camera.rs:8-13.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_pixel_index {
    pub pixel_i: i32,
    pub pixel_j: i32,
    pub pixel_sample_num: i32,
}

/// RayColorEntry (camera.rs:15-21) and RayColorStatus (camera.rs:34-38) of one path, plus its stream position.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_path_entry {
    pub attenuation: [f64; 3],
    pub origin: [f64; 3],
    pub direction: [f64; 3],
    pub color: [f64; 3],
    pub depth: i32,
    pub stage: u32,
    pub px: dray_pixel_index,
    pub finished: u32,
    pub hit_object_or_stop: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct dray_grid_info {
    pub lo: [f64; 3],
    pub hi: [f64; 3],
    pub n: [i32; 3],
    pub n_narrow: i32,
    pub n_wide: i32,
    pub n_loose: i32,
    pub n_entries: u64,
    pub max_cell_entries: u32,
    pub _pad: u32,
}

// ---- functions --------------------------------------------------------------------------------------------
/// `dray_present_fn`: called after every pass of `dray_context_render_progressive`, on the calling thread.
pub type dray_present_fn = Option<unsafe extern "C" fn(user: *mut c_void, pass: i32, samples_done: i32)>;

extern "C" {
    pub fn dray_version() -> *const c_char;
    pub fn dray_strerror(status: c_int) -> *const c_char;
    pub fn dray_device_count() -> c_int;
    pub fn dray_opts_default(opts: *mut dray_opts);

    pub fn dray_camera_default(cam: *mut dray_camera);
    pub fn dray_camera_initialize(cam: *const dray_camera, out: *mut dray_camera_derived) -> c_int;

    pub fn dray_render(
        cam: *const dray_camera,
        spheres: *const dray_sphere,
        n_spheres: usize,
        color: *mut u32,
        raw: *mut f64,
        count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        stats: *mut dray_stats,
    ) -> c_int;
    pub fn dray_render_multi(
        cam: *const dray_camera,
        spheres: *const dray_sphere,
        n_spheres: usize,
        color: *mut u32,
        raw: *mut f64,
        count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        devices: *const i32,
        n_devices: i32,
        stats: *mut dray_stats,
    ) -> c_int;

    pub fn dray_release_cache();
    pub fn dray_context_create(devices: *const i32, n_devices: i32, out: *mut *mut dray_context) -> c_int;
    pub fn dray_context_destroy(ctx: *mut dray_context);
    pub fn dray_context_set_scene(ctx: *mut dray_context, spheres: *const dray_sphere, n_spheres: usize) -> c_int;
    pub fn dray_context_render(
        ctx: *mut dray_context,
        cam: *const dray_camera,
        color: *mut u32,
        raw: *mut f64,
        count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        sync: u32,
        stats: *mut dray_stats,
    ) -> c_int;
    pub fn dray_context_render_progressive(
        ctx: *mut dray_context,
        cam: *const dray_camera,
        color: *mut u32,
        raw: *mut f64,
        count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        samples_per_pass: i32,
        present: dray_present_fn,
        user: *mut c_void,
        stats: *mut dray_stats,
    ) -> c_int;
    pub fn dray_context_host_alloc(ctx: *mut dray_context, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn dray_context_pin(ctx: *mut dray_context, host: *mut c_void, bytes: usize) -> c_int;
    pub fn dray_context_unpin(ctx: *mut dray_context);

    pub fn dray_scene_create(spheres: *const dray_sphere, n_spheres: usize, device: i32, out: *mut *mut dray_scene) -> c_int;
    pub fn dray_scene_destroy(scene: *mut dray_scene);
    pub fn dray_render_device(
        cam: *const dray_camera,
        scene: *mut dray_scene,
        d_color: *mut u32,
        d_raw: *mut f64,
        d_count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        stream: *mut c_void,
        stats: *mut dray_stats,
    ) -> c_int;
    pub fn dray_render_device_async(
        cam: *const dray_camera,
        scene: *mut dray_scene,
        d_color: *mut u32,
        d_raw: *mut f64,
        d_count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        stream: *mut c_void,
        ticket: *mut u64,
    ) -> c_int;
    pub fn dray_scene_collect(scene: *mut dray_scene, ticket: u64, stats: *mut dray_stats) -> c_int;
    pub fn dray_render_device_noalloc(
        cam: *const dray_camera,
        scene: *mut dray_scene,
        d_color: *mut u32,
        d_raw: *mut f64,
        d_count: *mut i32,
        buf_len: usize,
        opts: *const dray_opts,
        stream: *mut c_void,
        stats: *mut dray_stats,
    ) -> c_int;

    pub fn dray_camera_rays(
        cam: *const dray_camera,
        px: *const dray_pixel_index,
        n: usize,
        out: *mut dray_path_entry,
        opts: *const dray_opts,
    ) -> c_int;
    pub fn dray_path_step(
        scene: *mut dray_scene,
        image_width: i32,
        entries: *mut dray_path_entry,
        n: usize,
        opts: *const dray_opts,
        stats: *mut dray_stats,
    ) -> c_int;
    pub fn dray_camera_rays_device(
        cam: *const dray_camera,
        d_px: *const dray_pixel_index,
        n: usize,
        d_out: *mut dray_path_entry,
        opts: *const dray_opts,
        stream: *mut c_void,
    ) -> c_int;
    pub fn dray_path_step_device(
        scene: *mut dray_scene,
        image_width: i32,
        d_entries: *mut dray_path_entry,
        n: usize,
        opts: *const dray_opts,
        stream: *mut c_void,
    ) -> c_int;

    pub fn dray_grid_inspect(
        spheres: *const dray_sphere,
        n_spheres: usize,
        info: *mut dray_grid_info,
        cell_begin: *mut u32,
        entry_sphere: *mut i32,
        loose_sphere: *mut i32,
    ) -> c_int;

    pub fn dray_scene_final(seed: u64, out: *mut dray_sphere, cap: usize) -> c_int;
    pub fn dray_scene_recipe(
        seed: u64,
        lo: i32,
        hi: i32,
        p_diffuse: f64,
        p_metal: f64,
        fuzz_max: f64,
        out: *mut dray_sphere,
        cap: usize,
    ) -> c_int;
    pub fn dray_scene_three_spheres(out: *mut dray_sphere, cap: usize) -> c_int;

    pub fn dray_write_ppm(path: *const c_char, color: *const u32, width: i32, height: i32, binary: i32) -> c_int;
}
