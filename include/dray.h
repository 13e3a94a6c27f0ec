/*
 * dray.h -- C ABI of libdray_hip.so, the MI355X (gfx950) backend for the path
 * tracer's per-pixel render path.
 *
 * The library replaces the body of the reference's
 *     Camera::render(&mut self, world, window, color_buffer, raw_buffer, count_buffer)
 * (src/raytracer/camera.rs:208-236): camera set-up, the sample loop, ray_color,
 * the sphere-list intersection, the material scatter and the colour write.
 * A Rust `-sys` crate binds exactly these symbols; nothing here uses C++ or
 * torch types.  All functions return DRAY_OK (0) or a negative dray_status;
 * nothing throws or unwinds across this boundary.
 *
 * Randomness: the reference draws from an OS-seeded generator
 * (src/raytracer/prelude.rs:22-25).  This backend draws every uniform from a
 * Philox4x32-10 stream: key = seed, counter = (pixel index j*W+i, sample index,
 * stage, block).  Stage 0 is the camera ray (uniforms 0,1 = pixel offsets,
 * aperture attempt m = uniforms 2+2m, 3+2m); stage k >= 1 is the scatter after
 * the k-th intersection (unit-vector attempt a = uniforms 4a..4a+2, dielectric
 * reflect/refract choice = uniform 0).  Uniform n of a stage is word n%4 of
 * block n/4, mapped to [0,1) as (word >> 8) * 2^-24.  The image therefore
 * depends only on (scene, camera, seed, sample_offset), never on the GPU
 * count, the partition or the scheduling.
 */
#ifndef DRAY_H
#define DRAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dray_status {
    DRAY_OK = 0,
    DRAY_ERR_INVALID_ARG = -1,
    DRAY_ERR_BUFFER_TOO_SMALL = -2,
    DRAY_ERR_UNSUPPORTED = -3,
    DRAY_ERR_NO_DEVICE = -4,
    DRAY_ERR_HIP = -5,
    DRAY_ERR_INTERNAL = -6,
    DRAY_ERR_TOO_LARGE = -7 /* dray_render_device_noalloc: more pixels than DRAY_NOALLOC_MAX_PIXELS */
} dray_status;

/* The ten public camera fields, src/raytracer/camera.rs:42-53, same order. */
typedef struct dray_camera {
    double aspect_ratio;
    int32_t image_width;
    int32_t samples_per_pixel;
    int32_t max_depth;
    int32_t _pad;
    double vfov;
    double lookfrom[3];
    double lookat[3];
    double vup[3];
    double defocus_angle;
    double focus_dist;
} dray_camera;

/* The private camera fields computed by Camera::initialize, camera.rs:55-65,169-206. */
typedef struct dray_camera_derived {
    int32_t image_height;
    int32_t _pad;
    double pixel_samples_scale;
    double center[3];
    double pixel00_loc[3];
    double pixel_delta_u[3];
    double pixel_delta_v[3];
    double u[3], v[3], w[3];
    double defocus_disk_u[3];
    double defocus_disk_v[3];
} dray_camera_derived;

/* Material kinds: material.rs:10-19 (absorbs everything), :33-57, :59-81, :83-123. */
enum { DRAY_MAT_DEFAULT = 0, DRAY_MAT_LAMBERTIAN = 1, DRAY_MAT_METAL = 2, DRAY_MAT_DIELECTRIC = 3,
       DRAY_MAT_TRANSPARENT = 4 /* material.rs:22-31: hands the incoming ray back unchanged, attenuation (1,1,1) */ };

/* One Sphere (sphere.rs:6-10) with its material flattened in.  radius is used as
 * max(radius, 0) (sphere.rs:14); a metal's fuzz as min(fuzz, 1) (material.rs:67).
 * Order in the array = insertion order of HittableList::add (hittable_list.rs:30-32):
 * on equal hit distance the earlier sphere wins (hittable_list.rs:65-70). */
typedef struct dray_sphere {
    double center[3];
    double radius;
    uint32_t kind;
    uint32_t shape;     /* DRAY_SHAPE_SPHERE (0), or DRAY_SHAPE_BOX: the record is a dray_box */
    double albedo[3];   /* Lambertian / Metal */
    double fuzz_or_ior; /* Metal: fuzz; Dialectric: refraction index */
} dray_sphere;

/* A BoundingBox (bounding_box.rs:6-9: three Intervals) as an entry of the hittable list: a record of the size of a
 * dray_sphere with `shape` at the same place, passed in the same array at its place in the insertion order.  It is
 * intersected as bounding_box.rs:50-76 is written -- only axes 0 and 1 are looked at (:53) and a plane's distance is
 * `min - origin / direction` (:55-56) -- and its material is always Transparent (:65,71), which hands the incoming ray
 * back: a path that meets a box first walks through the depth it has left and ends black.  The reference-order modes
 * do exactly that; DRAY_MODE_F32 renders the sample black at once (as it does for Transparent spheres). */
enum { DRAY_SHAPE_SPHERE = 0, DRAY_SHAPE_BOX = 1 };
typedef struct dray_box {
    double min[3];         /* axes[a].min */
    double _unused0;
    uint32_t _unused_kind;
    uint32_t shape;        /* DRAY_SHAPE_BOX */
    double max[3];         /* axes[a].max */
    double _unused1;
} dray_box;

/* Arithmetic of the render. */
enum {
    DRAY_MODE_F64 = 0,      /* f64, operation order of the reference: the parity mode */
    DRAY_MODE_F32 = 1,      /* f32 throughput kernel (large spheres are intersected in f64) */
    DRAY_MODE_F32_PLAIN = 2 /* f32, operation order of the reference, no restructuring (A/B baseline) */
};

enum {
    /* f64 / f32-plain: multiply attenuations left to right as ray_color_iteration does
     * (camera.rs:118-144) instead of right to left as the recursion does (camera.rs:264-282). */
    DRAY_FLAG_ITERATIVE_PRODUCT = 1u,
    /* Rendered rows are stored back to back: row j goes to slab row (j-row_begin)/row_step.
     * Without it row j is stored at j*image_width as the reference does (colors.rs:37). */
    DRAY_FLAG_COMPACT_ROWS = 2u,
    /* f32 mode: switch the wave-level compaction of terminated paths off.  By default a lane whose path has ended
     * takes the next sample of its wave's pool at once, so all 64 lanes stay busy; with this flag it idles until
     * every path of its wave has ended and the wave refills as a whole (64 samples in lock-step, as a kernel
     * without compaction runs them).  The image is the same; only the time differs (an A/B switch). */
    DRAY_FLAG_NO_REFILL = 4u
};

/* Sphere-scan formulations of the f32 kernel; all give identical images.
 * DRAY_MODE_F64 reads the same field as the way its kernel is organised, again with identical results (bit for bit):
 * DRAY_SCAN_SCALAR = one thread per pixel, every sphere of the list tested per ray as hittable_list.rs:65-70 does;
 * DRAY_SCAN_HYBRID = one thread per pixel, closest hit through the grid walk; DRAY_SCAN_AUTO / DRAY_SCAN_GRID = the
 * grid walk with the frame run as a stream of samples over persistent waves, a pixel's samples still added in order
 * (the fast one; it keeps a ring of finished samples per wave, some 200 MB per scene handle, allocated on first use). */
enum {
    DRAY_SCAN_AUTO = 0,   /* grid for scenes of 40 spheres or more, else hybrid */
    DRAY_SCAN_SCALAR = 1, /* exact test on every sphere, geometry through scalar loads */
    DRAY_SCAN_HYBRID = 2, /* expanded-form filter (geometry from LDS + scalar loads) in front of the exact test;
                             DRAY_ERR_UNSUPPORTED when the sphere table does not fit in LDS (> 7168 spheres) */
    DRAY_SCAN_GRID = 3    /* every lane walks its ray through a uniform grid over the spheres and applies the exact
                             test to the spheres of the cells it crosses (from a copy of the grid in LDS when six
                             workgroups' copies fit a compute unit, from memory otherwise: the library decides) */
};

typedef struct dray_opts {
    uint64_t seed;
    uint32_t mode;
    uint32_t flags;
    uint32_t sample_offset;   /* index of the first sample's stream; a later call continuing the
                                 same buffers passes the number of samples already accumulated */
    int32_t device;           /* HIP device ordinal, -1 = the current device */
    int32_t row_begin;        /* scanline partition: rows row_begin, row_begin+row_step, ... */
    int32_t row_step;         /* 0 is read as 1 */
    int32_t pixels_per_pool;  /* f32 mode: consecutive pixels a wave draws from the work queue at a time,
                                 0 = library default; the image does not depend on it */
    int32_t variant;          /* f32 mode: sphere-scan formulation, a DRAY_SCAN_* value */
} dray_opts;

typedef struct dray_stats {
    uint64_t segments;  /* path segments = world.hit calls (camera.rs:270) */
    uint64_t samples;   /* get_ray + ray_color pairs (camera.rs:224-225) */
    double kernel_ms;   /* trace kernel, HIP events on the launch stream */
    double resolve_ms;  /* accumulate / gamma / pack kernel */
    double total_ms;    /* whole call, host clock */
    int32_t image_height;
    int32_t rows_rendered;
    int32_t n_spheres;
    int32_t n_wide_spheres; /* spheres routed to the f64 intersection in DRAY_MODE_F32 */
    uint64_t walk_steps;    /* grid scan: steps summed over waves (a step moves every walking lane of a wave on by
                               one cell or one sphere); 0 for the other scans */
} dray_stats;

typedef struct dray_scene dray_scene; /* device-resident sphere tables + workspace */

/* ---- housekeeping --------------------------------------------------------- */
const char* dray_version(void);
const char* dray_strerror(int status);
/* Number of HIP devices, or a negative dray_status. */
int dray_device_count(void);
void dray_opts_default(dray_opts* opts);

/* ---- host-side camera maths (no GPU needed) ------------------------------ */
/* Camera::new, camera.rs:147-163. */
void dray_camera_default(dray_camera* cam);
/* Camera::initialize, camera.rs:169-206, in f64 on the host. */
int dray_camera_initialize(const dray_camera* cam, dray_camera_derived* out);

/* ---- the drop-in render call --------------------------------------------- */
/* Camera::render, camera.rs:208-236, with host buffers as the reference has them:
 *   color  len >= buf_len, ARGB 0xFFrrggbb at j*W+i             (colors.rs:42-43)
 *   raw    3 f64 per pixel, running colour sum, added to        (colors.rs:38); may be NULL
 *   count  running sample count per pixel, added to             (colors.rs:39); may be NULL
 * buf_len is the length of each buffer in pixels and must be >= W*image_height; entries past
 * the rendered rows are left untouched (the reference's caller allocates W*W, main.rs:69-73).
 * Blocking.  The window refresh of colors.rs:44 is the caller's business. */
int dray_render(const dray_camera* cam, const dray_sphere* spheres, size_t n_spheres, uint32_t* color, double* raw,
                int32_t* count, size_t buf_len, const dray_opts* opts, dray_stats* stats);

/* The same call over several GPUs of one node from one process, as the single render call of main.rs:86 needs it
 * (the reference spreads a frame over machines through src/distributed; orchestrator_server.rs:59-68 deals the
 * rays out, :253-256 collects the pixels): entry k of `devices` renders scanlines k, k + n_devices, ... with the
 * whole scene replicated, every share enqueued from the calling thread (no host threads are started), and its rows
 * go straight into the caller's buffers.
 * Buffers, result and statistics are those of dray_render (kernel_ms: the slowest share).  devices == NULL means
 * devices 0 .. n_devices-1; n_devices <= 0 means every visible GPU; a device may be named more than once.
 * opts->device, row_begin, row_step and DRAY_FLAG_COMPACT_ROWS must be left at their defaults. */
int dray_render_multi(const dray_camera* cam, const dray_sphere* spheres, size_t n_spheres, uint32_t* color, double* raw,
                      int32_t* count, size_t buf_len, const dray_opts* opts, const int32_t* devices, int32_t n_devices,
                      dray_stats* stats);

/* dray_render and dray_render_multi keep what a second call can use again -- the scene tables and the grid built from
 * the sphere list (reused when the next call's list is byte for byte the same), device buffers, pinned staging
 * memory, streams -- in a cache per list of devices.  This frees all of it (no call may be in flight). */
void dray_release_cache(void);

/* ---- a render context: the one-call route with its set-up kept between calls ------------------------------- */
/* What dray_render_multi does per call, split so that a caller who renders the same world again and again -- the
 * passes of a progressive display, the frames of a camera move -- pays for it once: dray_context_create takes the
 * devices (entry k renders scanlines k, k + n_devices, ...; NULL / 0 = every visible GPU), dray_context_set_scene
 * builds the sphere tables and the grid on each of them, dray_context_render is Camera::render (camera.rs:208-236)
 * into the caller's host buffers with the contract of dray_render (opts->device, row_begin, row_step and
 * DRAY_FLAG_COMPACT_ROWS left at their defaults).  No counterpart in the reference, whose render call is made once.
 *
 * `sync` says what happens to the running sums (raw, count), which live in device memory between calls:
 *   DRAY_SYNC_UPLOAD_SUMS    the device copy is first loaded from the caller's raw / count (what dray_render does);
 *                            without it the render adds to what the context's previous call left on the device, which
 *                            must have been a call for the same frame size (else DRAY_ERR_INVALID_ARG)
 *   DRAY_SYNC_DOWNLOAD_SUMS  raw / count are written back to the caller's buffers; without it only `color` is
 * so a progressive render is: first pass UPLOAD, passes in between 0, last pass DOWNLOAD, and each pass moves only
 * the 4 bytes per pixel of the colour buffer.  dray_render = UPLOAD | DOWNLOAD in one call.
 *
 * Buffers in page-locked host memory are read and written by the copy engines directly, rows and all; others are
 * staged through the context's own pinned block by the CPU (two copies of every byte more).  dray_context_host_alloc
 * hands out zeroed page-locked memory that lives until dray_context_destroy; dray_context_pin registers memory the
 * caller already has (hipHostRegister; it must stay allocated until dray_context_unpin or dray_context_destroy, and
 * the call returns DRAY_ERR_HIP where the driver refuses such registrations -- the render then stages as usual).
 * A context serves one call at a time (calls from several threads take turns). */
typedef struct dray_context dray_context;
enum { DRAY_SYNC_UPLOAD_SUMS = 1u, DRAY_SYNC_DOWNLOAD_SUMS = 2u };
int dray_context_create(const int32_t* devices, int32_t n_devices, dray_context** out);
void dray_context_destroy(dray_context* ctx);
int dray_context_set_scene(dray_context* ctx, const dray_sphere* spheres, size_t n_spheres);
int dray_context_render(dray_context* ctx, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                        size_t buf_len, const dray_opts* opts, uint32_t sync, dray_stats* stats);
/* A frame's samples in passes with the picture handed out after each: what Camera::render does when it presents the
 * window while it renders (colors.rs:44 presents after every sample of every pixel; here after every pass of
 * `samples_per_pass` samples per pixel, the last pass taking what is left of cam->samples_per_pixel).  The running sums
 * go up before the first pass and come back after the last, as with dray_context_render(UPLOAD | DOWNLOAD) in one call,
 * and the three buffers end up bit for bit as after that call; opts->sample_offset is advanced pass by pass.  After
 * each pass, once its colours are in `color`, `present(user, pass, samples_done)` is called on the calling thread (NULL:
 * no call); the next pass is already rendering by then and the one after it is enqueued (three passes are under way
 * at a time), so what `present` takes does not hold the GPU up.  `color` must not be touched outside `present` while
 * the call runs, and `present` must not call into the same context (the call holds it).  stats: segments, samples
 * and kernel times added up over the passes.  No counterpart in the reference beyond that per-sample present. */
typedef void (*dray_present_fn)(void* user, int32_t pass, int32_t samples_done);
int dray_context_render_progressive(dray_context* ctx, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                                    size_t buf_len, const dray_opts* opts, int32_t samples_per_pass, dray_present_fn present,
                                    void* user, dray_stats* stats);
int dray_context_host_alloc(dray_context* ctx, size_t bytes, void** out);
int dray_context_pin(dray_context* ctx, void* host, size_t bytes);
void dray_context_unpin(dray_context* ctx);

/* ---- device-resident variant (buffers already in HBM) --------------------- */
/* Upload the sphere list of a HittableList (hittable_list.rs:7-9) to `device`. */
int dray_scene_create(const dray_sphere* spheres, size_t n_spheres, int32_t device, dray_scene** out);
void dray_scene_destroy(dray_scene* scene);
/* Same contract as dray_render but color/raw/count are device pointers on the scene's device
 * and the work is enqueued on `stream` (a hipStream_t, NULL = default stream).  With stats == NULL
 * the call returns once the work is enqueued; with stats it waits for completion and fills them.
 * The calls on one scene take turns: calls from several host threads are enqueued one after the other, and a call
 * on another stream than the scene's previous call first makes its stream wait for that call to finish on the GPU, so
 * the calls of a scene reach the caller's buffers in the order they were made.  (In DRAY_MODE_F32 the trace kernel of
 * a call, which reads the scene alone, runs on a stream of the scene's own and may begin while the previous call's
 * kernel is still running out; `stream` waits for it before anything touches the buffers.)  For renders that should
 * run side by side, create one scene per stream. */
int dray_render_device(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw, int32_t* d_count,
                       size_t buf_len, const dray_opts* opts, void* stream, dray_stats* stats);

/* The same call for a caller that enqueues several renders before it looks at any of them (the passes of a progressive
 * render, the steps of a benchmark): returns once the work is enqueued and hands out a ticket (0 when there was
 * nothing to render); dray_scene_collect waits for that call and fills in its statistics.  A scene remembers its last
 * 16 calls: an older ticket is DRAY_ERR_INVALID_ARG.  No counterpart in the reference, whose render call is synchronous. */
int dray_render_device_async(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw, int32_t* d_count,
                             size_t buf_len, const dray_opts* opts, void* stream, uint64_t* ticket);
int dray_scene_collect(dray_scene* scene, uint64_t ticket, dray_stats* stats);

/* dray_render_device for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  Same arguments, same results
 * bit for bit, but the call's workspace (the f32 kernel's fixed-point colour sums, 24 bytes per pixel, and the
 * counters) is a static array in the library's device code instead of buffers the scene allocates and grows: the
 * call never calls hipMalloc or hipFree.  Limits: the rows it renders times image_width may not exceed
 * DRAY_NOALLOC_MAX_PIXELS (DRAY_ERR_TOO_LARGE otherwise); every GPU the library runs on holds its own 25 MiB copy
 * of the array until the process ends; all such calls on one GPU share that GPU's copy, so two of them that
 * overlap in time (different streams or host threads, the same scene or not) race -- keep them on one stream.  With
 * stats the call blocks until the render is done.  Not for a stream that is being captured into a graph: like every
 * call on a scene it records the scene's events on `stream`, waits on the previous call's, and may synchronise on the
 * event of the call sixteen calls back. */
#define DRAY_NOALLOC_MAX_PIXELS 1048576
int dray_render_device_noalloc(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw,
                               int32_t* d_count, size_t buf_len, const dray_opts* opts, void* stream, dray_stats* stats);

/* ---- the path state the reference's distributed layer feeds on ---------------------------- */
/* PixelIndexEntry, camera.rs:8-13. */
typedef struct dray_pixel_index {
    int32_t pixel_i, pixel_j, pixel_sample_num;
} dray_pixel_index;

/* RayColorEntry (camera.rs:15-21) and RayColorStatus (camera.rs:34-38) of one path, plus what stands in for the
 * reference's global generator: the path's pixel index and the Philox stage its next scatter draws from. */
typedef struct dray_path_entry {
    double attenuation[3];
    double origin[3], direction[3]; /* ray */
    double color[3];
    int32_t depth;
    uint32_t stage; /* 1 for a fresh camera ray; +1 per scatter */
    dray_pixel_index px;
    uint32_t finished, hit_object_or_stop;
    uint32_t _pad;
} dray_path_entry;

/* CameraRayIterator::next (camera.rs:103-114) for n pixel indices in the caller's order (the reference walks a
 * shuffled order, camera.rs:96-98), each followed by RayColorEntry::new(ray, max_depth) (camera.rs:23-31).
 * f64 arithmetic in the reference's operation order.  Uses opts->seed, sample_offset and device. */
int dray_camera_rays(const dray_camera* cam, const dray_pixel_index* px, size_t n, dray_path_entry* out,
                     const dray_opts* opts);
/* One ray_color_iteration (camera.rs:118-144) on each of n entries, in place: a path is advanced by one segment
 * and can be resumed by the next call, on this or another GPU, in any batch and order.  Entries whose `finished` is
 * set are left alone.  image_width is the camera's (it enters the stream key).  f64, the reference's operation
 * order; with opts->mode = DRAY_MODE_F32 the segment is intersected and scattered at the throughput kernel's
 * precision instead (the entry stays f64; same streams; a list with bounding boxes takes the f64 step either way).
 * With stats, segments counts the intersected entries. */
int dray_path_step(dray_scene* scene, int32_t image_width, dray_path_entry* entries, size_t n, const dray_opts* opts,
                   dray_stats* stats);
/* The same two calls on device pointers (on the scene's device / opts->device), enqueued on `stream`. */
int dray_camera_rays_device(const dray_camera* cam, const dray_pixel_index* d_px, size_t n, dray_path_entry* d_out,
                            const dray_opts* opts, void* stream);
int dray_path_step_device(dray_scene* scene, int32_t image_width, dray_path_entry* d_entries, size_t n,
                          const dray_opts* opts, void* stream);

/* ---- the grid of the grid scan, as the host builds it (no GPU needed; for tools and tests) ---------------- */
typedef struct dray_grid_info {
    double lo[3], hi[3];       /* the grid's box */
    int32_t n[3];              /* cells per axis (all 0: no sphere is walked) */
    int32_t n_narrow, n_wide;  /* spheres scanned in f32 / intersected in f64 (radius >= 32) */
    int32_t n_loose;           /* narrow spheres kept out of the grid (more than 512 cells, or not finite) */
    uint64_t n_entries;        /* sphere listings over all cells */
    uint32_t max_cell_entries;
    uint32_t _pad;
} dray_grid_info;
/* Builds the uniform grid dray_scene_create builds for this list and reports it.  Call with the three arrays NULL
 * to get the sizes, then with cell_begin[n[0]*n[1]*n[2] + 1] (cell (x, y, z) at (z*n[1] + y)*n[0] + x lists
 * entry_sphere[cell_begin[c] .. cell_begin[c+1])), entry_sphere[n_entries] and loose_sphere[n_loose], which
 * receive positions in the caller's list. */
int dray_grid_inspect(const dray_sphere* spheres, size_t n_spheres, dray_grid_info* info, uint32_t* cell_begin,
                      int32_t* entry_sphere, int32_t* loose_sphere);

/* ---- scene recipes (host) ------------------------------------------------- */
/* The scene built by the reference's local binary, main.rs:13-50, with its uniforms taken from the
 * Philox stream (pixel = sample = stage = 0xFFFFFFFF) of `seed`.  Returns the sphere count or a
 * negative dray_status (cap too small). */
int dray_scene_final(uint64_t seed, dray_sphere* out, size_t cap);
/* The same recipe over cells a, b in [lo, hi) with other material shares: a cell is diffuse below p_diffuse, metal
 * (fuzz in [0, fuzz_max)) below p_diffuse + p_metal, glass otherwise.  dray_scene_final = (-11, 11, 0.8, 0.15, 0.5). */
int dray_scene_recipe(uint64_t seed, int32_t lo, int32_t hi, double p_diffuse, double p_metal, double fuzz_max,
                      dray_sphere* out, size_t cap);
/* Ground + Lambertian + Dialectric + Metal test scene (4 spheres). */
int dray_scene_three_spheres(dray_sphere* out, size_t cap);

/* ---- image output (host) -------------------------------------------------- */
/* Writes rows [0, height) of a colour buffer (0xFFrrggbb at j*width+i, colors.rs:42) as a PPM file: plain "P3"
 * with one "r g b" line per pixel when binary == 0, "P6" when binary != 0; maxval 255.  The reference's local
 * binary creates img.ppm and a writer for it (main.rs:8,52) but never writes to it; this is that missing step. */
int dray_write_ppm(const char* path, const uint32_t* color, int32_t width, int32_t height, int32_t binary);

#ifdef __cplusplus
}
#endif
#endif
