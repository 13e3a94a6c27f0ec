// Host-side declarations shared by the API translation unit and the kernel
// translation units of libdray_hip.so.  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dray.h"

namespace dray {

constexpr int kExactMaxDepth = 256; // deepest path whose attenuations are kept for the right-to-left product

// Camera constants as the kernels see them (derived on the host in f64, camera.rs:169-206).
template <typename T>
struct CamDev {
    T center[3], p00[3], du[3], dv[3], ddu[3], ddv[3];
    int32_t W, H, spp, max_depth;
    int32_t defocus; // defocus_angle > 0 (camera.rs:247)
};

// What one launch renders: rows row_begin, row_begin+row_step, ... (n_rows of them).
struct Partition {
    int32_t row_begin, row_step, n_rows;
    int32_t compact; // slab row index instead of image row index in the output address
};

struct RngKey {
    uint32_t k0, k1, sample_offset;
};

// What the f32 throughput kernel reads (device pointers, handed to the kernel one by one).
// "Narrow" spheres are scanned in f32, "wide" ones (large radius) are intersected in f64.
struct F32Tables {
    const float4* narrow_geom = nullptr;  // n_narrow_padded: cx, cy, cz, r*r, insertion order; padding never hits
    const float4* filter_xy = nullptr;    // n_narrow_padded/2: cx0, cy0, cx1, cy1 of two consecutive narrow spheres
    const float4* filter_zn = nullptr;    // n_narrow_padded/2: cz0, nw0, cz1, nw1; nw = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2)
    const int32_t* narrow_index = nullptr; // position of a narrow sphere in the caller's list
    const int32_t* wide_index = nullptr;   // positions of the wide spheres, ascending
    const double* geom64 = nullptr;        // n x 4: cx, cy, cz, radius
    const double* wide_geom64 = nullptr;   // n_wide x 4: the wide spheres' rows of geom64, one after the other
    // n_wide x 2: per wide sphere (cx, cy, cz, 1 / r) and (cy + r, cy - r, axial, 0) in f32.  axial = 1: the centre lies so
    // close to the y axis that |oc|^2 - r^2 = ocx^2 + ocz^2 + (cy + r - oy)(cy - r - oy) is formed in f32 without
    // cancellation (the ground of src/main.rs:14); 0: the sphere is intersected in f64
    const float4* wide_axial = nullptr;
    const float4* geom32 = nullptr;        // n: cx, cy, cz, radius
    const float4* mat32 = nullptr;         // n: albedo rgb, fuzz or refraction index
    const uint32_t* kind = nullptr;        // n: material kind
    int32_t n_narrow_padded = 0;
    int32_t n_wide = 0;
    // a ball (centre, radius with 1 % to spare) that holds every finite narrow sphere; radius inf: none to hold
    float bound[4] = {0.f, 0.f, 0.f, INFINITY};
};

// Uniform grid over the narrow spheres, walked by the f32 kernel's grid scan.  A sphere is entered in every cell
// its slightly inflated ball overlaps; spheres that would span too many cells stay outside ("loose") and are
// tested against every ray.
struct GridDev {
    float lo[3], hi[3];  // the grid's box (holds every listed sphere, inflated)
    float cell[3];       // cell size per axis
    float inv_cell[3];
    int32_t n[3];        // cells per axis; n[0] == 0: no grid (no narrow spheres)
    int32_t n_loose;
    const uint2* cells = nullptr;       // per cell (x fastest): first entry, entry count
    const float4* entry_geom = nullptr; // per entry: cx, cy, cz, r*r (the sphere's narrow_geom row)
    const int32_t* entry_pos = nullptr; // per entry: the sphere's position in the narrow table; ascending in a cell
    const int32_t* loose = nullptr;     // narrow positions of the loose spheres, ascending
    // the same lists as the f32 grid scan reads them: a test costs one load, and a hit names the list entry at once
    const int32_t* entry_index = nullptr;  // per entry: the sphere's position in the caller's list
    const float4* loose_geom = nullptr;    // the loose spheres' exact-table rows, padded to a multiple of four with
    const int32_t* loose_index = nullptr;  // ... rows no ray hits; their positions in the caller's list
    int32_t n_loose_padded = 0;
};

// Sphere tables in device memory (owned by the scene).
struct SceneDev {
    int32_t n = 0;
    // f64 tables (reference-order kernels)
    double* geom64 = nullptr; // n x 4: cx, cy, cz, radius
    double* mat64 = nullptr;  // n x 4: albedo rgb, fuzz or ior (a Dialectric's row: 1 / ior, r0 front, r0 back, ior)
    uint32_t* kind = nullptr; // n
    uint32_t* kind32 = nullptr; // n: the kinds as the throughput kernel takes them (see scene_create_impl)
    // f32 tables (throughput and f32-plain kernels)
    float* geom32 = nullptr; // n x 4: cx, cy, cz, radius
    float* mat32 = nullptr;  // n x 4
    int32_t n_narrow = 0, n_wide = 0, n_narrow_padded = 0;
    float* narrow_geom = nullptr;
    float* filter_xy = nullptr;
    float* filter_zn = nullptr;
    int32_t* narrow_index = nullptr;
    int32_t* wide_index = nullptr;
    F32Tables f32; // the pointers above, as the f32 kernel takes them
    // list entries that are bounding boxes (bounding_box.rs): min xyz, max xyz per box, and their list positions.
    // In the sphere tables such an entry is a sphere no ray can hit (NaN centre), its kind is Transparent.
    int32_t n_boxes = 0;
    double* box64 = nullptr;      // n_boxes x 6
    int32_t* box_index = nullptr; // n_boxes, ascending
    // grid scan
    uint2* grid_cells = nullptr;
    float* grid_entry_geom = nullptr;
    int32_t* grid_entry_pos = nullptr;
    int32_t* grid_loose = nullptr;
    float* loose_geom = nullptr;
    int32_t* loose_index = nullptr;
    double* wide_geom64 = nullptr;
    float* wide_axial = nullptr;
    size_t grid_entries = 0;
    uint32_t grid_max_cell_entries = 0; // the longest cell list
    GridDev grid{};
    // the same lists for the f64 stream kernel, with the f64 geometry beside them so that a test costs one load:
    double* entry_geom64 = nullptr;   // per grid entry: cx, cy, cz, radius
    int32_t* entry_index = nullptr;   // per grid entry: the sphere's position in the caller's list
    double* always_geom64 = nullptr;  // the spheres every ray is tested against (wide, then loose): cx, cy, cz, radius
    int32_t* always_index = nullptr;  // their positions in the caller's list
    int32_t n_always = 0;
};

// A workgroup of the f32 kernel may hold 64 KiB of LDS: its static arrays (pixel sums, tags, draw requests) plus,
// for the hybrid scan, the staged cx,cy table (8 B per sphere).  Larger tables take another scan.
constexpr size_t kF32StaticLds = 8 * 1024;                       // budget of the static arrays (asserted in the kernel)
constexpr size_t kHybridLdsLimit = 64 * 1024 - kF32StaticLds;    // largest table: 7 168 narrow spheres
// The grid scan's compact copy of the grid in LDS: six workgroups are resident on a compute unit (160 KiB of LDS).
constexpr size_t kGridLdsLimit = 160 * 1024 / 6 - kF32StaticLds - 512;

struct DeviceCounters {
    unsigned long long segments;
    unsigned long long samples;
    unsigned long long next_item; // work queue head of the throughput kernel
    unsigned long long walk_steps; // grid scan: wave-level steps
};

// render_ref.hip
// How the f64 render finds a ray's closest hit and how it runs its lanes; the results are the same bit for bit.
enum {
    kRefWholeList = 0,   // every sphere of the list, one thread per pixel on its own (DRAY_SCAN_SCALAR)
    kRefGridByPixel = 1, // walk through the scene's uniform grid, one thread per pixel on its own (DRAY_SCAN_HYBRID)
    kRefGridAsStream = 2 // the same walk, the frame run as a stream of samples (DRAY_SCAN_AUTO, DRAY_SCAN_GRID)
};
hipError_t launch_render_ref_f64(const CamDev<double>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, int closest_hit, uint32_t* d_color, double* d_raw,
                                 int32_t* d_count, DeviceCounters* d_counters, hipStream_t stream);
// The f64 render as a stream of samples: one persistent launch whose waves keep finished samples in rings of
// kF64RingEntries 32-byte entries each until they join their pixel's sum in sample order.
#ifndef DRAY_F64_RING
#define DRAY_F64_RING 2048
#endif
constexpr uint32_t kF64RingEntries = DRAY_F64_RING;
struct F64StreamPlan {
    int32_t grid, block;   // resident blocks of the persistent launch
    size_t ring_bytes;     // one ring per wave of the launch
    uint32_t chunk_pixels; // consecutive pixels a wave takes from the queue at a time
    size_t lds_bytes;      // dynamic LDS of the launch: the staged copy of the grid (0: the tables are read from memory)
};
// false: this frame or scene is not for the stream kernel (no grid over the scene, no samples)
bool plan_render_f64_stream(const CamDev<double>& cam, const Partition& part, const SceneDev& scene, int device,
                            F64StreamPlan* plan);
// `rings`: plan.ring_bytes of device memory (its contents do not matter)
hipError_t launch_render_f64_stream(const CamDev<double>& cam, const SceneDev& scene, const Partition& part,
                                    const RngKey& key, uint32_t flags, const F64StreamPlan& plan, uint32_t* d_color,
                                    double* d_raw, int32_t* d_count, DeviceCounters* d_counters, void* rings,
                                    hipStream_t stream);
hipError_t launch_render_ref_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, uint32_t* d_color, double* d_raw, int32_t* d_count,
                                 DeviceCounters* d_counters, hipStream_t stream);

hipError_t launch_camera_rays(const CamDev<double>& cam, const RngKey& key, const dray_pixel_index* d_px, size_t n,
                              dray_path_entry* d_out, hipStream_t stream);
hipError_t launch_path_step(const SceneDev& scene, const RngKey& key, int32_t image_width, bool use_grid,
                            dray_path_entry* d_entries, size_t n, DeviceCounters* d_counters, hipStream_t stream);
hipError_t launch_path_step_f32(const SceneDev& scene, const RngKey& key, int32_t image_width, dray_path_entry* d_entries,
                                size_t n, DeviceCounters* d_counters, hipStream_t stream);

// render_f32.hip
struct F32Plan {
    int32_t pixels_per_pool;  // consecutive pixels a wave hands out to its lanes at a time
    int32_t grab_pixels;      // pixels a wave takes from the global queue per atomic
    size_t accum_words;       // 3 * n_rows * W fixed-point sums (u64), zeroed before the launch
    int32_t grid, block;
    size_t lds_bytes;         // dynamic LDS of the launch
    int32_t scan;             // the DRAY_SCAN_* formulation the launch uses (never DRAY_SCAN_AUTO)
    int32_t no_refill;        // DRAY_FLAG_NO_REFILL: idle lanes wait until their whole wave is idle
    int32_t lds_grid;         // grid scan: the launch's workgroups stage a compact copy of the grid in LDS
    int32_t lds_occupancy;    // ... and so many of them fit a compute unit
};
F32Plan plan_render_f32(const CamDev<float>& cam, const Partition& part, int32_t pixels_per_pool, int32_t variant,
                        const SceneDev& scene, int device);
hipError_t launch_render_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part, const RngKey& key,
                             const F32Plan& plan, unsigned long long* d_accum, DeviceCounters* d_counters,
                             hipStream_t stream);
hipError_t launch_resolve_f32(const CamDev<float>& cam, const Partition& part, const unsigned long long* d_accum,
                              uint32_t* d_color, double* d_raw, int32_t* d_count, hipStream_t stream);

} // namespace dray
