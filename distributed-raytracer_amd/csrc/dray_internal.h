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

// Sphere tables in device memory.
struct SceneDev {
    int32_t n = 0;
    // f64 tables (reference-order kernels)
    double* geom64 = nullptr; // n x 4: cx, cy, cz, radius
    double* mat64 = nullptr;  // n x 4: albedo rgb, fuzz or ior
    uint32_t* kind = nullptr; // n
    // f32 tables (throughput and f32-plain kernels)
    float* geom32 = nullptr; // n x 4: cx, cy, cz, radius
    float* mat32 = nullptr;  // n x 4
    // throughput kernel: "narrow" spheres are scanned in f32, "wide" ones (large radius) in f64
    int32_t n_narrow = 0, n_wide = 0;
    float* narrow_geom = nullptr;    // n_narrow_padded x 4: cx, cy, cz, r*r  (padding never hits)
    float* filter_geom = nullptr;    // n_narrow_padded x 4: cx, cy, cz, w = r^2 - |c|^2 + slack (expanded-form filter)
    float* filter_zw = nullptr;      // n_narrow_padded/2 x 4: cz0, w0, cz1, w1 of the filter table
    float* narrow_pairs = nullptr;   // the same table as pairs: (cx0,cx1),(cy0,cy1),(cz0,cz1),(r0^2,r1^2)
    int32_t* narrow_index = nullptr; // position in the caller's list
    // matrix-core filter: tiles of 16 narrow spheres, 64 floats each: cx[16] cy[16] cz[16] w[16], w = r^2-|c|^2+slack,
    // plus one spare tile at the end (the kernel prefetches one tile ahead)
    float* narrow_tiles = nullptr;
    int32_t n_tiles = 0;
    int32_t n_narrow_padded = 0;
    int32_t* wide_index = nullptr; // positions of the wide spheres, ascending
};

struct DeviceCounters {
    unsigned long long segments;
    unsigned long long samples;
    unsigned long long next_item; // work queue head of the throughput kernel
    unsigned long long _pad;
};

// render_ref.hip
hipError_t launch_render_ref_f64(const CamDev<double>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, uint32_t* d_color, double* d_raw, int32_t* d_count,
                                 DeviceCounters* d_counters, hipStream_t stream);
hipError_t launch_render_ref_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, uint32_t* d_color, double* d_raw, int32_t* d_count,
                                 DeviceCounters* d_counters, hipStream_t stream);

// render_f32.hip
struct F32Plan {
    int32_t samples_per_item; // consecutive samples of one pixel that one lane takes at a time
    int32_t n_chunks;         // ceil(spp / samples_per_item)
    size_t accum_words;       // 3 * n_rows * W fixed-point sums (u64), zeroed before the launch
    int32_t grid, block;
    size_t lds_bytes; // dynamic LDS of the launch
    int32_t variant; // scan formulation: 0 one sphere per instruction, 1 two spheres per packed-f32 instruction
};
F32Plan plan_render_f32(const CamDev<float>& cam, const Partition& part, int32_t samples_per_item, int32_t variant,
                        int32_t n_narrow_padded, int device);
hipError_t launch_render_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part, const RngKey& key,
                             const F32Plan& plan, unsigned long long* d_accum, DeviceCounters* d_counters,
                             hipStream_t stream);
hipError_t launch_resolve_f32(const CamDev<float>& cam, const Partition& part, const unsigned long long* d_accum,
                              uint32_t* d_color, double* d_raw, int32_t* d_count, hipStream_t stream);

} // namespace dray
