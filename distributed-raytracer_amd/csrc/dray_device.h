// Device-side building blocks shared by the render kernels: the Philox4x32-10
// stream, the uniform mapping and the colour quantiser.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dray {

// One Philox4x32-10 block (Salmon et al., SC'11).  Stands in for rand::random
// (reference src/raytracer/prelude.rs:22-25), which is OS-seeded.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) yields the high and the low word together
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// U[0,1) with 24 significant bits: the same value in f32 and in f64.
template <typename T>
__device__ __forceinline__ T u01(uint32_t word) {
    return (T)(word >> 8) * (T)(1.0 / 16777216.0);
}

// The uniforms of one (pixel, sample, stage) stream, fetched a block of four at a time.
struct Stream {
    uint32_t k0, k1, pixel, sample, stage;
    __device__ __forceinline__ void block(uint32_t b, uint32_t (&w)[4]) const {
        philox4x32_10(pixel, sample, stage, b, k0, k1, w);
    }
};

// colors.rs:48-55 then :19: gamma 2, clamp to [0, 0.999], scale by 255.999, truncate.
// Evaluated in f64 from an f64 mean so that every mode quantises the same way.
__device__ __forceinline__ uint32_t to_byte(double mean) {
    double g = mean > 0. ? sqrt(mean) : 0.; // NaN > 0 is false
    if (g < 0.000) g = 0.000;
    if (g > 0.999) g = 0.999;
    const double v = 255.999 * g;
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t pack_argb(double r, double g, double b) { // colors.rs:42
    return (255u << 24) | (to_byte(r) << 16) | (to_byte(g) << 8) | to_byte(b);
}

} // namespace dray
