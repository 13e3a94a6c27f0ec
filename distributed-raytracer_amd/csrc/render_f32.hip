// f32 throughput render kernel for gfx950.
//
// One persistent launch.  Every lane carries one path; the recursion of the
// reference's ray_color (src/raytracer/camera.rs:264-282) is the iterative
// bounce loop of its twin ray_color_iteration (camera.rs:118-144) with an
// explicit throughput.  A wave draws whole pixels from a global queue and hands
// their samples to its lanes one at a time: when a lane's path ends it takes
// the pool's next sample, so all 64 lanes stay busy whatever their pixel,
// sample or bounce.
//
// Sample colours are summed as 36.28 fixed-point integers (integer addition is
// associative), so the image does not depend on how samples are grouped, on
// the partition, the GPU count or the order in which lanes finish.
//
// The closest hit is the one the reference's scan of the list in insertion
// order finds (hittable_list.rs:65-70: smallest t, on equal t the sphere
// inserted first), whichever of the three scans below looks for it: the test
// of every sphere, the same behind a cheap filter, or a walk through a uniform
// grid.  "Narrow" spheres are tested in f32; "wide" spheres, whose radius is
// so large that |oc|^2 - r^2 cancels catastrophically in f32 (the 1000-radius
// ground of src/main.rs:14), are tested in f32 in a form that does not cancel
// when their centre lies by the y axis, as every such ground's does, and in
// f64 otherwise.
//
// What bounds the scan on this chip (tools/ubench_operand.hip, tools/ubench_valu.hip): a VALU operation whose
// operands are all VGPRs costs ~2.2 cycles of a SIMD, one with an SGPR (or DPP) operand ~3.6 cycles of the
// SIMD's scalar-operand path, and the two overlap; packed f32 halves the instruction count but not the ALU
// cycles; f32 matrix instructions run on the same ALUs.  So the scan minimises VGPR-only ALU operations per
// ray-sphere test and keeps the SGPR-sourced ones under ~60 % of the mix (scan_hybrid below: 8 operations, 3
// of them SGPR-sourced, against 11 and 4 for the direct form of sphere.rs:29-36).
//
// Two more things the main loop is written around (tools/ubench_cost.hip): a compare or a select costs a SIMD 1.8 ns
// where an addition costs 1.05 -- so bounds are arranged for NaN to fail a test by itself, roots are ordered with
// min / max, whole-number coordinates are range-checked through their bits, and nothing is selected that the next
// instruction does not need; and a loop that changes per-lane values under a condition carries a copy of each of them
// round its edge -- so the case such a loop nearly always sees (one refill turn, one or two draw rounds, one wide
// sphere) is written out in front of it.
#include <cstdlib>
#include <mutex>

#include "dray_device.h"
#include "dray_internal.h"

namespace dray {
namespace {

constexpr float kFixScale = 268435456.f;      // 2^28
constexpr double kFixInv = 1.0 / 268435456.0; // 2^-28
constexpr float kFixMax = 15.99f;             // a sample colour channel is clamped to [0, 15.99]
constexpr int kUnroll = 8;                    // spheres per scan block (the tables are padded to a multiple)
constexpr int kPoolSamples = 64;               // samples a pool holds at least (one per lane)
constexpr int kGrabWork = 16384;               // sphere tests' worth of samples a wave aims to take from the queue per atomic
constexpr float kSlack = 7.62939453125e-6f;   // 2^-17, see scan_hybrid
constexpr int kSlots = 32;                     // pixel sums a wave keeps in LDS before they go to memory
constexpr uint32_t kNoPixel = 0xFFFFFFFFu;
// Grid scan: lanes that must still be walking for the wave to keep stepping.  A turn of the main loop costs the same
// whatever the number of lanes it shades, so the fewer lanes a turn leaves walking, the fewer turns a frame takes; but a
// step costs the same whatever the number of lanes it moves.  Headline frame (ms), tables read from memory: 8 -> 26.50,
// 12 -> 26.56, 16 -> 26.76, 24 -> 27.8 (the 10 001-sphere field at 3840x2160x64: 43.12, 42.90, 42.97); the grid staged
// in LDS: 6 -> 26.46, 8 -> 26.44, 12 -> 26.41, 16 -> 26.61 on one box, 8 -> 26.27 on another.
#ifndef DRAY_WALK_QUORUM
#define DRAY_WALK_QUORUM 12
#endif
#ifndef DRAY_WALK_QUORUM_LDS
#define DRAY_WALK_QUORUM_LDS 8
#endif

// Waves per SIMD the register allocation of the brute-force kernels aims at (8 = at most 64 VGPRs; hybrid scan,
// 1200x675x500 frame: 7 -> 96.4 ms, 8 -> 93.8 ms) and of the grid kernel, which keeps a lane's walk alive across
// the main loop and wants more registers per lane (5 -> 28.8 ms, 6 -> 27.3 ms, 7 -> 27.6 ms with 24 spilled registers).
#ifndef DRAY_WAVES_PER_EU
#define DRAY_WAVES_PER_EU 8
#endif
#ifndef DRAY_GRID_WAVES_PER_EU
#define DRAY_GRID_WAVES_PER_EU 6
#endif

// Diagnostic builds only (tools/isa_budget.py): DRAY_MARKS puts named comments into the instruction stream so that
// instructions can be counted per section of the main loop; DRAY_SECTION_COUNTERS counts how often each section runs.
#ifdef DRAY_MARKS
#define DRAY_MARK(name) asm volatile("; MARK " name)
#else
#define DRAY_MARK(name)
#endif
// DRAY_SECTION_CLOCKS (tools/f32_clocks.py): a wave adds up the shader clocks it spends in each section.
#ifdef DRAY_SECTION_CLOCKS
__device__ unsigned long long dray_section_clocks[16];
#define DRAY_CLOCK(slot)                                               \
    do {                                                               \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
        clk[slot] += now__ - clk_t;                                    \
        clk_t = now__;                                                 \
    } while (0)
#else
#define DRAY_CLOCK(slot)
#endif
#ifdef DRAY_SECTION_COUNTERS
__device__ unsigned long long dray_section_counts[16];
#define DRAY_COUNT(slot, n) (sec[slot] += (unsigned long long)(n))
#else
#define DRAY_COUNT(slot, n)
#endif

// Pointer into the constant address space: a wave-uniform read through it is a scalar load that the compiler
// knows no store can alias.  The sphere tables and the kernel's own arguments are read this way.
#define DRAY_CONSTANT __attribute__((address_space(4)))
typedef float V4 __attribute__((ext_vector_type(4))); // a float4 that can be read through any address space
template <typename T>
using ConstPtr = const DRAY_CONSTANT T*;
template <typename T>
__device__ __forceinline__ ConstPtr<T> as_constant(const T* p) {
    return (ConstPtr<T>)p;
}

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// The rejection samplers' attempts.  A coordinate is -1 + 2 u, u = k 2^-24 (prelude.rs:27-30).
__device__ __forceinline__ float attempt_coord(uint32_t w) { return fmaf(2.f, u01<float>(w), -1.f); }
// vec3.rs:80, 1e-160 <= |p|^2 <= 1 in f32 terms: 0 < l2 <= 1 (a sum of squares of multiples of 2^-23 is 0 or at least
// 2^-46), which for a value that is not negative is one unsigned compare of its bits.
__device__ __forceinline__ bool vector_kept(float l2) { return __float_as_uint(l2) - 1u < 0x3F800000u; }
// vec3.rs:76-84: rejection sampling in the cube; attempt a = words 0..2 of block a of the stage.
__device__ __forceinline__ bool unit_vector_attempt(const uint32_t (&w)[4], F3& out) {
    const F3 p = {attempt_coord(w[0]), attempt_coord(w[1]), attempt_coord(w[2])};
    const float l2 = dot(p, p);
    out = p * rsq(l2);
    return vector_kept(l2);
}

enum { kDrawNone = 0, kDrawUnitVector = 1, kDrawUnitDisk = 2 };

// The two rejection samplers of the path for all lanes of a wave at once.  Called by all 64 lanes (wave-uniform
// control flow).  `draw` says what the lane needs: a unit vector for its scatter (attempt a = words 0..2 of block a
// of its stage), a point in the unit disk for its new sample's aperture (vec3.rs:111-118; attempt m = uniforms 2+2m,
// 3+2m of stage 0, i.e. the second half of block 0, then two attempts per block), or nothing.  `w` holds block 0 of
// the lane's stream.  A unit-vector attempt succeeds with probability pi/6, so run lane by lane the wave would spend
// the MAXIMUM attempt count over its lanes (about six Philox blocks) with most lanes idle.  Instead, after every
// lane's own first attempt, the lanes still pending hand their stream to k = 1, 2, 4 or 8 server lanes each
// (k * pending <= 64), which evaluate the next k blocks in parallel; the pending lane takes the accepted attempt
// with the smallest index.  Same streams, same attempts, same winner as the sequential loops -- in about three
// rounds.  Requests travel through the wave's `req` slots in LDS (LDS keeps one wave's accesses in order), the kept
// attempt's coordinates come back through lane permutes, and only the kept attempt is normalised (vec3.rs:82).
// Returns the unit vector, or the disk point in x, y.
__device__ __forceinline__ F3 coop_draw(const Stream& st, const uint32_t (&w)[4], uint32_t draw, uint4* req, int lane,
                                        uint32_t* rounds = nullptr) {
    float rx = attempt_coord(w[0]), ry = attempt_coord(w[1]), rz = attempt_coord(w[2]); // the kept attempt
    bool pending = false;
    if (draw == kDrawUnitVector) {
        pending = !vector_kept(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
    } else if (draw == kDrawUnitDisk) { // attempt 0 of the aperture sample: uniforms 2, 3 of stage 0
        rx = rz;
        ry = attempt_coord(w[3]);
        pending = !(fmaf(rx, rx, ry * ry) < 1.f);
    }
    uint32_t base = 1u; // next block index, the same for every pending lane
    // one round: the pending lanes `pm` are served blocks base .. base + k - 1 of their streams
    auto round = [&](unsigned long long pm) {
        const uint32_t n_pending = (uint32_t)__popcll(pm);
        const uint32_t shift = n_pending <= 8u ? 3u : n_pending <= 16u ? 2u : n_pending <= 32u ? 1u : 0u; // log2 k
        const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
        if (pending) req[rank] = make_uint4(st.pixel, st.sample, st.stage, draw);
        const uint32_t served = (uint32_t)lane >> shift; // rank of the pending lane this lane serves
        const bool serving = served < n_pending;
        const uint4 r = req[served]; // (a lane that serves nobody reads a slot of 64 that nobody wrote: its verdict is not looked at)
        const Stream other = {st.k0, st.k1, r.x, r.y, r.z};
        uint32_t wb[4];
        other.block(base + ((uint32_t)lane & ((1u << shift) - 1u)), wb);
        // one unit-vector attempt (words 0..2), or two aperture attempts (words 0, 1 and 2, 3): the same four coordinates
        const float px = attempt_coord(wb[0]), py = attempt_coord(wb[1]), pz = attempt_coord(wb[2]), pq = attempt_coord(wb[3]);
        const bool ok_vector = vector_kept(fmaf(pz, pz, fmaf(py, py, px * px)));
        const bool ok_a = fmaf(px, px, py * py) < 1.f, ok_b = fmaf(pz, pz, pq * pq) < 1.f;
        const bool disk = r.w == kDrawUnitDisk;
        const bool second = disk & !ok_a; // the aperture's second attempt of the block is the one to hand back
        const float vx = second ? pz : px, vy = second ? pq : py;
        const bool ok = serving & (disk ? (ok_a | ok_b) : ok_vector);
        const unsigned long long okm = __ballot(ok);
        const uint32_t first = pending ? (rank << shift) : 0u; // my first server lane (< 64; a lane that is not pending has none)
        const uint32_t mine = (uint32_t)(okm >> first) & ((1u << (1u << shift)) - 1u); // my servers' verdicts
        const bool got = pending & (mine != 0u);
        const int from = got ? (int)(first + (uint32_t)__builtin_ctz(mine)) : lane;
        const float gx = __shfl(vx, from, 64), gy = __shfl(vy, from, 64), gz = __shfl(pz, from, 64);
        if (got) {
            rx = gx;
            ry = gy;
            rz = gz;
            pending = false;
        }
        base += 1u << shift;
        if (rounds) ++*rounds;
    };
    // (the first round, which nearly every call needs, is straight-line code: a loop carries the kept attempt through copies)
    unsigned long long pm = __ballot(pending);
    if (pm != 0ull) {
        round(pm);
        pm = __ballot(pending);
        if (pm != 0ull) {
            round(pm);
            for (pm = __ballot(pending); pm != 0ull; pm = __ballot(pending)) round(pm);
        }
    }
    F3 res = {rx, ry, 0.f};
    if (draw == kDrawUnitVector) {
        const F3 p = {rx, ry, rz};
        res = p * rsq(dot(p, p));
    }
    return res;
}

struct PathRay {
    F3 o, d;    // origin, unit direction
    float tmin; // 0.001 in units of the un-normalised direction (camera.rs:270)
};

__device__ __forceinline__ void set_dir(PathRay& r, F3 dir) {
    const float l2 = dot(dir, dir);
    const float inv = rsq(l2);
    r.d = dir * inv;
    r.tmin = 0.001f * (l2 * inv);
}

// camera.rs:238-262.  `w` holds block 0 of the sample's stage-0 stream (uniforms 0, 1: the offsets in the pixel);
// `disk` is the accepted aperture sample (vec3.rs:111-118), drawn by coop_draw.
__device__ __forceinline__ void camera_ray(ConstPtr<CamDev<float>> camp, int32_t i, int32_t j, const uint32_t (&w)[4],
                                           F3 disk, PathRay& r) {
    const DRAY_CONSTANT CamDev<float>& cam = *camp;
    const float fx = (float)i + (u01<float>(w[0]) - 0.5f), fy = (float)j + (u01<float>(w[1]) - 0.5f);
    F3 ps = {fmaf(fy, cam.dv[0], fmaf(fx, cam.du[0], cam.p00[0])), fmaf(fy, cam.dv[1], fmaf(fx, cam.du[1], cam.p00[1])),
             fmaf(fy, cam.dv[2], fmaf(fx, cam.du[2], cam.p00[2]))};
    F3 o = {cam.center[0], cam.center[1], cam.center[2]};
    if (cam.defocus) {
        o.x = fmaf(disk.y, cam.ddv[0], fmaf(disk.x, cam.ddu[0], o.x));
        o.y = fmaf(disk.y, cam.ddv[1], fmaf(disk.x, cam.ddu[1], o.y));
        o.z = fmaf(disk.y, cam.ddv[2], fmaf(disk.x, cam.ddu[2], o.z));
    }
    r.o = o;
    set_dir(r, ps - o);
}

__device__ __forceinline__ uint32_t to_fixed(float c) {
    // NaN and negatives become 0, large values saturate.
    return (uint32_t)(fminf(fmaxf(c, 0.f), kFixMax) * kFixScale);
}

struct ScanBest {
    float t;     // closest accepted root so far (INFINITY: none)
    int32_t pos; // its position in the narrow table, -1: none
};

// The exact f32 test of one narrow sphere, sphere.rs:29-48 with |d| = 1: disc = h^2 - (|oc|^2 - r^2).
// geom = cx, cy, cz, r^2.
template <typename V>
__device__ __forceinline__ void exact_test(ScanBest& b, const V s, const PathRay& ray, int32_t pos) {
    const float ocx = s.x - ray.o.x, ocy = s.y - ray.o.y, ocz = s.z - ray.o.z;
    const float h = fmaf(ocz, ray.d.z, fmaf(ocy, ray.d.y, ocx * ray.d.x));
    const float q = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, s.w)));
    const float disc = fmaf(h, h, q);
    if (disc >= 0.f) {
        const float sq = fsqrt(disc);
        const float t0 = h - sq;
        const bool far = !(t0 > ray.tmin);
        const float t = far ? h + sq : t0;
        if (t > ray.tmin && t < b.t) { // interval.rs:28-30; strict: on equal t the earlier sphere stays
            b.t = t;
            b.pos = pos;
        }
    }
}

// Scalar scan: the exact test on every sphere, 11 VALU operations per test of which 4 take an SGPR operand.
// The table address is the same for all lanes, so the geometry arrives through scalar loads (two
// s_load_dwordx16 per 8 spheres).  Used when the filter table does not fit in LDS, and as the A/B baseline.
__device__ __forceinline__ void scan_scalar(ConstPtr<V4> narrow_geom, int32_t n_padded, const PathRay& ray,
                                            ScanBest& b) {
    for (int32_t k = 0; k < n_padded; k += kUnroll) {
#pragma unroll
        for (int32_t u = 0; u < kUnroll; ++u) exact_test(b, narrow_geom[k + u], ray, k + u);
    }
}

// Hybrid scan: an expanded-form filter in front of the exact test.
//   h  = d.c - d.o                 3 fma, the first starting from -d.o
//   q  = 2 o.c - |o|^2 (1 - 2^-17) 3 fma, the first starting from -|o|^2 (1 - 2^-17)
//   listed  <=>  h^2 + q >= nw,    nw = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2)          1 fma, 1 compare
// |c|^2 - 2 o.c + |o|^2 cancels catastrophically, so this form only FILTERS: the two slacks exceed its rounding
// error plus that of the exact test (46 eps ((|c|+|o|)^2 + r^2) against 128 eps (|c|^2 + |o|^2 + r^2), eps = 2^-24),
// so every sphere the exact test would accept is listed, and listed spheres (about one test in 2 500 on the
// 485-sphere scene) then take the exact test.  Results are therefore identical to scan_scalar's.
// Operand sources: cx, cy of two spheres per wave-uniform ds_read_b128 (LDS, VGPR operands); cz and nw through
// scalar loads (3 SGPR-sourced operations per test).
__device__ __forceinline__ void scan_hybrid(const V4* lds_xy, ConstPtr<V4> filter_zn,
                                            ConstPtr<V4> narrow_geom, int32_t n_padded, const PathRay& ray,
                                            ScanBest& b) {
    const float dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    const float ox2 = 2.f * ray.o.x, oy2 = 2.f * ray.o.y, oz2 = 2.f * ray.o.z;
    const float ndo = -dot(ray.d, ray.o);
    const float oo = dot(ray.o, ray.o);
    const float noo = fmaf(oo, kSlack, -oo);
    for (int32_t k = 0; k < n_padded; k += kUnroll) {
        float lhs[kUnroll], rhs[kUnroll];
#pragma unroll
        for (int32_t u = 0; u < kUnroll; u += 2) {
            const V4 xy = lds_xy[(k + u) >> 1];    // cx0, cy0, cx1, cy1
            const V4 zn = filter_zn[(k + u) >> 1]; // cz0, nw0, cz1, nw1
            const float h0 = fmaf(dz, zn.x, fmaf(dy, xy.y, fmaf(dx, xy.x, ndo)));
            const float q0 = fmaf(oz2, zn.x, fmaf(oy2, xy.y, fmaf(ox2, xy.x, noo)));
            lhs[u] = fmaf(h0, h0, q0);
            rhs[u] = zn.y;
            const float h1 = fmaf(dz, zn.z, fmaf(dy, xy.w, fmaf(dx, xy.z, ndo)));
            const float q1 = fmaf(oz2, zn.z, fmaf(oy2, xy.w, fmaf(ox2, xy.z, noo)));
            lhs[u + 1] = fmaf(h1, h1, q1);
            rhs[u + 1] = zn.w;
        }
#pragma unroll
        for (int32_t u = 0; u < kUnroll; ++u)
            if (lhs[u] >= rhs[u]) exact_test(b, narrow_geom[k + u], ray, k + u);
    }
}

constexpr int kScanScalar = DRAY_SCAN_SCALAR, kScanHybrid = DRAY_SCAN_HYBRID, kScanGrid = DRAY_SCAN_GRID;
constexpr int kScanGridLds = 4; // names the kernel that walks the grid from tables staged in LDS (not a caller's choice)

// Grid scan: every lane walks its own ray through the uniform grid (3-D DDA) and applies the exact test to the
// spheres listed in the cells it crosses, front to back, until the closest accepted root lies before the exit of
// the current cell.  The result is the minimum over (t, insertion position) of the accepted roots, which is what
// the scalar scan returns, provided no sphere the exact test would accept at the minimum is skipped:
//   * a sphere is listed in every cell its ball, grown by 1/256 of a cell, overlaps, and the grid box holds all
//     grown balls: the f32 rounding of the walk (positions good to ~1e-6 of the scene size) or of the exact test
//     cannot move a ray or a root across that margin;
//   * the walk stops only when the best root is before the cell's exit by more than the exit's rounding error;
//   * boundary times are recomputed from the integer cell index each step, never accumulated.
// Spheres spanning too many cells are kept in a short list every ray tests first.
// A lane's walk survives the kernel's main loop: lanes whose walk is over are shaded and given their next ray
// while the long walks of other lanes go on, instead of the whole wave waiting for its longest walk.
struct Walk {
    // time at which the ray leaves cell k of an axis: fma(k, A, B); A = cell / d, B = (lo - o + [d >= 0] cell) / d
    float A[3], B[3];
    float slack;       // bound on the rounding error of those times
    float c[3];        // current cell (small integers, exact in f32)
    uint32_t e, e_end; // entries of the current cell still to test
    // Closest accepted root so far as one ordered key: bits of t (positive floats order like integers) in the high
    // word, the sphere's position in the caller's list in the low one.  The minimum key is the scalar scan's
    // answer: smallest t, on equal t the sphere inserted first (hittable_list.rs:65-70).  (Which of its two roots
    // the winner was accepted with is not kept: the winner is evaluated again before shading, and every select
    // spared here is spared eight times a segment.)
    unsigned long long best;
};

constexpr unsigned long long kNoHit = 0x7F800000FFFFFFFFull; // t = +inf

__device__ __forceinline__ void walk_test(Walk& w, const V4 s, int32_t pos, const PathRay& ray) {
    // exact_test without branches: in a walked cell a test succeeds often enough that some lane of the wave nearly
    // always takes them, and every level of branching costs mask bookkeeping on the scalar unit
    const float ocx = s.x - ray.o.x, ocy = s.y - ray.o.y, ocz = s.z - ray.o.z;
    const float h = fmaf(ocz, ray.d.z, fmaf(ocy, ray.d.y, ocx * ray.d.x));
    const float q = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, s.w)));
    const float disc = fmaf(h, h, q);
    const float sq = fsqrt(disc); // NaN for a negative discriminant: then t is NaN and fails the comparison below
    const float t0 = h - sq;
    const bool far = !(t0 > ray.tmin);
    const float t = far ? h + sq : t0;
    const bool valid = t > ray.tmin; // t > tmin > 0: its bits order like an unsigned integer
    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)pos;
    w.best = (valid & (key < w.best)) ? key : w.best;
}

// Tests the loose spheres, then places the ray in its first cell.  Returns false when there is nothing to walk.
template <typename Grid>
__device__ __forceinline__ bool walk_begin(ConstPtr<GridDev> gp, const Grid& view, const PathRay& ray, Walk& w) {
    const DRAY_CONSTANT GridDev& g = *gp;
    w.best = kNoHit;
    ConstPtr<V4> loose_geom = as_constant((const V4*)g.loose_geom);
    ConstPtr<int32_t> loose_index = as_constant(g.loose_index);
    const int32_t n_loose = g.n_loose; // wave-uniform; rows and list positions arrive through scalar loads, in fours
    int32_t k = 0;
    for (; k + 4 <= n_loose; k += 4) {
#pragma unroll
        for (int32_t u = 0; u < 4; ++u) walk_test(w, loose_geom[k + u], loose_index[k + u], ray);
    }
    for (; k < n_loose; ++k) walk_test(w, loose_geom[k], loose_index[k], ray); // (padding rows are never read)
    if (g.n[0] == 0) return false;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
    float t_in = 0.f, t_out = INFINITY, bmax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // A direction parallel to the axis' planes (or NaN) walks as one that crosses them after 10^12 cell sizes: from
        // inside the slab every plane is ~10^12 away and never the next one, from outside t_in > t_out.  (No compares
        // or selects: each costs a SIMD almost twice an addition, and the set-up had nine per axis.)
        const float da = copysignf(fmaxf(fabsf(d[a]), 1e-12f), d[a]);
        const float inv = rcp(da);
        w.A[a] = g.cell[a] * inv;
        const float t0 = (g.lo[a] - o[a]) * inv, t1 = (g.hi[a] - o[a]) * inv;
        w.B[a] = t0 + fmaxf(w.A[a], 0.f); // leaving cell k towards larger indices happens at plane k + 1
        t_in = fmaxf(t_in, fminf(t0, t1));
        t_out = fminf(t_out, fmaxf(t0, t1));
        bmax = fmaxf(bmax, fmaxf(fabsf(t0), fabsf(w.B[a])));
    }
    w.slack = bmax * 3.814697265625e-6f; // 2^-18 of the largest term the boundary times are formed from
    const bool walking = t_in <= t_out + w.slack;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float at = (fmaf(d[a], t_in, o[a]) - g.lo[a]) * g.inv_cell[a];
        w.c[a] = fminf(fmaxf(floorf(at), 0.f), (float)(g.n[a] - 1));
    }
    w.e = w.e_end = 0u;
    if (walking) {
        const uint32_t ci = (uint32_t)fmaf(fmaf(w.c[2], (float)g.n[1], w.c[1]), (float)g.n[0], w.c[0]);
        const uint2 rec = cell_record(view, ci);
        w.e = rec.x;
        w.e_end = rec.x + rec.y;
    }
    return walking;
}

// What a step reads of the grid, held in scalar registers while a wave is stepping.
struct GridRegs {
    float n0, n1, n2; // cells per axis (at most 256: products and sums below are exact in f32)
    const uint2* cells;
    const V4* entry_geom;
    const int32_t* entry_index;
};

// The same for a workgroup that has staged the grid in LDS, in a compact form (so that six workgroups' copies fit a
// compute unit): a cell record is one word -- first entry (12 bits) | entry count (4 bits) | list position of the
// first entry's sphere (16 bits) --, an entry the 16-bit list position of its sphere, and the spheres' rows (cx, cy,
// cz, r*r) are kept once, by list position.  A lane that enters a cell has the first sphere to test from the cell's
// record, a lane that stays in its cell reads the entry table: either way two reads that depend on one another per
// step, as with the tables in memory.
struct GridLds {
    float n0, n1, n2;
    const uint32_t* cells;
    const V4* geom;
    const uint16_t* entry_index;
};
constexpr uint32_t kLdsBeginBits = 12u, kLdsCountBits = 4u; // entries < 2^12, entries per cell < 2^4 (checked by the plan)

__device__ __forceinline__ uint2 cell_record(const GridRegs& g, uint32_t ci) { return g.cells[ci]; }
__device__ __forceinline__ uint2 cell_record(const GridLds& g, uint32_t ci) {
    const uint32_t rec = g.cells[ci];
    return make_uint2(rec & ((1u << kLdsBeginBits) - 1u), (rec >> kLdsBeginBits) & ((1u << kLdsCountBits) - 1u));
}

// One step of a walking lane: leave a finished cell (or stop), then test one listed sphere.  Returns false when
// the walk is over.  Written with selects and bitwise conditions instead of nested branches: the lanes of a wave
// rarely agree here.
__device__ __forceinline__ bool walk_step(const GridRegs& g, const PathRay& ray, Walk& w) {
    bool go = true;
    if (w.e == w.e_end) { // this cell is done: stop, or step into the next one
        const float tx = fmaf(w.c[0], w.A[0], w.B[0]);
        const float ty = fmaf(w.c[1], w.A[1], w.B[1]);
        const float tz = fmaf(w.c[2], w.A[2], w.B[2]);
        const float t_exit = fminf(tx, fminf(ty, tz));
        const bool stop = __uint_as_float((uint32_t)(w.best >> 32)) < fmaf(t_exit, 1.f - 3.814697265625e-6f, -w.slack);
        const bool sx = tx == t_exit;
        const bool sy = !sx & (ty == t_exit);
        // one step along the axis left first; A carries the direction's sign (a parallel axis is never chosen)
        w.c[0] += sx ? copysignf(1.f, w.A[0]) : 0.f;
        w.c[1] += sy ? copysignf(1.f, w.A[1]) : 0.f;
        w.c[2] += (sx | sy) ? 0.f : copysignf(1.f, w.A[2]);
        // still inside the grid: a coordinate is a whole number from -1 to n, and as unsigned integers the bits of -1
        // lie above those of every count (three compares; picking the axis that moved and its count costs four selects)
        const bool in0 = __float_as_uint(w.c[0]) < __float_as_uint(g.n0), in1 = __float_as_uint(w.c[1]) < __float_as_uint(g.n1);
        const bool in2 = __float_as_uint(w.c[2]) < __float_as_uint(g.n2);
        const bool inside = in0 & in1 & in2;
        go = inside & !stop;
        const uint32_t ci = (uint32_t)fmaf(fmaf(w.c[2], g.n1, w.c[1]), g.n0, w.c[0]);
        const uint2 rec = g.cells[go ? ci : 0u];
        w.e = rec.x;
        w.e_end = go ? rec.x + rec.y : rec.x;
    }
    if (w.e != w.e_end) {
        walk_test(w, g.entry_geom[w.e], g.entry_index[w.e], ray);
        ++w.e;
    }
    return go;
}

// The same step over the copy in LDS.
__device__ __forceinline__ bool walk_step(const GridLds& g, const PathRay& ray, Walk& w) {
    bool go = true;
    uint32_t idx = 0u;
    if (w.e == w.e_end) {
        const float tx = fmaf(w.c[0], w.A[0], w.B[0]);
        const float ty = fmaf(w.c[1], w.A[1], w.B[1]);
        const float tz = fmaf(w.c[2], w.A[2], w.B[2]);
        const float t_exit = fminf(tx, fminf(ty, tz));
        const bool stop = __uint_as_float((uint32_t)(w.best >> 32)) < fmaf(t_exit, 1.f - 3.814697265625e-6f, -w.slack);
        const bool sx = tx == t_exit;
        const bool sy = !sx & (ty == t_exit);
        w.c[0] += sx ? copysignf(1.f, w.A[0]) : 0.f;
        w.c[1] += sy ? copysignf(1.f, w.A[1]) : 0.f;
        w.c[2] += (sx | sy) ? 0.f : copysignf(1.f, w.A[2]);
        // still inside the grid: a coordinate is a whole number from -1 to n, and as unsigned integers the bits of -1
        // lie above those of every count (three compares; picking the axis that moved and its count costs four selects)
        const bool in0 = __float_as_uint(w.c[0]) < __float_as_uint(g.n0), in1 = __float_as_uint(w.c[1]) < __float_as_uint(g.n1);
        const bool in2 = __float_as_uint(w.c[2]) < __float_as_uint(g.n2);
        const bool inside = in0 & in1 & in2;
        go = inside & !stop;
        const uint32_t ci = (uint32_t)fmaf(fmaf(w.c[2], g.n1, w.c[1]), g.n0, w.c[0]);
        const uint32_t rec = g.cells[go ? ci : 0u];
        w.e = rec & ((1u << kLdsBeginBits) - 1u);
        w.e_end = go ? w.e + ((rec >> kLdsBeginBits) & ((1u << kLdsCountBits) - 1u)) : w.e;
        idx = rec >> (kLdsBeginBits + kLdsCountBits); // the new cell's first sphere
    } else {
        idx = g.entry_index[w.e]; // (issued beside the other lanes' cell read: the two do not depend on one another)
    }
    if (w.e != w.e_end) {
        walk_test(w, g.geom[idx], (int32_t)idx, ray);
        ++w.e;
    }
    return go;
}

// Everything a launch of the trace kernel is given.
struct F32Args {
    CamDev<float> cam;
    const V4* narrow_geom;
    const V4* filter_xy;
    const V4* filter_zn;
    const int32_t* narrow_index;
    const int32_t* wide_index;
    const double* geom64;
    const double* wide_geom64;
    const V4* wide_axial;
    const V4* geom32;
    const V4* mat32;
    const uint32_t* kind;
    int32_t n_narrow_padded, n_wide;
    const double* box64;      // bounding boxes of the list: min xyz, max xyz (kernels built with BOXES only)
    const int32_t* box_index; // their positions in the caller's list, ascending
    int32_t n_boxes;
    float bound[4]; // F32Tables::bound
    GridDev grid;
    Partition part;
    RngKey key;
    int32_t pixels_per_pool;
    int32_t grab_pixels; // pixels a wave takes from the global queue per atomic (a multiple of pixels_per_pool)
    int32_t no_refill;   // idle lanes take new samples only once every lane of the wave is idle (compaction off)
    int32_t n_cells, n_entries, n_list; // cells and entries of the grid, length of the caller's list (the kernel that stages the grid in LDS)
    unsigned long long* accum;
    DeviceCounters* counters;
};

// Hides where a pointer came from, so that reads through it are not moved across this point.
template <typename T>
__device__ __forceinline__ ConstPtr<T> opaque(ConstPtr<T> p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (ConstPtr<T>)v;
}

// BOXES: the list holds bounding boxes (bounding_box.rs) beside its spheres.  Such a kernel carries two more values
// per lane (the closest box of the lane's ray), so scenes without boxes -- all of the reference's render scenes --
// run the kernels built without.
// LDSGRID (grid scan only): the workgroup stages the grid in LDS (GridLds) and its lanes walk that copy: a walking
// lane's reads depend on one another (cell record, entry, the sphere's row), and an LDS read comes back in a fraction
// of the time a read through the vector memory path takes.
template <int SCAN, bool BOXES, bool LDSGRID>
__device__ __forceinline__ void render_f32_body() {
    static_assert(!LDSGRID || SCAN == kScanGrid, "only the grid scan has tables to stage");
    // The arguments (some sixty scalar registers' worth) are read where they are used, straight from the kernel
    // argument segment, and the pointer to it is made opaque once per path segment.  Left to itself the compiler
    // loads them all above the main loop, runs out of scalar registers and spills inside the sphere scan -- where a
    // change anywhere else in the kernel then decides how many reloads the scan carries.
    ConstPtr<F32Args> a = (ConstPtr<F32Args>)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = __lane_id();
    extern __shared__ V4 lds_xy[]; // hybrid scan: cx, cy of two consecutive narrow spheres per entry
    // LDSGRID: the spheres' rows by list position (16 B each), then the cell records (4 B), then the entries (2 B)
    if constexpr (SCAN == kScanHybrid) {
        const int32_t half = a->n_narrow_padded / 2;
        ConstPtr<V4> filter_xy = as_constant(a->filter_xy);
        for (int32_t k = threadIdx.x; k < half; k += blockDim.x) lds_xy[k] = filter_xy[k];
        __syncthreads(); // the only block-wide wait: every wave passes here once, before any work is drawn
    }
    if constexpr (LDSGRID) {
        const int32_t n_entries = a->n_entries, n_cells = a->n_cells, n_list = a->n_list;
        ConstPtr<V4> narrow_geom = as_constant(a->narrow_geom);
        ConstPtr<int32_t> narrow_index = as_constant(a->narrow_index);
        const uint2* cells = a->grid.cells;
        ConstPtr<int32_t> entry_index = as_constant(a->grid.entry_index);
        uint32_t* lds_cells = (uint32_t*)(lds_xy + n_list);
        uint16_t* lds_entries = (uint16_t*)(lds_cells + n_cells);
        // (rows of list positions that hold no narrow sphere -- wide spheres, boxes -- are never read: no entry names them)
        for (int32_t k = threadIdx.x; k < a->n_narrow_padded; k += blockDim.x) {
            const int32_t at = narrow_index[k];
            if (at >= 0 && at < n_list) lds_xy[at] = narrow_geom[k];
        }
        for (int32_t k = threadIdx.x; k < n_cells; k += blockDim.x) {
            const uint2 rec = cells[k];
            const uint32_t first = rec.y != 0u ? (uint32_t)entry_index[rec.x] : 0u;
            lds_cells[k] = rec.x | (rec.y << kLdsBeginBits) | (first << (kLdsBeginBits + kLdsCountBits));
        }
        for (int32_t k = threadIdx.x; k < n_entries; k += blockDim.x) lds_entries[k] = (uint16_t)entry_index[k];
        __syncthreads(); // as above: once, before any work is drawn
    }

    // A wave takes whole pixels from the global queue (a pool = all samples of `pixels_per_pool` consecutive
    // pixels) and hands the samples to its lanes one at a time, so the samples of a pixel finish in one wave.
    // Their fixed-point sums meet in a wave-private LDS cache (the wave's pixels take its kSlots slots in turn)
    // and go to memory once, as a plain store, when the slot is given to a later pixel; a sample that ends after its
    // pixel has left the cache adds to memory directly (an atomic on top of that store).  Integer sums: the image is the same whatever the grouping.  LDS keeps
    // one wave's accesses in order, and tags and sums are only ever touched by their own wave: no barrier.
    __shared__ unsigned long long lds_sum[4][kSlots][3]; // [wave of the block][slot][r, g, b]
    __shared__ uint32_t lds_tag[4][kSlots];              // the slot's pixel: its Philox pixel word (j * W + i)
    __shared__ uint4 lds_req[4][64];                     // coop_draw: streams handed to server lanes
    static_assert(sizeof(lds_sum) + sizeof(lds_tag) + sizeof(lds_req) <= kF32StaticLds, "static LDS outgrew its budget");
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (the same in every lane of a wave: said so for the compiler)
    if (lane < kSlots) {
        __hip_atomic_store(&lds_tag[wv][lane], kNoPixel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_sum[wv][lane][0] = lds_sum[wv][lane][1] = lds_sum[wv][lane][2] = 0ull;
    }
    // index of image pixel j * W + i in the launch's row partition (the sum planes and the queue count these)
    auto partition_index = [&](uint32_t image_pixel) {
        const uint32_t W = (uint32_t)a->cam.W;
        const uint32_t j = image_pixel / W, i = image_pixel - j * W;
        return ((j - (uint32_t)a->part.row_begin) / (uint32_t)a->part.row_step) * W + i;
    };
    auto image_pixel_of = [&](uint32_t partition_pixel) {
        const uint32_t W = (uint32_t)a->cam.W;
        const uint32_t r = partition_pixel / W, i = partition_pixel - r * W;
        return ((uint32_t)a->part.row_begin + r * (uint32_t)a->part.row_step) * W + i;
    };
    auto pixel_count = [&]() { return (uint32_t)a->part.n_rows * (uint32_t)a->cam.W; };
    // lane `l < n` retires slot `first_slot + l` and, if `retag`, gives it to partition pixel `first_pixel + l`
    auto retire_slots = [&](uint32_t first_slot, uint32_t first_pixel, uint32_t n, bool retag) {
        if ((uint32_t)lane < n) {
            const uint32_t slot = (first_slot + (uint32_t)lane) & (uint32_t)(kSlots - 1);
            const uint32_t old = __hip_atomic_load(&lds_tag[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old != kNoPixel) {
                const uint32_t at = partition_index(old);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned long long v = __hip_atomic_exchange(&lds_sum[wv][slot][c], 0ull, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    // A pixel's sums leave the cache once, here, as a plain store -- also when they are zero --, so the
                    // planes need no clearing before a launch: every pixel of the partition is written exactly once.
                    a->accum[(size_t)c * pixel_count() + at] = v;
                }
            }
            __hip_atomic_store(&lds_tag[wv][slot], retag ? image_pixel_of(first_pixel + (uint32_t)lane) : kNoPixel,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // adds one finished sample's colour to its pixel (colors.rs:38): the wave's LDS slot, or memory if the pixel has
    // left the cache
    auto deposit = [&](float cr, float cg, float cb, uint32_t pixel_word, uint32_t slot) {
        const unsigned long long fr = to_fixed(cr), fg = to_fixed(cg), fb = to_fixed(cb);
#ifdef DRAY_EXPERIMENT_SPREAD_DEPOSITS
        // Diagnostic builds only (the image is wrong): every lane adds to a slot of its own, which takes the
        // same-address conflicts of the deposits out of the LDS counters: what is left are the grid copy's.
        slot = (uint32_t)lane & (uint32_t)(kSlots - 1);
        pixel_word = __hip_atomic_load(&lds_tag[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        if (__hip_atomic_load(&lds_tag[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == pixel_word) {
            __hip_atomic_fetch_add(&lds_sum[wv][slot][0], fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds_sum[wv][slot][1], fg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds_sum[wv][slot][2], fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // the pixel has left the cache: its sums are in memory, stored by a lane of this wave.  That store must
            // have been performed before this addition is: the fence waits for the wave's outstanding stores.
            const uint32_t at = partition_index(pixel_word), npix = pixel_count();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(&a->accum[at], fr);
            atomicAdd(&a->accum[(size_t)npix + at], fg);
            atomicAdd(&a->accum[2 * (size_t)npix + at], fb);
        }
    };

    // per-lane path state
    uint32_t gpix = 0u; // Philox pixel word j * W + i of the lane's sample
    uint32_t ss = 0u;   // sample index within the call (24 bits) | cache slot << 24
    int32_t stage = 0;  // 0: idle, k > 0: on the k-th segment of a path, < 0: the queue has run out
    unsigned long long n_seg = 0ull; // segments of this wave, counted on the scalar unit from the lane mask
    PathRay ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.001f};
    F3 thr = {1.f, 1.f, 1.f};
    bool fresh = false;   // the ray has not been intersected yet
    bool camera = false;  // a sample has been taken and its camera ray is still to be made
    bool walking = false; // grid scan: the lane is on its way through the grid
    ScanBest sb = {INFINITY, -1};
    Walk walk = {};
    float box_t = INFINITY; // BOXES: distance of the closest box along the lane's ray, and the box's list position
    int32_t box_idx = -1;
    // wave-uniform queue state
    uint32_t pool_pix = 0u, pool_q = 0u, pool_n = 0u; // first pixel, next sample, sample count of the pool
    uint32_t pool_r0 = 0u, pool_i0 = 0u;               // row and column of the pool's first pixel in the partition
    uint32_t pool_dp = 0u, pool_smp = 0u;              // pool_q as (pixel, sample), kept when `stepwise`
    uint32_t pool_slot = 0u, pool_pixels = 0u;         // cache slot of the pool's first pixel, pixels in the pool
    unsigned long long n_samples = 0ull;               // samples of the pools this wave has opened
    uint32_t n_steps = 0u;                             // grid scan: steps this wave has taken
    uint32_t grab_next = 0u, grab_end = 0u;            // pixels grabbed from the global queue, not yet in a pool
    uint32_t grab_pixels = (uint32_t)a->grab_pixels;   // size of the next grab
    bool exhausted = false;

#ifdef DRAY_SECTION_COUNTERS
    unsigned long long sec[16] = {};
#endif
#ifdef DRAY_SECTION_CLOCKS
    unsigned long long clk[16] = {};
    unsigned long long clk_t = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        a = opaque(a);
        DRAY_CLOCK(9); // loop edge (and, on the first turn, the kernel's entry)

        // ---- closest hit over the sphere list: hittable_list.rs:60-74 ---------------------------
        DRAY_MARK("A_fresh_setup");
        DRAY_COUNT(0, 1);                                          // main-loop iterations
        DRAY_COUNT(1, __ballot(fresh) != 0ull ? 1 : 0);            // iterations that set up a scan
        n_seg += (unsigned long long)__popcll(__ballot(fresh));
        if (fresh) {
            // The origin moves along the ray up to the ball that holds the narrow spheres (the ray's parameters,
            // tmin included, move with it).  The exact test takes |oc|^2 from r^2, and seen from far away -- a bounce
            // off a huge enclosing sphere, a distant camera -- r^2 is lost in the rounding of |oc|^2: the test then
            // accepts spheres the ray passes at a distance (0.02 units at |oc| = 50), and which of those a scan meets
            // depends on the scan.  From the ball's rim |oc| is at most its diameter.  Nothing narrow lies before the
            // rim, so the narrow tests may take max(tmin, 0) for their lower bound; the wide spheres keep tmin.
            if constexpr (BOXES) {
                // BoundingBox::hit, bounding_box.rs:50-76, in f64 as it is written: axes 0 and 1 only (:53), plane
                // distances `min - origin / direction` (:55-56) -- which are not invariant under scaling the direction,
                // so the reference's un-normalised direction is rebuilt from the unit one (tmin = 0.001 |direction|).
                // Tested here because the origin is about to be moved.  Distances come out in units of the unit direction.
                box_t = INFINITY;
                box_idx = -1;
                const double len = (double)(ray.tmin * 1000.f);
                const double bo[2] = {(double)ray.o.x, (double)ray.o.y};
                const double bd[2] = {(double)ray.d.x * len, (double)ray.d.y * len};
                ConstPtr<double> box64 = as_constant(a->box64);
                for (int32_t q = 0; q < a->n_boxes; ++q) { // wave-uniform
                    double t_min = -INFINITY, t_max = INFINITY;
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
                        if (bd[ax] != 0.0) {
                            const double t0 = box64[6 * q + ax] - bo[ax] / bd[ax];
                            const double t1 = box64[6 * q + 3 + ax] - bo[ax] / bd[ax];
                            t_min = fmax(t_min, fmin(t0, t1));
                            t_max = fmin(t_max, fmax(t0, t1));
                        }
                    }
                    if (t_max >= t_min) {
                        double t = t_min;
                        if (!(0.001 < t)) t = t_max; // interval.rs:28-30
                        const float tu = (float)(t * len);
                        if (0.001 < t && tu < box_t) { // ascending list positions: on equal distance the earlier box stays
                            box_t = tu;
                            box_idx = as_constant(a->box_index)[q];
                        }
                    }
                }
            }
            {
                const float bx = a->bound[0] - ray.o.x, by = a->bound[1] - ray.o.y, bz = a->bound[2] - ray.o.z;
                const float s = fmaxf(fmaf(bz, ray.d.z, fmaf(by, ray.d.y, bx * ray.d.x)) - a->bound[3], 0.f);
                ray.o = {fmaf(s, ray.d.x, ray.o.x), fmaf(s, ray.d.y, ray.o.y), fmaf(s, ray.d.z, ray.o.z)};
                ray.tmin -= s;
                if constexpr (BOXES) box_t -= s; // distances are counted from the origin: it has just moved by s
            }
            PathRay nray = ray;
            nray.tmin = fmaxf(ray.tmin, 0.f);
            sb = {INFINITY, -1};
            if constexpr (SCAN == kScanHybrid)
                scan_hybrid(lds_xy, as_constant(a->filter_zn), as_constant(a->narrow_geom), a->n_narrow_padded, nray, sb);
            else if constexpr (SCAN == kScanGrid)
            {
                if constexpr (LDSGRID) {
                    const GridLds view = {0.f, 0.f, 0.f, (const uint32_t*)(lds_xy + a->n_list), lds_xy, nullptr};
                    walking = walk_begin(&a->grid, view, nray, walk);
                } else {
                    const GridRegs view = {0.f, 0.f, 0.f, a->grid.cells, nullptr, nullptr};
                    walking = walk_begin(&a->grid, view, nray, walk);
                }
            }
            else
                scan_scalar(as_constant(a->narrow_geom), a->n_narrow_padded, nray, sb);
            fresh = false;
        }
        if constexpr (SCAN == kScanGrid) {
            // walk until fewer than a quorum of lanes are still on their way (all of them once the queue is empty)
            const int quorum = exhausted ? 1 : (LDSGRID ? DRAY_WALK_QUORUM_LDS : DRAY_WALK_QUORUM);
            auto grid_view = [&]() {
                if constexpr (LDSGRID) {
                    const uint32_t* cells = (const uint32_t*)(lds_xy + a->n_list);
                    return GridLds{(float)a->grid.n[0], (float)a->grid.n[1], (float)a->grid.n[2], cells, lds_xy,
                                   (const uint16_t*)(cells + a->n_cells)};
                } else {
                    return GridRegs{(float)a->grid.n[0], (float)a->grid.n[1], (float)a->grid.n[2], a->grid.cells,
                                    (const V4*)a->grid.entry_geom, a->grid.entry_index};
                }
            };
            const auto gr = grid_view();
            PathRay nray = ray;
            nray.tmin = fmaxf(ray.tmin, 0.f); // see the origin's move above
            DRAY_CLOCK(0);
            DRAY_MARK("B_walk_steps");
            do { // two steps per count: the count is scalar-unit work every walking lane waits for
#ifndef DRAY_STEPS_PER_COUNT
#define DRAY_STEPS_PER_COUNT 2
#endif
                DRAY_COUNT(2, DRAY_STEPS_PER_COUNT);                                  // walk steps (wave level)
                DRAY_COUNT(3, DRAY_STEPS_PER_COUNT * __popcll(__ballot(walking)));    // lanes walking at those steps
#pragma unroll
                for (int step = 0; step < DRAY_STEPS_PER_COUNT; ++step)
                    if (walking) walking = walk_step(gr, nray, walk);
                n_steps += (uint32_t)DRAY_STEPS_PER_COUNT;
            } while (__popcll(__ballot(walking)) >= quorum);
        }
        DRAY_CLOCK(1);
        DRAY_MARK("C_hit");

        // ---- the hit: winner of the narrow scan against the wide spheres, point and normal --------------
        const bool shading = stage > 0 && !walking;
        int32_t bidx = -1; // position of the closest sphere in the caller's list
        F3 hp = {0.f, 0.f, 0.f}, hn = {0.f, 0.f, 0.f};
        uint32_t mk = 0xFFu; // its material kind; 0xFF: no hit
        V4 m = {0.f, 0.f, 0.f, 0.f};
        if (shading) {
            if constexpr (SCAN == kScanGrid) {
                sb.t = __uint_as_float((uint32_t)(walk.best >> 32));
                sb.pos = (int32_t)(uint32_t)walk.best; // (no hit: all ones, -1)
            }
            const float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
            const float tmin = ray.tmin;
            float best = sb.t;
            // Re-evaluate the winner from the perpendicular remainder |oc - h d|, which does not cancel.
            if (sb.pos >= 0) {
                if constexpr (SCAN == kScanGrid) bidx = sb.pos; // the walk's key names the list entry itself
                else bidx = as_constant(a->narrow_index)[sb.pos];
                const V4 s = as_constant(a->geom32)[bidx];
                const F3 oc = {s.x - ox, s.y - oy, s.z - oz};
                const float h = dot(oc, ray.d);
                const F3 l = {fmaf(-h, dx, oc.x), fmaf(-h, dy, oc.y), fmaf(-h, dz, oc.z)};
                const float disc = fmaxf(fmaf(s.w, s.w, -dot(l, l)), 0.f);
                const float sq = fsqrt(disc);
                best = h - sq; // sphere.rs:38-44
                if (!(best > tmin)) best = h + sq;
                hp = {fmaf(best, dx, ox), fmaf(best, dy, oy), fmaf(best, dz, oz)};
                const float ir = rcp(s.w);
                hn = {(hp.x - s.x) * ir, (hp.y - s.y) * ir, (hp.z - s.z) * ir}; // sphere.rs:52
            }
            const int32_t n_wide = a->n_wide;
            ConstPtr<int32_t> wide_index = as_constant(a->wide_index);
            ConstPtr<double> geom64 = as_constant(a->wide_geom64); // the wide spheres' rows, side by side
            ConstPtr<V4> wide_axial = as_constant(a->wide_axial);
            auto wide_sphere = [&](int32_t wq) {
#ifndef DRAY_NO_AXIAL
                if (wide_axial[2 * wq + 1].z != 0.f) {
                    // A wide sphere whose centre lies by the y axis (the ground, src/main.rs:14), sphere.rs:28-57 in f32 without
                    // the cancellation of |oc|^2 - r^2: c = ocx^2 + ocz^2 + (cy + r - oy)(cy - r - oy), the two sums
                    // rounded once on the host; and of the two roots the one that is a sum comes from h and the square
                    // root, the other from c divided by it (their product is c): no difference of large numbers anywhere.
                    const int32_t idx = wide_index[wq];
                    const V4 g = wide_axial[2 * wq], k = wide_axial[2 * wq + 1];
                    const float ocx = g.x - ox, ocy = g.y - oy, ocz = g.z - oz;
                    const float h = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
                    const float c = fmaf(k.x - oy, k.y - oy, fmaf(ocz, ocz, ocx * ocx));
                    const float disc = fmaf(h, h, -c);
                    if (disc >= 0.f) {
                        const float sq = fsqrt(disc);
                        const float big = h + copysignf(sq, h); // the root that is a sum
                        const float small = c * rcp(big);       // (big = 0 only with c = 0: NaN, which fails the comparisons below)
                        float t = fminf(big, small);            // the near root: the sum if h < 0, the quotient otherwise
                        if (!(t > tmin)) t = fmaxf(big, small);
                        if (t > tmin && (t < best || (t == best && idx < bidx))) {
                            best = t;
                            bidx = idx;
                            hp = {fmaf(t, dx, ox), fmaf(t, dy, oy), fmaf(t, dz, oz)};
                            hn = {(hp.x - g.x) * g.w, (hp.y - g.y) * g.w, (hp.z - g.z) * g.w}; // sphere.rs:52
                        }
                    }
                    return;
                }
#endif
                // sphere.rs:28-57 in f64 for a sphere too large for f32: the cancelling terms (oc, |oc|^2 - r^2,
                // the discriminant, the hit point relative to the centre) are f64; divisions are avoided
                // (1/|d|^2 = 2 - |d|^2 to 1e-14 because d is unit to f32 rounding; sqrt = f32 estimate + one Newton step)
                const int32_t idx = wide_index[wq];
                const double cx = geom64[4 * wq + 0], cy = geom64[4 * wq + 1], cz = geom64[4 * wq + 2];
                const double rad = geom64[4 * wq + 3];
                const double ocx = cx - (double)ox, ocy = cy - (double)oy, ocz = cz - (double)oz;
                const double ddx = dx, ddy = dy, ddz = dz;
                const double aa = fma(ddz, ddz, fma(ddy, ddy, ddx * ddx));
                const double h = fma(ddz, ocz, fma(ddy, ocy, ddx * ocx));
                const double c = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -(rad * rad))));
                const double disc = fma(h, h, -(aa * c));
                if (disc >= 0.) {
                    const float s0f = fsqrt((float)disc);
                    const double s0 = (double)s0f;
                    const double sq = s0f > 0.f ? fma(fma(-s0, s0, disc), (double)(0.5f * rcp(s0f)), s0) : 0.;
                    const double inva = 2.0 - aa;
                    double t = (h - sq) * inva;
                    if (!(t > (double)tmin)) t = (h + sq) * inva;
                    const float tf = (float)t;
                    if (t > (double)tmin && (tf < best || (tf == best && idx < bidx))) {
                        best = tf;
                        bidx = idx;
                        const double px = fma(t, ddx, (double)ox), py = fma(t, ddy, (double)oy), pz = fma(t, ddz, (double)oz);
                        hp = {(float)px, (float)py, (float)pz};
                        const float ir = rcp((float)rad);
                        hn = {(float)(px - cx) * ir, (float)(py - cy) * ir, (float)(pz - cz) * ir}; // sphere.rs:52
                    }
                }
            };
            // (one wide sphere, the usual case, is straight-line code: a loop carries best, point and normal through copies)
            if (n_wide == 1) wide_sphere(0);
            else for (int32_t wq = 0; wq < n_wide; ++wq) wide_sphere(wq);
            if constexpr (BOXES) {
                // a box in front of the closest sphere (on equal distance: earlier in the list) is what the ray hits; its
                // material, Transparent, makes the sample black (its kind in this kernel's table is the absorber's)
                if (box_idx >= 0 && (box_t < best || (box_t == best && box_idx < bidx))) {
                    best = box_t;
                    bidx = box_idx;
                }
            }
            if (bidx >= 0) {
                mk = as_constant(a->kind)[bidx];
                m = as_constant(a->mat32)[bidx];
            }
        }

        // ---- paths that end here without scattering: sky (camera.rs:279-281), an absorbing material
        // (material.rs:13-19), or no depth left for the scattered ray (camera.rs:265-267: its colour is 0) ------
        DRAY_CLOCK(2);
        DRAY_MARK("D_deposit");
        DRAY_COUNT(4, __popcll(__ballot(shading)));                // lanes shading
        DRAY_COUNT(5, __ballot(shading) != 0ull ? 1 : 0);          // iterations with a lane shading
        const bool ends_now = shading && (bidx < 0 || mk == DRAY_MAT_DEFAULT || stage >= a->cam.max_depth);
        if (ends_now) {
            if (bidx < 0) {
                const float ta = 0.5f * (ray.d.y + 1.0f);
                const float tw = 1.0f - ta;
                deposit(thr.x * fmaf(ta, 0.5f, tw), thr.y * fmaf(ta, 0.7f, tw), thr.z * fmaf(ta, 1.0f, tw), gpix, ss >> 24);
            } // else the sample's colour is zero: nothing to add (colors.rs:38)
            stage = 0;
        }

        // ---- refill: camera.rs:217-224.  Idle lanes (their path just ended, or the launch just began) take the
        // pool's next samples; their camera ray is made further down, so that its first Philox block is evaluated
        // together with the scattering lanes' first block ----------------------------------------------
        DRAY_CLOCK(3);
        DRAY_MARK("E_refill");
        bool need = stage == 0 && !camera;
        unsigned long long want = __ballot(need);
        DRAY_COUNT(6, want != 0ull ? 1 : 0);                       // iterations that refill
        // compaction switched off: the wave refills only as a whole, once its last path has ended
        if (a->no_refill != 0 && __ballot(stage > 0) != 0ull) want = 0ull;
        {
        const uint32_t W = (uint32_t)a->cam.W, spp = (uint32_t)a->cam.spp;
        // a hand-out of at most 64 samples crosses at most 64 / spp pixels (stepped over one by one, cheaper than
        // two integer divisions per lane down to 8 samples per pixel) and, with rows longer than a pool, one row end
        const bool stepwise = spp >= 8u && W >= (uint32_t)kSlots;
        // hands the open pool's next samples to the lanes that want one; returns the lanes still wanting
        auto hand_out = [&](unsigned long long want) -> unsigned long long {
            const uint32_t avail = pool_n - pool_q;
            const uint32_t wanted = (uint32_t)__popcll(want);
            const uint32_t take = wanted < avail ? wanted : avail;
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
            if (need && rank < take) {
                // the lane's place in the pool: pixel `dp` places after the pool's first, sample `smp` of it
                uint32_t dp, smp, r, i;
                if (stepwise) { // a few conditional steps instead of two integer divisions per hand-out
                    smp = pool_smp + rank;
                    dp = pool_dp;
                    while (smp >= spp) { smp -= spp; dp++; }
                    i = pool_i0 + dp;
                    r = pool_r0;
                    if (i >= W) { i -= W; r++; }
                } else {
                    const uint32_t q = pool_q + rank;
                    dp = q / spp;
                    smp = q - dp * spp;
                    const uint32_t pix = pool_pix + dp;
                    r = pix / W;
                    i = pix - r * W;
                }
                const int32_t pi = (int32_t)i, pj = a->part.row_begin + (int32_t)r * a->part.row_step;
                gpix = (uint32_t)pj * W + (uint32_t)pi;
                ss = smp | (((pool_slot + dp) & (uint32_t)(kSlots - 1)) << 24);
                ray.o = {(float)pi, (float)pj, 0.f}; // exact; the camera step below turns them into the ray
                camera = true;
                need = false;
            }
            pool_q += take;
            pool_smp += take;
            while (pool_smp >= spp) { pool_smp -= spp; pool_dp++; } // only read when stepwise
            return __ballot(need);
        };
        // the usual turn -- the open pool has samples left -- is straight-line code: the loop below, which may open
        // pools, carries every lane's new pixel, sample and ray origin through copies
        if (want != 0ull && pool_q != pool_n) want = hand_out(want);
        while (want != 0ull) {
            if (pool_q == pool_n && !exhausted) { // open the next pool
                const uint32_t npix = pixel_count(), pixels_per_pool = (uint32_t)a->pixels_per_pool;
                if (grab_next == grab_end) {
                    // One returning atomic on the global pixel counter grabs pixels for several pools: the counter
                    // sustains only ~80 M returning atomics per second (a 100-spp frame with one atomic per pixel
                    // was bound by exactly that), so a grab covers several pools (plan_render_f32 sizes it).  Grabs shrink as the
                    // queue drains (half the remaining pixels spread over all waves), which keeps the last waves'
                    // tail short.
                    unsigned long long base = 0ull;
                    if (lane == 0) base = atomicAdd(&a->counters->next_item, (unsigned long long)grab_pixels);
                    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)base, 0);
                    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(base >> 32), 0);
                    if (hi != 0u || lo >= npix) {
                        exhausted = true;
                    } else {
                        grab_next = lo;
                        grab_end = lo + min(grab_pixels, npix - lo);
                        const uint32_t fair = (npix - grab_end) / (2u * gridDim.x * (blockDim.x >> 6));
                        grab_pixels = max(pixels_per_pool, min((uint32_t)a->grab_pixels, fair));
                    }
                }
                if (!exhausted) {
                    const uint32_t lo = grab_next;
                    pool_slot = (pool_slot + pool_pixels) & (uint32_t)(kSlots - 1);
                    pool_pixels = min(pixels_per_pool, grab_end - lo);
                    grab_next = lo + pool_pixels;
                    pool_pix = lo;
                    pool_r0 = lo / W;
                    pool_i0 = lo - pool_r0 * W;
                    pool_q = pool_dp = pool_smp = 0u;
                    pool_n = pool_pixels * spp;
                    n_samples += (unsigned long long)pool_n;
                    retire_slots(pool_slot, lo, pool_pixels, true);
                }
            }
            if (exhausted) { // every idle lane gets here once the queue is empty: the wave then drains and exits
                if (need) stage = -1;
                break;
            }
            want = hand_out(want);
        }
        }
        if (__ballot(stage > 0 || camera) == 0ull) break;

        // ---- first uniforms of the lane's next step, drawn by the whole wave ---------------------------
        // A scattering lane starts with block 0 of its stage (Lambertian and Metal: first unit-vector attempt;
        // Dialectric: the reflect/refract uniform), a lane with a new sample with block 0 of stage 0 (pixel offsets,
        // first aperture attempt): one evaluation serves both.  The further attempts of the unit-vector rejection
        // sampler, and of the aperture's, are shared out over all 64 lanes, busy or not (coop_draw).
        DRAY_CLOCK(4);
        DRAY_MARK("F_philox_block0");
        const bool scatters = shading && !ends_now;
        const bool wants_vector = scatters && (mk == DRAY_MAT_LAMBERTIAN || mk == DRAY_MAT_METAL);
        const Stream st = {a->key.k0, a->key.k1, gpix, a->key.sample_offset + (ss & 0xFFFFFFu),
                           camera ? 0u : (uint32_t)stage};
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (camera || scatters) st.block(0, w);
        // the unit vector is drawn even when a metal's fuzz is 0 (material.rs:75); the aperture sample only with
        // defocus_angle > 0 (camera.rs:247)
        const uint32_t draw = wants_vector ? (uint32_t)kDrawUnitVector
                                           : (camera && a->cam.defocus != 0) ? (uint32_t)kDrawUnitDisk : (uint32_t)kDrawNone;
        DRAY_CLOCK(5);
        DRAY_MARK("G_coop_draw");
#ifdef DRAY_SECTION_COUNTERS
        uint32_t rounds = 0u;
        const F3 ruv = coop_draw(st, w, draw, lds_req[wv], lane, &rounds);
        DRAY_COUNT(7, rounds);                                     // cooperative draw rounds
        DRAY_COUNT(8, __ballot(camera || scatters) != 0ull ? 1 : 0); // iterations that evaluate block 0
        DRAY_COUNT(9, __popcll(__ballot(camera)));                 // camera rays made
#else
        const F3 ruv = coop_draw(st, w, draw, lds_req[wv], lane);
#endif
        DRAY_CLOCK(6);
        DRAY_MARK("H_scatter");

        // ---- scatter: material.rs ---------------------------------------------------------------------
        if (scatters) {
            const float dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
            const bool front = dot(ray.d, hn) < 0.f; // hittable.rs:16-22
            const F3 n = front ? hn : F3{-hn.x, -hn.y, -hn.z};
            F3 nd = {0.f, 0.f, 0.f};
            bool scattered = true;
            if (mk == DRAY_MAT_LAMBERTIAN) { // material.rs:46-56
                nd = n + ruv;
                if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
                thr = {thr.x * m.x, thr.y * m.y, thr.z * m.z};
            } else if (mk == DRAY_MAT_METAL) { // material.rs:73-80; reflect(d, n) is already unit
                const float dn = dot(ray.d, n);
                const F3 refl = {fmaf(-2.f * dn, n.x, dx), fmaf(-2.f * dn, n.y, dy), fmaf(-2.f * dn, n.z, dz)};
                nd = {fmaf(m.w, ruv.x, refl.x), fmaf(m.w, ruv.y, refl.y), fmaf(m.w, ruv.z, refl.z)};
                thr = {thr.x * m.x, thr.y * m.y, thr.z * m.z};
                scattered = dot(nd, n) > 0.f;
            } else { // Dialectric, material.rs:103-122
                const float ri = front ? rcp(m.w) : m.w;
                const float cos_theta = fminf(-dot(ray.d, n), 1.0f);
                const float sin_theta = fsqrt(fmaxf(fmaf(-cos_theta, cos_theta, 1.0f), 0.f));
                bool reflects = ri * sin_theta > 1.0f;
                if (!reflects) { // the uniform counts only when refraction is possible (material.rs:114)
                    float r0 = (1.f - ri) * rcp(1.f + ri);
                    r0 = r0 * r0;
                    const float x = 1.f - cos_theta, x2 = x * x;
                    reflects = fmaf(1.f - r0, x * x2 * x2, r0) > u01<float>(w[0]);
                }
                if (reflects) { // vec3.rs:97-102
                    nd = {fmaf(2.f * cos_theta, n.x, dx), fmaf(2.f * cos_theta, n.y, dy), fmaf(2.f * cos_theta, n.z, dz)};
                } else { // vec3.rs:104-109
                    const F3 perp = {ri * fmaf(cos_theta, n.x, dx), ri * fmaf(cos_theta, n.y, dy), ri * fmaf(cos_theta, n.z, dz)};
                    const float kpar = -fsqrt(fabsf(1.0f - dot(perp, perp)));
                    nd = {fmaf(kpar, n.x, perp.x), fmaf(kpar, n.y, perp.y), fmaf(kpar, n.z, perp.z)};
                }
            }
            if (scattered) { // camera.rs:271-274; depth is left for the next segment (see ends_now)
                stage++;
                ray.o = hp;
                set_dir(ray, nd);
                fresh = true;
            } else { // absorbed (camera.rs:276): the sample's colour is zero, nothing to add; idle until the next refill
                stage = 0;
            }
        }

        // ---- camera ray of a new sample: camera.rs:238-262 --------------------------------------------
        DRAY_CLOCK(7);
        DRAY_MARK("I_camera_ray");
        if (camera) {
            camera_ray(&a->cam, (int32_t)ray.o.x, (int32_t)ray.o.y, w, ruv, ray);
            thr = {1.f, 1.f, 1.f};
            stage = 1;
            fresh = true;
            camera = false;
        }
        DRAY_CLOCK(8);
    }

    DRAY_MARK("Z_exit");
#ifdef DRAY_SECTION_CLOCKS
    if (lane == 0)
        for (int q = 0; q < 16; ++q)
            if (clk[q]) atomicAdd(&dray_section_clocks[q], clk[q]);
#endif
    retire_slots(0u, 0u, (uint32_t)kSlots, false);
#ifdef DRAY_SECTION_COUNTERS
    if (lane == 0)
        for (int q = 0; q < 16; ++q)
            if (sec[q]) atomicAdd(&dray_section_counts[q], sec[q]);
#endif

    if (lane == 0) {
        atomicAdd(&a->counters->segments, n_seg);
        atomicAdd(&a->counters->samples, n_samples);
        if constexpr (SCAN == kScanGrid) atomicAdd(&a->counters->walk_steps, (unsigned long long)n_steps);
    }
}

// The kernels differ in the sphere scan and in the register budget that suits it (and in whether the list holds
// bounding boxes).  The argument block is only ever read through the kernel-argument segment pointer (see
// render_f32_body).
template <int SCAN, bool BOXES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_WAVES_PER_EU, DRAY_WAVES_PER_EU))) void render_f32_kernel(
    F32Args) {
    render_f32_body<SCAN, BOXES, false>();
}
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_GRID_WAVES_PER_EU, DRAY_GRID_WAVES_PER_EU))) void render_f32_kernel<kScanGrid, false>(
    F32Args) {
    render_f32_body<kScanGrid, false, false>();
}
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_GRID_WAVES_PER_EU, DRAY_GRID_WAVES_PER_EU))) void render_f32_kernel<kScanGrid, true>(
    F32Args) {
    render_f32_body<kScanGrid, true, false>();
}
// the grid scan over a copy of the grid staged in LDS (scenes without boxes whose compact tables fit)
template <>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_GRID_WAVES_PER_EU, DRAY_GRID_WAVES_PER_EU))) void render_f32_kernel<kScanGridLds, false>(
    F32Args) {
    render_f32_body<kScanGrid, false, true>();
}

// Adds the fixed-point sums into the caller's buffers and writes the packed colour: colors.rs:38-43.
__global__ __launch_bounds__(256) void resolve_f32_kernel(int32_t W, int32_t spp, Partition part,
                                                          const unsigned long long* __restrict__ accum,
                                                          uint32_t* __restrict__ color, double* __restrict__ raw,
                                                          int32_t* __restrict__ count) {
    const long long npix = (long long)part.n_rows * W;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int32_t r = (int32_t)(p / W), i = (int32_t)(p - (long long)r * W);
    const int32_t j = part.row_begin + r * part.row_step;
    const size_t out = (size_t)(part.compact ? r : j) * (size_t)W + (size_t)i;
    double s0 = (double)accum[0 * npix + p] * kFixInv;
    double s1 = (double)accum[1 * npix + p] * kFixInv;
    double s2 = (double)accum[2 * npix + p] * kFixInv;
    if (raw) {
        s0 += raw[3 * out + 0]; s1 += raw[3 * out + 1]; s2 += raw[3 * out + 2];
        raw[3 * out + 0] = s0; raw[3 * out + 1] = s1; raw[3 * out + 2] = s2;
    }
    const int32_t cnt = (count ? count[out] : 0) + spp;
    if (count) count[out] = cnt;
    const double denom = cnt != 0 ? (double)cnt : 1.;
    color[out] = pack_argb(s0 / denom, s1 / denom, s2 / denom);
}


// ---- the path step in f32: ray_color_iteration (camera.rs:118-144) for a batch, at the throughput kernel's precision ----
// One thread per entry.  The entry stays the reference's RayColorEntry in f64 (include/dray.h: dray_path_entry); the
// segment is intersected and scattered in f32 with the building blocks of the trace kernel above -- the origin's move
// up to the narrow spheres' ball, the grid walk with the exact test and the ordered key, the winner re-evaluated from
// the perpendicular remainder, wide spheres in the cancellation-free form or in f64, the materials of material.rs
// -- and with the same streams (stage k: the k-th scatter's draws), so that a path stepped here lands where the trace
// kernel's sample lands, to f32 rounding.  Lists with bounding boxes take the f64 step (launch_path_step_f32 refuses them).
struct StepArgs {
    const V4* geom32;
    const V4* mat32;
    const uint32_t* kind;
    const int32_t* wide_index;
    const double* wide_geom64;
    const V4* wide_axial;
    int32_t n_wide;
    float bound[4];
    GridDev grid;
    RngKey key;
    int32_t W;
    dray_path_entry* entries;
    unsigned long long n;
    DeviceCounters* counters;
};

__global__ __launch_bounds__(256) void path_step_f32_kernel(StepArgs) {
    ConstPtr<StepArgs> a = (ConstPtr<StepArgs>)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long my_segments = 0ull;
    if (k < a->n && !a->entries[k].finished) {
        dray_path_entry e = a->entries[k];
        F3 att = {(float)e.attenuation[0], (float)e.attenuation[1], (float)e.attenuation[2]};
        F3 c = {0.f, 0.f, 0.f};
        bool finished = true, hit_or_stop = true;
        if (e.depth > 0) { // else camera.rs:120-123: black
            my_segments = 1ull;
            PathRay ray;
            ray.o = {(float)e.origin[0], (float)e.origin[1], (float)e.origin[2]};
            set_dir(ray, F3{(float)e.direction[0], (float)e.direction[1], (float)e.direction[2]});
            { // the origin moves along the ray up to the ball of the narrow spheres (see the trace kernel)
                const float bx = a->bound[0] - ray.o.x, by = a->bound[1] - ray.o.y, bz = a->bound[2] - ray.o.z;
                const float s = fmaxf(fmaf(bz, ray.d.z, fmaf(by, ray.d.y, bx * ray.d.x)) - a->bound[3], 0.f);
                ray.o = {fmaf(s, ray.d.x, ray.o.x), fmaf(s, ray.d.y, ray.o.y), fmaf(s, ray.d.z, ray.o.z)};
                ray.tmin -= s;
            }
            PathRay nray = ray;
            nray.tmin = fmaxf(ray.tmin, 0.f);
            Walk walk = {};
            const GridRegs none = {0.f, 0.f, 0.f, a->grid.cells, nullptr, nullptr};
            bool walking = walk_begin(&a->grid, none, nray, walk);
            const GridRegs gr = {(float)a->grid.n[0], (float)a->grid.n[1], (float)a->grid.n[2], a->grid.cells,
                                 (const V4*)a->grid.entry_geom, a->grid.entry_index};
            while (walking) walking = walk_step(gr, nray, walk);
            // the hit: winner of the walk against the wide spheres (hittable_list.rs:60-74), point and normal
            const float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
            const float tmin = ray.tmin;
            float best = __uint_as_float((uint32_t)(walk.best >> 32));
            int32_t bidx = -1;
            F3 hp = {0.f, 0.f, 0.f}, hn = {0.f, 0.f, 0.f};
            if (walk.best != kNoHit) {
                bidx = (int32_t)(uint32_t)walk.best;
                const V4 s = a->geom32[bidx];
                const F3 oc = {s.x - ox, s.y - oy, s.z - oz};
                const float h = dot(oc, ray.d);
                const F3 l = {fmaf(-h, dx, oc.x), fmaf(-h, dy, oc.y), fmaf(-h, dz, oc.z)};
                const float sq = fsqrt(fmaxf(fmaf(s.w, s.w, -dot(l, l)), 0.f));
                best = h - sq; // sphere.rs:38-44
                if (!(best > tmin)) best = h + sq;
                hp = {fmaf(best, dx, ox), fmaf(best, dy, oy), fmaf(best, dz, oz)};
                const float ir = rcp(s.w);
                hn = {(hp.x - s.x) * ir, (hp.y - s.y) * ir, (hp.z - s.z) * ir}; // sphere.rs:52
            }
            for (int32_t wq = 0; wq < a->n_wide; ++wq) {
                const int32_t idx = a->wide_index[wq];
                const V4 g = a->wide_axial[2 * wq], kk = a->wide_axial[2 * wq + 1];
                float t = -1.f;
                F3 p = {0.f, 0.f, 0.f};
                if (kk.z != 0.f) { // by the y axis: f32 without cancellation (see the trace kernel)
                    const float ocx = g.x - ox, ocy = g.y - oy, ocz = g.z - oz;
                    const float h = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
                    const float cc = fmaf(kk.x - oy, kk.y - oy, fmaf(ocz, ocz, ocx * ocx));
                    const float disc = fmaf(h, h, -cc);
                    if (disc >= 0.f) {
                        const float sq = fsqrt(disc);
                        const bool up = h >= 0.f;
                        const float big = up ? h + sq : h - sq;
                        const float small = cc * rcp(big);
                        t = up ? small : big;
                        if (!(t > tmin)) t = up ? big : small;
                        if (!(t > tmin)) t = -INFINITY;
                        p = {fmaf(t, dx, ox), fmaf(t, dy, oy), fmaf(t, dz, oz)};
                    } else {
                        t = -INFINITY;
                    }
                } else { // sphere.rs:28-57 in f64 (see the trace kernel)
                    const double cx = a->wide_geom64[4 * wq + 0], cy = a->wide_geom64[4 * wq + 1], cz = a->wide_geom64[4 * wq + 2];
                    const double rad = a->wide_geom64[4 * wq + 3];
                    const double ocx = cx - (double)ox, ocy = cy - (double)oy, ocz = cz - (double)oz;
                    const double ddx = dx, ddy = dy, ddz = dz;
                    const double aa = fma(ddz, ddz, fma(ddy, ddy, ddx * ddx));
                    const double h = fma(ddz, ocz, fma(ddy, ocy, ddx * ocx));
                    const double cc = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -(rad * rad))));
                    const double disc = fma(h, h, -(aa * cc));
                    t = -INFINITY;
                    if (disc >= 0.) {
                        const float s0f = fsqrt((float)disc);
                        const double s0 = (double)s0f;
                        const double sq = s0f > 0.f ? fma(fma(-s0, s0, disc), (double)(0.5f * rcp(s0f)), s0) : 0.;
                        const double inva = 2.0 - aa;
                        double td = (h - sq) * inva;
                        if (!(td > (double)tmin)) td = (h + sq) * inva;
                        if (td > (double)tmin) {
                            t = (float)td;
                            p = {(float)fma(td, ddx, (double)ox), (float)fma(td, ddy, (double)oy), (float)fma(td, ddz, (double)oz)};
                        }
                    }
                }
                if (t > tmin && (t < best || (t == best && idx < bidx))) {
                    best = t;
                    bidx = idx;
                    hp = p;
                    hn = {(hp.x - g.x) * g.w, (hp.y - g.y) * g.w, (hp.z - g.z) * g.w}; // sphere.rs:52
                }
            }
            if (bidx < 0) { // camera.rs:140-143
                const float ta = 0.5f * (dy + 1.0f), tw = 1.0f - ta;
                c = {att.x * fmaf(ta, 0.5f, tw), att.y * fmaf(ta, 0.7f, tw), att.z * fmaf(ta, 1.0f, tw)};
                hit_or_stop = false;
            } else { // material.rs, as in the trace kernel
                const uint32_t mk = a->kind[bidx];
                const V4 m = a->mat32[bidx];
                const Stream st = {a->key.k0, a->key.k1, (uint32_t)e.px.pixel_j * (uint32_t)a->W + (uint32_t)e.px.pixel_i,
                                   a->key.sample_offset + (uint32_t)e.px.pixel_sample_num, e.stage};
                const bool front = dot(ray.d, hn) < 0.f; // hittable.rs:16-22
                const F3 n = front ? hn : F3{-hn.x, -hn.y, -hn.z};
                F3 nd = {0.f, 0.f, 0.f};
                bool scattered = false;
                if (mk == DRAY_MAT_LAMBERTIAN || mk == DRAY_MAT_METAL) {
                    F3 ruv = {0.f, 0.f, 0.f};
                    for (uint32_t at = 0;; ++at) { // vec3.rs:76-84
                        uint32_t w[4];
                        st.block(at, w);
                        if (unit_vector_attempt(w, ruv)) break;
                    }
                    if (mk == DRAY_MAT_LAMBERTIAN) { // material.rs:46-56
                        nd = n + ruv;
                        if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
                        scattered = true;
                    } else { // material.rs:73-80
                        const float dn = dot(ray.d, n);
                        const F3 refl = {fmaf(-2.f * dn, n.x, dx), fmaf(-2.f * dn, n.y, dy), fmaf(-2.f * dn, n.z, dz)};
                        nd = {fmaf(m.w, ruv.x, refl.x), fmaf(m.w, ruv.y, refl.y), fmaf(m.w, ruv.z, refl.z)};
                        scattered = dot(nd, n) > 0.f;
                    }
                    att = {att.x * m.x, att.y * m.y, att.z * m.z};
                } else if (mk == DRAY_MAT_DIELECTRIC) { // material.rs:103-122
                    uint32_t w[4];
                    st.block(0, w);
                    const float ri = front ? rcp(m.w) : m.w;
                    const float cos_theta = fminf(-dot(ray.d, n), 1.0f);
                    const float sin_theta = fsqrt(fmaxf(fmaf(-cos_theta, cos_theta, 1.0f), 0.f));
                    bool reflects = ri * sin_theta > 1.0f;
                    if (!reflects) {
                        float r0 = (1.f - ri) * rcp(1.f + ri);
                        r0 = r0 * r0;
                        const float x = 1.f - cos_theta, x2 = x * x;
                        reflects = fmaf(1.f - r0, x * x2 * x2, r0) > u01<float>(w[0]);
                    }
                    if (reflects) {
                        nd = {fmaf(2.f * cos_theta, n.x, dx), fmaf(2.f * cos_theta, n.y, dy), fmaf(2.f * cos_theta, n.z, dz)};
                    } else {
                        const F3 perp = {ri * fmaf(cos_theta, n.x, dx), ri * fmaf(cos_theta, n.y, dy), ri * fmaf(cos_theta, n.z, dz)};
                        const float kpar = -fsqrt(fabsf(1.0f - dot(perp, perp)));
                        nd = {fmaf(kpar, n.x, perp.x), fmaf(kpar, n.y, perp.y), fmaf(kpar, n.z, perp.z)};
                    }
                    scattered = true;
                } // else the absorber (material.rs:13-19; Transparent is rendered as one, as in the trace kernel)
                if (scattered) { // camera.rs:129-133
                    e.origin[0] = hp.x; e.origin[1] = hp.y; e.origin[2] = hp.z;
                    e.direction[0] = nd.x; e.direction[1] = nd.y; e.direction[2] = nd.z;
                    e.depth -= 1;
                    e.stage += 1u;
                    finished = false;
                } // else camera.rs:135: black
            }
        }
        e.attenuation[0] = att.x; e.attenuation[1] = att.y; e.attenuation[2] = att.z;
        if (finished) { e.color[0] = c.x; e.color[1] = c.y; e.color[2] = c.z; }
        e.finished = finished ? 1u : 0u;
        e.hit_object_or_stop = hit_or_stop ? 1u : 0u;
        a->entries[k] = e;
    }
    for (int off = 32; off > 0; off >>= 1) my_segments += __shfl_down(my_segments, off, 64);
    if ((threadIdx.x & 63) == 0 && a->counters && my_segments) atomicAdd(&a->counters->segments, my_segments);
}

} // namespace

template <int SCAN>
hipError_t occupancy_of(int* occ, int block, size_t lds) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, render_f32_kernel<SCAN, false>, block, lds);
}

F32Plan plan_render_f32(const CamDev<float>& cam, const Partition& part, int32_t pixels_per_pool, int32_t variant,
                        const SceneDev& scene, int device) {
    F32Plan plan{};
    plan.block = 256;
    const size_t table_bytes = (size_t)scene.n_narrow_padded * 8u; // cx, cy per sphere
    const bool hybrid_fits = table_bytes <= kHybridLdsLimit;
    if (variant == DRAY_SCAN_AUTO) // a grid walk pays once a ray would otherwise test a few dozen spheres
        // (recipe scenes at 100 spp, hybrid / grid: 8 spheres 3.8 / 4.4 ms, 20: 4.0 / 4.2, 40: 4.7-5.5 / 4.6, 66: 6.1 / 5.0)
        plan.scan = scene.n_narrow >= 40 ? DRAY_SCAN_GRID : (hybrid_fits ? DRAY_SCAN_HYBRID : DRAY_SCAN_SCALAR);
    else if (variant == DRAY_SCAN_HYBRID)
        plan.scan = hybrid_fits ? DRAY_SCAN_HYBRID : DRAY_SCAN_SCALAR;
    else
        plan.scan = variant;
    plan.lds_bytes = plan.scan == DRAY_SCAN_HYBRID ? table_bytes : 0u;
    plan.lds_grid = 0;
    if (plan.scan == DRAY_SCAN_GRID && scene.n_boxes == 0 && scene.grid.n[0] > 0) {
        // the compact copy of the grid (GridLds): staged when six such workgroups -- the kernel's residency -- fit a
        // compute unit's LDS beside their static arrays, and the packed fields hold the grid's numbers
        const size_t n_cells = (size_t)scene.grid.n[0] * (size_t)scene.grid.n[1] * (size_t)scene.grid.n[2];
        const size_t bytes = (size_t)scene.n * 16u + n_cells * 4u + ((scene.grid_entries * 2u + 15u) & ~(size_t)15u);
        const char* env = std::getenv("DRAY_GRID_LDS"); // "0": the tables are read from memory (tests, tuning runs)
        int occ_lds = 0;
        if (!(env && env[0] == '0') && scene.n <= 65535 && scene.grid_entries < (1u << kLdsBeginBits) &&
            scene.grid_max_cell_entries < (1u << kLdsCountBits) && bytes <= kGridLdsLimit &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_lds, render_f32_kernel<kScanGridLds, false>, 256, bytes) == hipSuccess &&
            occ_lds >= DRAY_GRID_WAVES_PER_EU) {
            plan.lds_grid = 1;
            plan.lds_bytes = bytes;
            plan.lds_occupancy = occ_lds;
        }
    }
    // compute units of the device and resident blocks per unit of the two kernels without dynamic LDS: asked once per device
    struct DeviceFacts { int cus = 0, occ_scalar = 0, occ_grid = 0; };
    static DeviceFacts facts[64];
    static std::mutex facts_mu;
    int cus = 256, per_cu = 4, occ = 0;
    {
        std::lock_guard<std::mutex> lock(facts_mu);
        DeviceFacts local;
        DeviceFacts& f = (device >= 0 && device < 64) ? facts[device] : local;
        if (f.cus == 0) {
            int v = 0;
            f.cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) ? v : 256;
            if (occupancy_of<kScanScalar>(&v, 256, 0) == hipSuccess && v > 0) f.occ_scalar = v;
            if (occupancy_of<kScanGrid>(&v, 256, 0) == hipSuccess && v > 0) f.occ_grid = v;
        }
        cus = f.cus;
        occ = plan.scan == DRAY_SCAN_GRID ? f.occ_grid : plan.scan == DRAY_SCAN_SCALAR ? f.occ_scalar : 0;
    }
    if (plan.scan == DRAY_SCAN_HYBRID && occupancy_of<kScanHybrid>(&occ, plan.block, plan.lds_bytes) != hipSuccess) occ = 0;
    if (plan.lds_grid) occ = plan.lds_occupancy;
    if (occ > 0) per_cu = occ;
    if (const char* env = std::getenv("DRAY_BLOCKS_PER_CU")) { // tuning runs only: fewer resident workgroups than fit
        const int v = std::atoi(env);
        if (v > 0 && v < per_cu) per_cu = v;
    }
    plan.grid = cus * per_cu;
    const long long npix = (long long)part.n_rows * cam.W;
    const int32_t spp = cam.spp > 0 ? cam.spp : 1;
    // the first round of grabs may take a sixteenth of the launch at most (see below)
    const long long sixteenth = (long long)(part.n_rows > 0 ? part.n_rows : 0) * cam.W * spp /
                                (16LL * plan.grid * (plan.block / 64));
    int32_t ppp = (kPoolSamples + spp - 1) / spp; // >= kPoolSamples samples ...
    if (plan.scan == DRAY_SCAN_GRID || scene.n_narrow_padded <= 64) {
        // Where samples are cheap (the grid scan, a list of a few spheres) a pool of some 800 to 1 000 samples is the
        // better unit: a pool is what a wave asks the queue for at least, it waits for the answer, and opening one costs
        // some 200 instructions.  1200x675 frame, grid scan, ms at 1 / 2 / 4 / 8 pixels per pool: 500 spp 26.37 / 25.96 /
        // 26.26 / -, 250 spp 13.53 / 13.39 / 13.26 / -, 100 spp 5.83 / 5.74 / 5.72 / 5.76, 50 spp (2 / 4 / 8 / 12 / 16
        // pixels) 3.22 / 3.23 / 3.14 / 3.16 / 3.37; three spheres at 1920x1080x100 spp (1 / 2 / 4 / 8 / 10 / 12 / 16): 8.56 /
        // 8.41 / 8.80 / 8.07 / 8.08 / 8.14 / 8.25.  A small launch keeps small pools.
        int32_t want = (800 + spp - 1) / spp;
        if (want > 12) want = 12;
        const int32_t fits = (int32_t)(sixteenth / spp < 12 ? sixteenth / spp : 12);
        if (want > fits) want = fits;
        if (want > ppp) ppp = want;
    }
    if (pixels_per_pool > 0) ppp = pixels_per_pool;
    if (ppp > kSlots / 2) ppp = kSlots / 2; // ... while the previous pool's pixels stay cached
    plan.pixels_per_pool = ppp;
    // Cheap samples drain the queue fast and need big grabs to stay under the counter's atomic rate; dear ones
    // (a long brute-force scan per segment) need small grabs to keep the last waves' tail short.  Measured: 3-sphere
    // full-HD frame 13.2 / 9.6 / 9.3 ms at 256 / 512 / 1024 samples per grab, 485-sphere hybrid scan 21.0 / 22.1 /
    // 25.4 ms, grid scan 9.7 / 9.5 / 9.8 ms (100 spp).
    // With the grid scan (a segment costs about as much as 8 tests plus the 16 of shading): 1200x675 frame of 125 /
    // 250 / 500 spp 7.80 / 14.86 / 27.21 ms at 341 samples per grab, 7.36 / 14.03 / 27.24 ms at 700; one rank's
    // eighth of it 1.49 / 2.31 / 3.91 against 1.89 / 2.40 / 3.97 ms -- a small launch needs small grabs, so the first
    // round of grabs may take a sixteenth of the launch at most.
    const int32_t tests_per_segment = plan.scan == DRAY_SCAN_GRID ? 8 : scene.n_narrow_padded;
    int32_t grab_samples = kGrabWork / (tests_per_segment + 16);
    // (at least 256: a 66-sphere scene at 100 spp took 10.3 ms with the hybrid scan at one pixel per grab, bound by
    // the counter, and 6.2 ms at two)
    grab_samples = grab_samples < 256 ? 256 : (grab_samples > 1024 ? 1024 : grab_samples);
    // ... but not below 256 samples: at the grid scan's 15 G samples/s, 128-sample grabs ask the counter for more
    // than the 80 M returning atomics per second it serves (1200x675x10 spp: 1.81 ms at 70 samples per grab, 1.15 ms at 210)
    const int32_t least = grab_samples < 256 ? grab_samples : 256;
    if (sixteenth < grab_samples) grab_samples = sixteenth < least ? least : (int32_t)sixteenth;
    if (const char* env = std::getenv("DRAY_GRAB_SAMPLES")) { // tuning runs only
        const int v = std::atoi(env);
        if (v > 0) grab_samples = v;
    }
    int32_t pools = grab_samples / (ppp * spp);
    plan.grab_pixels = ppp * (pools < 1 ? 1 : (pools > 64 ? 64 : pools));
    plan.accum_words = (size_t)3 * (size_t)(npix > 0 ? npix : 0);
    const long long blocks_needed = (npix * spp + plan.block - 1) / plan.block;
    if (blocks_needed < plan.grid) plan.grid = (int32_t)(blocks_needed > 0 ? blocks_needed : 1);
    return plan;
}

hipError_t launch_render_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part, const RngKey& key,
                             const F32Plan& plan, unsigned long long* d_accum, DeviceCounters* d_counters,
                             hipStream_t stream) {
    const long long npix = (long long)part.n_rows * cam.W;
    if (npix <= 0 || cam.spp <= 0) return hipSuccess;
    if (npix > 0x7fffffffLL / 3) return hipErrorInvalidValue; // the kernel indexes the three sum planes with 32 bits
    if (cam.spp > 0xFFFFFF) return hipErrorInvalidValue;     // a lane keeps its sample index in 24 bits
    const F32Tables& t = scene.f32;
    F32Args args{};
    args.cam = cam;
    args.narrow_geom = (const V4*)t.narrow_geom; args.filter_xy = (const V4*)t.filter_xy; args.filter_zn = (const V4*)t.filter_zn;
    args.narrow_index = t.narrow_index; args.wide_index = t.wide_index;
    args.geom64 = t.geom64; args.wide_geom64 = t.wide_geom64; args.wide_axial = (const V4*)t.wide_axial; args.geom32 = (const V4*)t.geom32; args.mat32 = (const V4*)t.mat32; args.kind = t.kind;
    args.n_narrow_padded = t.n_narrow_padded; args.n_wide = t.n_wide;
    args.box64 = scene.box64; args.box_index = scene.box_index; args.n_boxes = scene.n_boxes;
    for (int q = 0; q < 4; ++q) args.bound[q] = t.bound[q];
    args.grid = scene.grid;
    args.part = part; args.key = key; args.pixels_per_pool = plan.pixels_per_pool; args.grab_pixels = plan.grab_pixels;
    args.no_refill = plan.no_refill;
    args.n_cells = scene.grid.n[0] * scene.grid.n[1] * scene.grid.n[2];
    args.n_entries = (int32_t)scene.grid_entries;
    args.n_list = scene.n;
    args.accum = d_accum; args.counters = d_counters;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(plan.grid), dim3(plan.block), plan.lds_bytes, stream, args);
    };
    const bool boxes = scene.n_boxes > 0;
    if (plan.scan == DRAY_SCAN_HYBRID)
        boxes ? launch(render_f32_kernel<kScanHybrid, true>) : launch(render_f32_kernel<kScanHybrid, false>);
    else if (plan.scan == DRAY_SCAN_GRID && plan.lds_grid)
        launch(render_f32_kernel<kScanGridLds, false>);
    else if (plan.scan == DRAY_SCAN_GRID)
        boxes ? launch(render_f32_kernel<kScanGrid, true>) : launch(render_f32_kernel<kScanGrid, false>);
    else
        boxes ? launch(render_f32_kernel<kScanScalar, true>) : launch(render_f32_kernel<kScanScalar, false>);
    return hipGetLastError();
}

#ifdef DRAY_SECTION_CLOCKS
// Diagnostic builds only: reads and clears the section clocks.
extern "C" int dray_debug_section_clocks(unsigned long long* out) {
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dray_section_clocks), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(dray_section_clocks), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

#ifdef DRAY_SECTION_COUNTERS
// Diagnostic builds only: reads and clears the section counters.
extern "C" int dray_debug_section_counts(unsigned long long* out) {
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dray_section_counts), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(dray_section_counts), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_resolve_f32(const CamDev<float>& cam, const Partition& part, const unsigned long long* d_accum,
                              uint32_t* d_color, double* d_raw, int32_t* d_count, hipStream_t stream) {
    const long long npix = (long long)part.n_rows * cam.W;
    if (npix <= 0 || cam.spp <= 0) return hipSuccess;
#ifndef DRAY_RESOLVE_BLOCK
#define DRAY_RESOLVE_BLOCK 256
#endif
    const int block = DRAY_RESOLVE_BLOCK;
    const long long grid = (npix + block - 1) / block;
    hipLaunchKernelGGL(resolve_f32_kernel, dim3((unsigned)grid), dim3(block), 0, stream, cam.W, cam.spp, part, d_accum,
                       d_color, d_raw, d_count);
    return hipGetLastError();
}

// The f32 path step (path_step_f32_kernel).  Lists with bounding boxes are not for it: hipErrorNotSupported, and the
// caller takes the f64 step.
hipError_t launch_path_step_f32(const SceneDev& scene, const RngKey& key, int32_t image_width, dray_path_entry* d_entries,
                                size_t n, DeviceCounters* d_counters, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (scene.n_boxes > 0) return hipErrorNotSupported;
    const size_t grid = (n + 255) / 256;
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    const F32Tables& t = scene.f32;
    StepArgs args{};
    args.geom32 = (const V4*)t.geom32; args.mat32 = (const V4*)t.mat32; args.kind = t.kind;
    args.wide_index = t.wide_index; args.wide_geom64 = t.wide_geom64; args.wide_axial = (const V4*)t.wide_axial;
    args.n_wide = t.n_wide;
    for (int q = 0; q < 4; ++q) args.bound[q] = t.bound[q];
    args.grid = scene.grid;
    args.key = key; args.W = image_width; args.entries = d_entries; args.n = n; args.counters = d_counters;
    hipLaunchKernelGGL(path_step_f32_kernel, dim3((unsigned)grid), dim3(256), 0, stream, args);
    return hipGetLastError();
}

} // namespace dray
