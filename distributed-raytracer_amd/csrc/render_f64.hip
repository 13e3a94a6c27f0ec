// The f64 render as a stream of samples, for gfx950: the default kernel of DRAY_MODE_F64.  The reference's arithmetic
// in the reference's order (ref_device.h), so that sums, colours and segment counts equal the CPU oracle's bit for
// bit.  Build with -ffp-contract=off.
//
// render_ref_kernel (render_ref.hip) gives every pixel a thread that runs the pixel's loop on its own.  Its lanes wait
// for each other wherever their trip counts differ -- the grid walk (one to dozens of cells), a cell's sphere list,
// the rejection samplers -- and a wave lives as long as its dearest pixel takes for all of its samples, so that the
// machine is half empty for much of a launch.  The kernel below computes the same values and adds them in the same
// order, organised like the f32 throughput kernel (render_f32.hip):
//   * one persistent launch; a wave draws chunks of consecutive pixels from a global queue and hands their samples,
//     in order, to whichever of its lanes is free: a lane carries one sample's path;
//   * a lane's walk through the grid is kept in registers across turns of the main loop; the wave takes walk steps
//     (a step takes a lane out of a finished cell into the next, and tests one listed sphere) only while enough
//     lanes are walking, then the lanes whose walk is over are shaded and given their next ray while the long walks
//     go on;
//   * the two rejection samplers (vec3.rs:76-84, 111-118) are evaluated by the whole wave: lanes whose first attempt
//     failed hand their stream to server lanes, which try the following blocks in parallel; the lane takes the
//     accepted attempt with the smallest index -- the one its own loop would have stopped at;
//   * the first Philox block of new samples and of the scattering lanes' stage is one evaluation.
// Samples end out of order, and f64 sums depend on the order (colors.rs:38 adds a pixel's samples one after the
// other), so a finished sample's colour goes into the wave's ring in memory, at the place of its number in the
// wave's stream, and the wave retires the ring in stream order: the pixel's running sum takes the colours of
// samples 0, 1, 2, ... exactly as the per-pixel loop does.  A lane may not take a sample more than a ring's length
// ahead of the oldest unfinished one.
// Closest hit: as list_hit_grid in ref_device.h (minimum over (root, list position) of sphere_root's roots over the
// wide, the loose and the walked spheres).  A ray that is not finite, or a walk that outlasts a straight line, takes
// the scan of the whole list, as there.
#pragma clang fp contract(off)

#include <cmath>
#include <cstdlib>
#include <mutex>

#include "ref_device.h"

namespace dray {
namespace {

// Pointer into the constant address space: a wave-uniform read through it is a scalar load.
#define DRAY_CONSTANT __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const DRAY_CONSTANT T* as_constant(const T* p) {
    return (const DRAY_CONSTANT T*)p;
}

// lane `k`'s value (k the same in every lane)
__device__ __forceinline__ double readlane_f64(double v, int k) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, k);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

enum { kDraw64None = 0, kDraw64UnitVector = 1, kDraw64UnitDisk = 2 };
#ifndef DRAY_F64_WALK_QUORUM
#define DRAY_F64_WALK_QUORUM 8
#endif

struct Drawn {
    uint32_t a, b, c; // the accepted attempt's words: a unit vector's three, or a disk point's two
};

// Whether the reference keeps an attempt of its rejection samplers, decided in f32 wherever f32 can tell.  An attempt's
// coordinates are -1 + 2 u with u = k 2^-24, exact in f32 as in f64, and the f64 sum of their squares is exact too
// (integers below 2^49 in units of 2^-46), so the reference's comparisons (vec3.rs:80: 1e-160 <= |p|^2 <= 1; vec3.rs:114:
// |p|^2 < 1) are comparisons of the true sum, which the f32 sum misses by at most 2^-22 of itself.  `unsure`: the f32 sum is
// too close to a bound to tell (one attempt in some 10^6); the caller then has the f64 expressions decide.
__device__ __forceinline__ bool vector_kept_f32(uint32_t w0, uint32_t w1, uint32_t w2, bool& unsure) {
    const float x = fmaf(2.f, u01<float>(w0), -1.f), y = fmaf(2.f, u01<float>(w1), -1.f), z = fmaf(2.f, u01<float>(w2), -1.f);
    const float l2 = fmaf(z, z, fmaf(y, y, x * x));
    unsure = (fabsf(l2 - 1.f) < 1e-6f) | (l2 < 1e-12f);
    return l2 <= 1.f;
}
__device__ __forceinline__ bool disk_kept_f32(uint32_t w0, uint32_t w1, bool& unsure) {
    const float x = fmaf(2.f, u01<float>(w0), -1.f), y = fmaf(2.f, u01<float>(w1), -1.f);
    const float l2 = fmaf(y, y, x * x);
    unsure = fabsf(l2 - 1.f) < 1e-6f;
    return l2 < 1.f;
}

// Both rejection samplers for all lanes of a wave (wave-uniform control flow; every lane calls).  `w` is block 0 of
// the lane's stream, `draw` what the lane needs.  Acceptance is the reference's (see above): the f64 expressions decide
// whenever f32 cannot.
__device__ __forceinline__ Drawn coop_draw64(const Stream& st, const uint32_t (&w)[4], uint32_t draw, uint4* req, int lane) {
    Drawn res = {w[0], w[1], w[2]};
    bool pending = false;
    {
        bool unsure_v, unsure_d;
        const bool kept_v = vector_kept_f32(w[0], w[1], w[2], unsure_v), kept_d = disk_kept_f32(w[2], w[3], unsure_d);
        if (draw == kDraw64UnitVector) {
            pending = !kept_v;
        } else if (draw == kDraw64UnitDisk) { // attempt 0 of the aperture sample: uniforms 2, 3 of stage 0
            res = {w[2], w[3], 0u};
            pending = !kept_d;
        }
        if (__ballot((draw == kDraw64UnitVector && unsure_v) || (draw == kDraw64UnitDisk && unsure_d)) != 0ull) {
            V3<double> p;
            double l2, qx, qy;
            if (draw == kDraw64UnitVector) pending = !unit_vector_attempt<double>(w, p, l2);
            else if (draw == kDraw64UnitDisk) pending = !unit_disk_attempt<double>(w[2], w[3], qx, qy);
        }
    }
    uint32_t base = 1u; // next block index, the same for every pending lane
    // one round: the pending lanes `pm` are served blocks base .. base + k - 1 of their streams
    auto round = [&](unsigned long long pm) {
        const uint32_t n_pending = (uint32_t)__popcll(pm);
        const uint32_t shift = n_pending <= 8u ? 3u : n_pending <= 16u ? 2u : n_pending <= 32u ? 1u : 0u; // log2 servers per lane
        const uint32_t rank = (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
        if (pending) req[rank] = make_uint4(st.pixel, st.sample, st.stage, draw);
        const uint32_t served = (uint32_t)lane >> shift;
        const bool serving = served < n_pending;
        const uint4 r = req[serving ? served : 0u];
        const Stream other = {st.k0, st.k1, r.x, r.y, r.z};
        uint32_t wb[4];
        other.block(base + ((uint32_t)lane & ((1u << shift) - 1u)), wb);
        const bool disk = r.w == kDraw64UnitDisk;
        bool unsure_v, unsure_a, unsure_b;
        bool ok_vector = vector_kept_f32(wb[0], wb[1], wb[2], unsure_v);
        // a block after the first holds two aperture attempts: uniforms 4b, 4b+1 and 4b+2, 4b+3
        bool ok_a = disk_kept_f32(wb[0], wb[1], unsure_a), ok_b = disk_kept_f32(wb[2], wb[3], unsure_b);
        if (__ballot(serving && (disk ? (unsure_a || unsure_b) : unsure_v)) != 0ull) {
            V3<double> p;
            double l2, qx, qy;
            ok_vector = unit_vector_attempt<double>(wb, p, l2);
            ok_a = unit_disk_attempt<double>(wb[0], wb[1], qx, qy);
            ok_b = unit_disk_attempt<double>(wb[2], wb[3], qx, qy);
        }
        const uint32_t va = disk ? (ok_a ? wb[0] : wb[2]) : wb[0], vb = disk ? (ok_a ? wb[1] : wb[3]) : wb[1], vc = wb[2];
        const bool ok = serving & (disk ? (ok_a | ok_b) : ok_vector);
        const unsigned long long okm = __ballot(ok);
        const uint32_t first = pending ? (rank << shift) : 0u; // my first server lane
        const uint32_t mine = (uint32_t)(okm >> first) & ((1u << (1u << shift)) - 1u);
        const bool got = pending & (mine != 0u);
        const int from = got ? (int)(first + (uint32_t)__builtin_ctz(mine)) : lane;
        const uint32_t ga = __shfl(va, from, 64), gb = __shfl(vb, from, 64), gc = __shfl(vc, from, 64);
        if (got) {
            res = {ga, gb, gc};
            pending = false;
        }
        base += 1u << shift;
    };
    // (the first round, which nearly every call needs, is straight-line code: a loop carries the kept words through copies)
    unsigned long long pm = __ballot(pending);
    if (pm != 0ull) {
        round(pm);
        for (pm = __ballot(pending); pm != 0ull; pm = __ballot(pending)) round(pm);
    }
    return res;
}

// ---- closest hit: a conservative f32 search in front of the reference's f64 test --------------------------------
// The reference's scan returns the minimum over (root, list position) of the roots sphere.rs:29-48 accepts.  Which
// spheres can hold that minimum is found in f32 -- a walk through the grid with a test that brackets every sphere's
// root between bounds no rounding can cross -- and only those (one or two per ray) take the reference's own f64
// expressions (sphere_root_of), so root, point and normal are bit for bit the linear scan's.
//
// The f32 view of a ray: origin moved along the ray up to the ball that holds the narrow spheres (so that its
// coordinates are of the scene's size), unit direction, and the reference's lower bound 0.001 |d| in arc length from
// that origin.  For a sphere (centre c~, squared radius r2~, both rounded to f32) the test forms oc = c~ - o~, h = oc.u~,
// disc = h^2 + r2~ - |oc|^2 and brackets the true values (those of the unrounded ray and sphere in real arithmetic, which
// the reference's f64 results match to ~2^-50 of the same magnitudes):
//   |disc - disc*| <= E  = 2^-18 (|oc|^2 + r2) + 2^-19 |o~| |oc| + 2^-26 |o~|^2      (measured bound: 26, 14 and 0 units of 2^-24)
//   |h - h*|       <= Eh = 2^-20 (|oc| + |o~| + sqrt(disc + E))                       (measured bound: 8 and 4 units of 2^-24)
// with |o~| enlarged by 2^-9 of the distance the origin was moved, which covers the f64 rounding of the reference's own
// expressions seen from the unmoved origin (2^-50 of that distance squared in disc, 2^-50 of it in h).  The test uses
// them in a form that needs no |oc| (a square root per sphere): |o~| |oc| <= (|o~|^2 + |oc|^2) / 2 and |oc| <= |o~| + C,
// C the largest |c~| of the narrow spheres, so that
//   E  <= (2^-18 + 2^-20) (|oc|^2 + r2) + (2^-20 + 2^-26) |o~|^2,     Eh <= 2^-20 (2 |o~| + C + sqrt(disc + E)),
// each a multiple of what the sphere contributes plus a term of the ray alone (`e0`, `eh0`).  Hence
//   near root in [h - sqrt(disc + E) - Eh, h - sqrt(disc - E) + Eh],  far root in [h + sqrt(disc - E) - Eh, h + sqrt(disc + E) + Eh],
// a sphere with disc + E < 0 or far bound below the lower bound is missed for certain, one with disc - E > 0 and a root
// bound above the lower bound is hit for certain and caps the search (`ub`); everything in between stays a candidate.
// Values that are not finite fail every comparison and so stay candidates too.
struct Filter32 {
    float ox, oy, oz, ux, uy, uz;
    float tm_lo, tm_hi; // the reference's lower bound on a root (interval.rs:28-30), a hair below and above
    float e0, eh0;      // the ray's own share of E and of Eh (filter_consts)
};
// The ray's share of the error bounds, from `on` = |o~| + 2^-9 of the move and the narrow spheres' largest |c~|.
// (The floors 1e-30 and 1e-15 stand above anything f32 underflow can do to a scene of the sizes the search is used for.)
__device__ __forceinline__ void filter_consts(Filter32& f, float on, float cmax) {
    f.e0 = fmaf(9.685754776000977e-07f * on, on, 1e-30f);
    f.eh0 = fmaf(9.5367431640625e-07f, fmaf(2.f, on, cmax), 1e-15f);
}
constexpr uint32_t kNoCand = 0xFFFFFFFFu;
struct Cand {
    float ub;        // an upper bound on the closest accepted root, from spheres hit for certain
    float lo0;       // a lower bound on the root of the sphere in the first slot (NaN: the slot is free)
    uint32_t i0, i1; // list positions of the spheres that may hold the closest root (kNoCand: free slot)
};

// A sphere's root brackets against the search so far: the near root lies in [near_lo, near_hi], the far root in
// [far_lo, far_hi].  What is not known is NaN, which fails every comparison: far_hi NaN -- no root for certain (also a
// sphere that is not finite, which sphere.rs:33-36 never hits: its discriminant is NaN or negative) --, near_hi and
// far_lo NaN -- not known to have two roots.  A sphere that is hit for certain lowers `ub`; the sphere in the first slot
// leaves it once its lower bound lies beyond `ub` -- its root is then farther than another sphere's --, so that a ray
// nearly always ends its search with one sphere to take the reference's test, and the second slot is for roots the
// brackets cannot tell apart.  (Compares and selects cost a SIMD almost twice what an addition does: the nine and six
// here are what is left of fourteen and seven.)  Returns true when the sphere is a candidate and both slots are taken:
// the caller has the slots evaluated and tests the sphere again.
__device__ __forceinline__ bool slot_update(float tm_lo, float tm_hi, float near_lo, float near_hi, float far_lo, float far_hi,
                                            uint32_t k, Cand& c) {
    const bool hit = far_hi >= tm_lo;
    const float lo = near_hi < tm_lo ? far_lo : near_lo;
    const bool certain = far_lo > tm_hi; // two roots, and the far one is accepted if the near one is not
    const float hi = near_lo > tm_hi ? near_hi : far_hi;
    c.ub = fminf(c.ub, certain ? hi : INFINITY);
    // a sphere met again (it is listed in several cells) is not taken twice
    const bool keep = hit & !(lo > c.ub) & (k != c.i0) & (k != c.i1);
    const bool take0 = keep & !(c.lo0 <= c.ub); // a free slot (lo0 NaN), or one whose sphere has just fallen behind
    const bool free1 = c.i1 == kNoCand;
    c.i1 = (keep & !take0 & free1) ? k : c.i1;
    c.i0 = take0 ? k : c.i0;
    c.lo0 = take0 ? lo : c.lo0;
    return keep & !take0 & !free1;
}

// One narrow sphere against the search so far.
__device__ __forceinline__ bool filter_test(const Filter32& f, float cx, float cy, float cz, float r2, uint32_t k, Cand& c) {
    const float ocx = cx - f.ox, ocy = cy - f.oy, ocz = cz - f.oz;
    const float h = fmaf(ocz, f.uz, fmaf(ocy, f.uy, ocx * f.ux));
    const float oc2 = fmaf(ocz, ocz, fmaf(ocy, ocy, ocx * ocx));
    const float disc = fmaf(h, h, r2 - oc2);
    const float e = fmaf(4.76837158203125e-06f, oc2 + r2, f.e0);
    const float dhi = disc + e, dlo = disc - e;
    const float sq_hi = __builtin_amdgcn_sqrtf(dhi), sq_lo = __builtin_amdgcn_sqrtf(dlo); // NaN: no root for certain, not two for certain
    const float eh = fmaf(9.5367431640625e-07f, sq_hi, f.eh0);
    const float wide = sq_hi + eh, tight = sq_lo - eh; // the brackets' half-widths about h: outer and inner
    return slot_update(f.tm_lo, f.tm_hi, h - wide, h - tight, h + tight, h + wide, k, c);
}

// One wide sphere: in f32 its |oc|^2 - r^2 cancels, and a bracket as wide as f32 leaves it could not tell the root 0 of
// a ray that starts on the sphere from the reference's lower bound.  So oc, h = oc.u, disc = h^2 + r^2 - |oc|^2, its
// square root (v_rsq_f64 and one Newton step) and the two roots h -+ sqrt are formed in f64, from the ray as the
// reference has it -- origin `ro`, the unit direction `u` (good to 2^-40), roots counted from the f32 view's origin, `s0`
// along the ray -- and rounded to f32 at the end.  Errors: u's 2^-40 and the f64 roundings leave disc within
// E = 2^-36 (|oc|^2 + r^2) of the true value (the reference's own discriminant within 2^-48 of the same), which moves
// the square root by at most E / max(sqrt, sqrt E); h and the sums are good to 2^-36 (|oc| + sqrt); rounding a root to
// f32 adds 2^-24 of its size, and 2^-20 |o~| is the narrow spheres' allowance for the rounded origin.
__device__ __forceinline__ bool wide_test(float tm_lo, float tm_hi, float on, V3<double> ro, V3<double> u, double s0, double cx,
                                          double cy, double cz, double radius, uint32_t k, Cand& c) {
    const double ocx = cx - ro.x, ocy = cy - ro.y, ocz = cz - ro.z;
    const double h = __builtin_fma(ocz, u.z, __builtin_fma(ocy, u.y, ocx * u.x));
    const double oc2 = __builtin_fma(ocz, ocz, __builtin_fma(ocy, ocy, ocx * ocx));
    const double r2 = radius * radius;
    const double disc = __builtin_fma(h, h, r2 - oc2);
    const double dpos = fmax(disc, 1e-300);
    const double y = __builtin_amdgcn_rsq(dpos);
    double sq = dpos * y;
    sq = __builtin_fma(__builtin_fma(-sq, sq, dpos), 0.5 * y, sq);
    const double hs = h - s0;
    const float near = (float)(hs - sq), far = (float)(hs + sq);
    const float size2 = (float)(oc2 + r2), sqf = (float)sq;
    const float e = fmaf(1.4551915228366852e-11f, size2, 1e-37f); // E
    const float se = __builtin_amdgcn_sqrtf(e);
    const float dsq = e * __builtin_amdgcn_rcpf(fmaxf(sqf, se)) * 1.0001f;
    const float size = __builtin_amdgcn_sqrtf(size2);
    const float eps = fmaf(9.5367431640625e-07f, on, fmaf(1.4551915228366852e-11f, size + sqf, dsq));
    const float en = fmaf(1.1920928955078125e-07f, fabsf(near), eps), ef = fmaf(1.1920928955078125e-07f, fabsf(far), eps);
    const float disc_f = (float)disc;
    const bool never = disc_f < -e, real = disc_f > e * 1.0001f + fabsf(disc_f) * 1.1920928955078125e-07f;
    return slot_update(tm_lo, tm_hi, near - en, real ? near + en : NAN, real ? far - ef : NAN, never ? NAN : far + ef, k, c);
}

// A lane's walk through the grid (3-D DDA along the f32 ray), kept across turns of the main loop.  Boundary times are
// recomputed from the cell index at every step, never accumulated; the walk stops when `ub` lies before the current
// cell's exit by more than the exit time's rounding error.  A sphere is listed in every cell its ball grown by 1/256
// cell (and 1e-5 of the scene's reach) overlaps, which the f32 rounding of ray and walk cannot bridge.
struct Walk32 {
    float c[3];        // current cell (small integers, exact in f32)
    uint32_t e, e_end; // entries of the current cell still to test
};
// What stays the same for the length of a ray -- the f32 view of the ray and the walk's boundary times (the ray leaves
// cell k of an axis at fma(k, A, B); `slack` bounds the rounding error of those times) -- is kept in LDS, sixteen words
// per lane, and read at the head of every turn's steps: registers are what this kernel is short of (the shading
// sections between two turns' steps need none of it).
struct RayConst {
    float A[3], B[3], slack;
    Filter32 f;
};

// Diagnostic builds only (tools/f64_clocks.py): with DRAY_F64_CLOCKS a wave adds up the shader clocks it spends in
// each section of its main loop, and how often a few things happen.
#ifdef DRAY_F64_CLOCKS
__device__ unsigned long long dray_f64_clocks[32];
#define F64_CLOCK(slot)                                              \
    do {                                                             \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
        clk[slot] += now__ - clk_t;                                  \
        clk_t = now__;                                               \
    } while (0)
#define F64_COUNT(slot, v) (clk[slot] += (unsigned long long)(v))
#else
#define F64_CLOCK(slot)
#define F64_COUNT(slot, v)
#endif
// DRAY_MARKS (tools/isa_budget.py --kernel f64): named comments in the instruction stream, so that the instructions of
// each section of the main loop can be counted.
#ifdef DRAY_MARKS
#define F64_MARK(name) asm volatile("; MARK " name)
#else
#define F64_MARK(name)
#endif

// One finished sample in a wave's ring: the colour that joins the pixel's sum (camera.rs:223-225: 0 + sample).
typedef double D2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(32))) RingEntry {
    D2 rg;   // one 16-byte store
    double b;
    double unused; // entries are 32 bytes apart
};
static_assert(sizeof(RingEntry) == 32, "ring entries are 32 bytes apart");
constexpr uint32_t kFifoSlots = 16; // chunks a wave may have grabbed and not begun to retire

#ifndef DRAY_F64_WAVES_PER_EU
#define DRAY_F64_WAVES_PER_EU 4
#endif

typedef float V4f __attribute__((ext_vector_type(4))); // a float4 that can be read through any address space

// Everything a launch of the stream kernel is given.  The kernel reads it where it is used, straight from the
// kernel-argument segment (scalar loads), instead of holding all of it in scalar registers for the whole launch.
struct F64Args {
    CamDev<double> cam;
    const double* geom;
    const double* mat;
    const uint32_t* kind;
    int32_t n;
    RefGrid rg;
    Partition part;
    RngKey key;
    uint32_t* color;
    double* raw;
    int32_t* count;
    DeviceCounters* counters;
    RingEntry* rings;
    uint32_t chunk_pixels;
    int32_t n_cells, n_entries; // the grid's tables (the kernel that stages a copy of the grid in LDS)
    // the f32 search (Filter32): the narrow spheres' rows cx, cy, cz, r*r and their list positions, the ball that holds
    // them (centre, radius; radius +inf: no narrow sphere), and the limits inside which the search may be trusted
    const V4f* narrow_geom;
    const int32_t* narrow_index;
    int32_t n_narrow_padded;
    int32_t filter_ok;   // the scene's size is one f32 arithmetic has room for
    double ball[4];
    float narrow_cmax;   // no finite narrow sphere's centre lies farther from the origin of coordinates than this
    double ball_far2;    // an origin farther than this (squared) from the ball's centre after the move: the ray passes the ball by
    double reach;        // largest origin coordinate the move is made for (beyond: the scan of the whole list)
};

// The copy of the grid a workgroup of the LDS kernel stages at its entry, so compact that four workgroups' copies fit
// a compute unit: the narrow spheres' f32 rows (cx, cy, cz, r*r) by list position, then a word per cell -- first entry
// (12 bits) | entry count (4 bits) | list position of the first entry's sphere (16 bits) --, then the entries as 16-bit
// list positions.  A lane entering a cell has its first sphere from the cell's word, a lane staying in its cell reads
// the entry table: two dependent reads per step, both from LDS.
constexpr uint32_t kLds64BeginBits = 12u, kLds64CountBits = 4u;

// Hides where a pointer came from, so that reads through it are not moved across this point.
template <typename T>
__device__ __forceinline__ const DRAY_CONSTANT T* opaque(const DRAY_CONSTANT T* p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (const DRAY_CONSTANT T*)v;
}

// A wave-uniform value kept in a scalar register from here on: what produced it cannot be moved below this point.
template <typename T> __device__ __forceinline__ void held(T& v) { asm volatile("" : "+s"(v)); }

// sphere.rs:28-48 for one sphere against the closest hit so far (smallest root, on equal roots the earlier entry),
// with the two possible divisions by |d|^2 made from one refined reciprocal where the operands allow (div_by).
__device__ __forceinline__ void exact_test(V3<double> center, double radius, int32_t k, V3<double> ro, V3<double> rd,
                                           double& closest, int32_t& best) {
    const double tmin = 0.001;
    V3<double> oc = vsub(center, ro);
    double a = vlen2(rd);
    double h = vdot(rd, oc);
    double c = vlen2(oc) - radius * radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.) return;
    double sqrtd = vsqrt(discriminant);
    const bool a_ok = div_band(a);
    const double y = refined_rcp(a);
    double num = h - sqrtd;
    double root = (a_ok && div_band(num)) ? div_by(num, a, y) : num / a;
    if (!(tmin < root)) { // interval.rs:28-30, strict
        num = h + sqrtd;
        root = (a_ok && div_band(num)) ? div_by(num, a, y) : num / a;
        if (!(tmin < root)) return;
    }
    if (root < INFINITY && (root < closest || (root == closest && k < best))) {
        closest = root;
        best = k;
    }
}

template <bool ITERATIVE, bool LDSGRID>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_F64_WAVES_PER_EU, DRAY_F64_WAVES_PER_EU)))
void render_f64_stream_kernel(F64Args) {
    const DRAY_CONSTANT F64Args* ka = (const DRAY_CONSTANT F64Args*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ V4f lds_rows[]; // LDSGRID: f32 rows by list position, then the cell words, then the entries
    if constexpr (LDSGRID) {
        const int32_t n_list = ka->n, n_cells = ka->n_cells, n_entries = ka->n_entries;
        const DRAY_CONSTANT V4f* narrow_geom = as_constant(ka->narrow_geom);
        const DRAY_CONSTANT int32_t* narrow_index = as_constant(ka->narrow_index);
        const uint2* cells = ka->rg.grid.cells;
        const int32_t* entry_index = ka->rg.entry_index;
        uint32_t* lds_cells = (uint32_t*)(lds_rows + n_list);
        uint16_t* lds_entries = (uint16_t*)(lds_cells + n_cells);
        // (rows of list positions that hold no narrow sphere -- wide spheres -- are never read: no entry names them)
        for (int32_t k = threadIdx.x; k < ka->n_narrow_padded; k += blockDim.x) {
            const int32_t at = narrow_index[k];
            if (at >= 0 && at < n_list) lds_rows[at] = narrow_geom[k];
        }
        for (int32_t k = threadIdx.x; k < n_cells; k += blockDim.x) {
            const uint2 rec = cells[k];
            const uint32_t first = rec.y != 0u ? (uint32_t)entry_index[rec.x] : 0u;
            lds_cells[k] = rec.x | (rec.y << kLds64BeginBits) | (first << (kLds64BeginBits + kLds64CountBits));
        }
        for (int32_t k = threadIdx.x; k < n_entries; k += blockDim.x) lds_entries[k] = (uint16_t)entry_index[k];
        __syncthreads(); // the only block-wide wait: once, before any work is drawn
    }
    // the cell's record: first entry, entry count, and (LDSGRID) the first entry's sphere
    auto cell_of = [&](uint32_t ci, uint32_t& first_sphere) -> uint2 {
        if constexpr (LDSGRID) {
            const uint32_t rec = ((const uint32_t*)(lds_rows + ka->n))[ci];
            first_sphere = rec >> (kLds64BeginBits + kLds64CountBits);
            return make_uint2(rec & ((1u << kLds64BeginBits) - 1u), (rec >> kLds64BeginBits) & ((1u << kLds64CountBits) - 1u));
        } else {
            first_sphere = 0u;
            return ka->rg.grid.cells[ci];
        }
    };
    // per wave: the colours of the samples being retired, 64 x 3 doubles -- and, in another section of the turn, the
    // streams coop_draw64 hands to server lanes, 64 x 16 bytes, in the same place (LDS keeps one wave's accesses in order)
    __shared__ double lds_stage_all[4][64][3];
    __shared__ uint32_t lds_fifo[4][kFifoSlots]; // first pixel of the chunks the wave has grabbed, oldest first
    __shared__ V4f lds_ray[4][4][64];            // RayConst of every lane: [wave][A slack | B on | o tm_lo | u tm_hi][lane]
    __shared__ double lds_acc[4][4];             // running sum of the pixel the wave is retiring: red, green, blue
    static_assert(sizeof(lds_stage_all[0]) >= 64 * sizeof(uint4), "the draw requests share the retirement's staging");
    // (the wave's number in its workgroup is the same in every lane; said so that what is read through it -- the
    // chunk queue -- and every count of the wave's that depends on that stays in scalar registers)
    const int lane = (int)(threadIdx.x & 63u), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double (*lds_stage)[3] = lds_stage_all[wv];
    uint4* lds_req = (uint4*)&lds_stage_all[wv][0][0];
    const uint32_t W = (uint32_t)ka->cam.W, spp = (uint32_t)ka->cam.spp, npix = (uint32_t)ka->part.n_rows * W;
    constexpr bool iterative = ITERATIVE; // how a sample's attenuations are multiplied (camera.rs:130 or :274)
    RingEntry* ring = ka->rings + (size_t)(blockIdx.x * 4u + (uint32_t)wv) * kF64RingEntries;
    unsigned long long my_segments = 0;

    // where partition pixel `pp` lives: its place in the caller's buffers and its Philox pixel word j * W + i
    auto place_of = [&](uint32_t pp, size_t& out, uint32_t& word) {
        const uint32_t r = pp / W, i = pp - r * W;
        const uint32_t j = (uint32_t)ka->part.row_begin + r * (uint32_t)ka->part.row_step;
        out = (size_t)(ka->part.compact ? r : j) * (size_t)W + (size_t)i;
        word = j * W + i;
    };

    // per-lane state: one sample's path
    uint32_t stack[ITERATIVE ? 1 : kExactMaxDepth]; // spheres whose albedo multiplies the sample, outermost first
    V3<double> ro = {0., 0., 0.}, rd = {0., 0., 1.};
    V3<double> att = {1., 1., 1.}; // camera.rs:26 (iterative form)
    uint32_t pw = 0u, smp = 0u, item = 0u; // the sample: Philox pixel word, sample number, number in the wave's stream
    int32_t depth = 0, sp = 0;
    bool has = false;      // the lane carries a sample
    bool camera = false;   // ... whose camera ray is still to be made
    bool walking = false;  // on its way through the grid
    bool paused = false;   // ... and stopped with both candidate slots taken: they are evaluated, then the walk goes on
    bool fallback = false; // the closest hit of this ray comes from the scan of the whole list
    Walk32 wk = {};
    Cand cd = {INFINITY, NAN, kNoCand, kNoCand};
    double closest = INFINITY; // smallest root the reference's test has accepted so far, and its sphere's list position
    int32_t best = 0x7fffffff;

    // wave state (the same in every lane)
    uint32_t issued = 0u, retired = 0u;       // samples of the wave's stream handed out / added to their pixels
    uint32_t grabbed = 0u, begun = 0u;        // chunks taken from the queue / chunks whose retirement has begun
    uint32_t iss_p0 = 0u, iss_q = 0u, iss_left = 0u;  // the chunk being handed out: first pixel, next sample, samples left
    uint32_t ret_p0 = 0u, ret_dp = 0u, ret_np = 0u, ret_s = 0u; // the chunk being retired: first pixel, pixel, pixels; sample
    bool exhausted = false;
    uint32_t idle_turns = 0u; // turns in a row in which no lane carried a sample (see the loop's exit)

#ifdef DRAY_F64_CLOCKS
    unsigned long long clk[32] = {};
    unsigned long long clk_t = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        ka = opaque(ka);
        F64_MARK("A_walk");
        F64_COUNT(16, 1);                              // turns
        F64_COUNT(17, __popcll(__ballot(has)));        // lanes carrying a sample
        // ---- walk steps, while enough lanes are on their way --------------------------------------------------
        {
            const int quorum = exhausted ? 1 : DRAY_F64_WALK_QUORUM;
            const float gn0 = (float)ka->rg.grid.n[0], gn1 = (float)ka->rg.grid.n[1], gn2 = (float)ka->rg.grid.n[2];
            const V4f q0 = lds_ray[wv][0][lane], q1 = lds_ray[wv][1][lane], q2 = lds_ray[wv][2][lane], q3 = lds_ray[wv][3][lane];
            const float wA[3] = {q0.x, q0.y, q0.z}, wB[3] = {q1.x, q1.y, q1.z}, wslack = q0.w;
            Filter32 flt = {q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, q2.w, q3.w, 0.f, 0.f};
            filter_consts(flt, q1.w, ka->narrow_cmax);
            // where the cell words and the entries lie in the LDS copy: worked out once a turn and kept in scalar
            // registers (left to the compiler they are read from the argument block, and waited for, at every step)
            uint32_t cells_off = (uint32_t)ka->n * (uint32_t)sizeof(V4f);
            uint32_t entries_off = cells_off + (uint32_t)ka->n_cells * 4u;
            held(cells_off);
            held(entries_off);
            const uint32_t* wcells = (const uint32_t*)((const char*)lds_rows + cells_off);
            const uint16_t* wentries = (const uint16_t*)((const char*)lds_rows + entries_off);
            do { // at least one step a turn: fewer lanes than a quorum may be all the wave has
                F64_COUNT(18, 1);                              // walk steps
                F64_COUNT(19, __popcll(__ballot(walking)));    // lanes walking at them
                if (walking) {
                    uint32_t sphere = 0u; // LDSGRID: list position of the sphere this step tests
                    bool entered = false;
                    if (wk.e == wk.e_end) { // this cell is done: stop, or step into the next one
                        const float tx = fmaf(wk.c[0], wA[0], wB[0]), ty = fmaf(wk.c[1], wA[1], wB[1]), tz = fmaf(wk.c[2], wA[2], wB[2]);
                        const float t_exit = fminf(tx, fminf(ty, tz));
                        const bool stop = cd.ub < fmaf(t_exit, 1.f - 3.814697265625e-6f, -wslack); // nothing beyond this cell can be closer
                        const bool sx = tx == t_exit;
                        const bool sy = !sx & (ty == t_exit);
                        // one step along the axis left first; A carries the direction's sign (a parallel axis is never chosen)
                        wk.c[0] += sx ? copysignf(1.f, wA[0]) : 0.f;
                        wk.c[1] += sy ? copysignf(1.f, wA[1]) : 0.f;
                        wk.c[2] += (sx | sy) ? 0.f : copysignf(1.f, wA[2]);
                        // still inside the grid: a coordinate is a whole number from -1 to n, and as unsigned integers the
                        // bits of -1 lie above those of every count (three compares instead of four selects and two)
                        const bool in0 = __float_as_uint(wk.c[0]) < __float_as_uint(gn0), in1 = __float_as_uint(wk.c[1]) < __float_as_uint(gn1);
                        const bool in2 = __float_as_uint(wk.c[2]) < __float_as_uint(gn2);
                        const bool inside = in0 & in1 & in2;
                        const bool go = inside & !stop;
                        walking = go;
                        const uint32_t ci = (uint32_t)fmaf(fmaf(wk.c[2], gn1, wk.c[1]), gn0, wk.c[0]);
                        uint2 cellrec;
                        if constexpr (LDSGRID) {
                            const uint32_t rec = wcells[go ? ci : 0u];
                            sphere = rec >> (kLds64BeginBits + kLds64CountBits);
                            cellrec = make_uint2(rec & ((1u << kLds64BeginBits) - 1u), (rec >> kLds64BeginBits) & ((1u << kLds64CountBits) - 1u));
                        } else {
                            cellrec = cell_of(go ? ci : 0u, sphere);
                        }
                        wk.e = cellrec.x;
                        wk.e_end = go ? cellrec.x + cellrec.y : cellrec.x;
                        entered = true;
                    }
                    if (wk.e != wk.e_end) {
                        V4f row;
                        if constexpr (LDSGRID) {
                            if (!entered) // (beside the other lanes' cell read: the two do not depend on one another)
                                sphere = wentries[wk.e];
                            row = lds_rows[sphere];
                        } else {
                            row = ((const V4f*)ka->rg.grid.entry_geom)[wk.e];
                            sphere = (uint32_t)ka->rg.grid.entry_index[wk.e];
                        }
                        if (filter_test(flt, row.x, row.y, row.z, row.w, sphere, cd)) { // both slots taken: evaluate them, then go on
                            paused = true;
                            walking = false;
                        } else {
                            ++wk.e;
                        }
                    }
                }
            } while (__popcll(__ballot(walking)) >= quorum);
        }

        F64_CLOCK(0);
        F64_MARK("B_retire");
        // ---- retire: the ring's finished samples, in the order of the stream, join their pixels (colors.rs:38-43) ----
        // Every sample of the stream below `issued` is either in a lane or finished, so the samples before the oldest
        // one in a lane are all in the ring.  The wave retires them 64 at a time (a look at the ring is a trip to
        // memory), and whatever is finished once the queue has run out.
        for (;;) {
            const uint32_t waiting = issued - retired;
            int n_ok = 0;
            if (waiting >= 64u && __ballot(has && item - retired < 64u) == 0ull) {
                n_ok = 64;
            } else if (exhausted && waiting != 0u) {
                uint32_t oldest = has ? item - retired : waiting; // of the samples still in lanes
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) oldest = min(oldest, (uint32_t)__shfl_xor((int)oldest, off, 64));
                // (every lane holds the minimum now; said so that what depends on it stays wave-uniform for the compiler too:
                // else the loop's exit, and with it every count the wave keeps, is taken to differ from lane to lane)
                n_ok = (int)min((uint32_t)__builtin_amdgcn_readfirstlane((int)oldest), 64u);
            }
            if (n_ok == 0) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the lanes' stores to the ring have completed
            const RingEntry* e = ring + ((retired + (uint32_t)lane) & (kF64RingEntries - 1u));
            // (lanes past n_ok read entries nobody uses.)  The colours go through LDS to lanes 0, 1, 2, which add up
            // one channel each, sample after sample: one load and one addition per sample
            const D2 rg = e->rg;
            lds_stage[lane][0] = rg.x;
            lds_stage[lane][1] = rg.y;
            lds_stage[lane][2] = e->b;
            const int ch = lane < 3 ? lane : 2;
            double acc = lds_acc[wv][ch]; // lane 0 red, 1 green, 2 blue
            int k = 0;
            while (k < n_ok) {
                if (ret_s == 0u) { // a pixel begins
                    if (ret_dp == ret_np) { // ... and with it a chunk
                        ret_p0 = lds_fifo[wv][begun & (kFifoSlots - 1u)];
                        ret_np = min(ka->chunk_pixels, npix - ret_p0);
                        ret_dp = 0u;
                        begun++;
                    }
                    uint32_t word;
                    size_t ret_out; // (worked out again when the pixel is complete: two scalar registers the loop does not hold)
                    place_of(ret_p0 + ret_dp, ret_out, word);
                    acc = ka->raw ? ka->raw[3 * ret_out + (size_t)ch] : 0.;
                }
                const int take = min(n_ok - k, (int)(spp - ret_s));
#pragma unroll 8
                for (int t = 0; t < take; ++t) acc += lds_stage[k + t][ch]; // colors.rs:38
                k += take;
                ret_s += (uint32_t)take;
                if (ret_s == spp) { // the pixel is complete
                    const double a0 = readlane_f64(acc, 0), a1 = readlane_f64(acc, 1), a2 = readlane_f64(acc, 2);
                    uint32_t word;
                    size_t ret_out;
                    place_of(ret_p0 + ret_dp, ret_out, word);
                    if (lane == 0) {
                        const int32_t cnt = (ka->count ? ka->count[ret_out] : 0) + (int32_t)spp; // colors.rs:39
                        const double denom = cnt != 0 ? (double)cnt : 1.;                 // colors.rs:40
                        ka->color[ret_out] = pack_argb(a0 / denom, a1 / denom, a2 / denom);
                        if (ka->raw) { ka->raw[3 * ret_out + 0] = a0; ka->raw[3 * ret_out + 1] = a1; ka->raw[3 * ret_out + 2] = a2; }
                        if (ka->count) ka->count[ret_out] = cnt;
                    }
                    ret_s = 0u;
                    ret_dp++;
                }
            }
            if (lane < 3) lds_acc[wv][lane] = acc;
            retired += (uint32_t)n_ok;
        }

        F64_CLOCK(3);
        F64_MARK("C_settle_hit");
        // ---- lanes whose search is over, or has stopped with its slots full: the candidates take the reference's test ----
        {
            const bool settle = has && !walking && !camera && !fallback && depth > 0;
            if (__ballot(settle && (cd.i0 != kNoCand || cd.i1 != kNoCand)) != 0ull) {
                F64_COUNT(21, 1);                                                  // turns that settle candidates
                F64_COUNT(22, __popcll(__ballot(settle && cd.i0 != kNoCand)));     // first slots evaluated
                F64_COUNT(23, __ballot(settle && cd.i1 != kNoCand) != 0ull);       // turns with a second slot in some lane
                F64_COUNT(24, __popcll(__ballot(settle && cd.i1 != kNoCand)));     // second slots evaluated
                const uint32_t k0 = settle ? cd.i0 : kNoCand, k1 = settle ? cd.i1 : kNoCand;
                // both rows are asked for before either is used: one trip to memory
                const double* g0 = ka->geom + 4 * (size_t)(k0 != kNoCand ? k0 : 0u);
                const double* g1 = ka->geom + 4 * (size_t)(k1 != kNoCand ? k1 : 0u);
                const D2 g0a = *(const D2*)g0, g0b = *(const D2*)(g0 + 2), g1a = *(const D2*)g1, g1b = *(const D2*)(g1 + 2);
                if (k0 != kNoCand) exact_test(V3<double>{g0a.x, g0a.y, g0b.x}, g0b.y, (int32_t)k0, ro, rd, closest, best);
                if (k1 != kNoCand) exact_test(V3<double>{g1a.x, g1a.y, g1b.x}, g1b.y, (int32_t)k1, ro, rd, closest, best);
                if (settle) { cd.i0 = cd.i1 = kNoCand; cd.lo0 = NAN; }
            }
            if (paused) { // the sphere the walk stopped at is tested again at the next step, with the slots free
                walking = true;
                paused = false;
            }
        }
        // ---- lanes whose walk is over: the hit (hittable_list.rs:60-74), or the sky, or the end of the path ----------
        // Which sphere or box and how far is all that is settled here.  Point and normal, the scatter, the sky's colour and
        // the sample's product are made after the wave's draws: none of them sits in registers through those, and the
        // square roots and divisions of the sky and of the materials are then one piece of code (see below).  A lane
        // whose sample ends is free for the next one at once; `end_item` keeps the ended sample's place in the ring.
        const bool ready = has && !walking && !camera;
        bool scatters = false, ended = false, sky = false, is_box = false;
        uint32_t mk = 0xFFu;
        const uint32_t end_item = item;
        if (ready) {
            if (depth <= 0) { // camera.rs:265-267 / :120-123, or absorbed (camera.rs:276; the scatter left no depth): black
                ended = true;
            } else {
                if (fallback) { // the reference's own scan, hittable_list.rs:65-70
                    closest = INFINITY;
                    best = 0x7fffffff;
                    for (int32_t k = 0; k < ka->n; ++k) {
                        double root;
                        if (sphere_root<double>(ka->geom, k, ro, rd, root) && root < closest) {
                            closest = root;
                            best = k;
                        }
                    }
                }
                is_box = boxes_closest<double>(BoxList{ka->rg.box64, ka->rg.box_index, ka->rg.n_boxes}, ro, rd, closest, best);
                if (best != 0x7fffffff) {
                    mk = ka->kind[best];
                    scatters = true;
                } else {
                    ended = sky = true; // camera.rs:279-281
                }
            }
            if (ended) has = false;
        }

        F64_CLOCK(1);
        F64_COUNT(20, __popcll(__ballot(ready)));      // lanes shading
        F64_MARK("D_refill");
        // ---- refill: free lanes take the next samples of the stream (camera.rs:217-224) ------------------------
        {
            bool need = !has;
            unsigned long long want = __ballot(need);
            // hands the next samples of the open chunk to the lanes that want one; returns the lanes still wanting
            // (`none`: there was no sample or no room in the ring to hand out)
            auto hand_out = [&](unsigned long long want, bool& none) -> unsigned long long {
                const uint32_t room = kF64RingEntries - (issued - retired); // a ring's length ahead of the oldest at most
                const uint32_t avail = min(iss_left, room);
                const uint32_t wanted = (uint32_t)__popcll(want);
                const uint32_t take = wanted < avail ? wanted : avail;
                none = take == 0u;
                if (none) return want;
                const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                if (need && rank < take) {
                    const uint32_t q = iss_q + rank;
                    const uint32_t dp = q / spp;
                    size_t out;
                    place_of(iss_p0 + dp, out, pw);
                    smp = q - dp * spp;
                    item = issued + rank;
                    has = true;
                    camera = true;
                    need = false;
                }
                iss_q += take;
                iss_left -= take;
                issued += take;
                return __ballot(need);
            };
            // the usual turn -- the open chunk has samples left -- is straight-line code: the loop below, which may grab
            // chunks, carries every lane's new sample through copies
            bool none = false;
            if (want != 0ull && iss_left != 0u) want = hand_out(want, none);
            while (want != 0ull && !none) {
                if (iss_left == 0u && !exhausted) { // the next chunk of pixels
                    if (grabbed - begun == kFifoSlots) break;
                    unsigned long long base = 0ull;
                    if (lane == 0) base = atomicAdd(&ka->counters->next_item, (unsigned long long)ka->chunk_pixels);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
                    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                    if (hi != 0u || lo >= npix) {
                        exhausted = true;
                    } else {
                        if (lane == 0) lds_fifo[wv][grabbed & (kFifoSlots - 1u)] = lo;
                        grabbed++;
                        iss_p0 = lo;
                        iss_q = 0u;
                        iss_left = min(ka->chunk_pixels, npix - lo) * spp;
                    }
                }
                if (exhausted) break;
                want = hand_out(want, none);
            }
        }
        if (__ballot(has) == 0ull) {
            if (retired == issued) break; // nothing in flight, nothing left to take
            // No lane carries a sample, yet some are issued and not retired: they are all in the ring, and the retirement
            // above takes them 64 at a time or, with the queue exhausted, all at once -- a few turns at most.  A wave that
            // finds itself here turn after turn has lost count: it leaves (the call's sample total then falls short,
            // which its statistics show) instead of holding the GPU for ever.
            if (++idle_turns > 4096u) break;
        } else {
            idle_turns = 0u;
        }

        F64_CLOCK(4);
        F64_MARK("E_philox_block0");
        // ---- the first uniforms of the next step, and the rejection samplers, by the whole wave -----------------
        const bool draws = scatters || camera;
        const Stream st = {ka->key.k0, ka->key.k1, pw, ka->key.sample_offset + smp, camera ? 0u : (uint32_t)(ka->cam.max_depth - depth + 1)};
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (draws) st.block(0, w);
        const uint32_t draw = (scatters && (mk == DRAY_MAT_LAMBERTIAN || mk == DRAY_MAT_METAL)) ? (uint32_t)kDraw64UnitVector
                              : (camera && ka->cam.defocus != 0)                                      ? (uint32_t)kDraw64UnitDisk
                                                                                                  : (uint32_t)kDraw64None;
        F64_CLOCK(5);
        F64_MARK("F_coop_draw");
        const Drawn got = coop_draw64(st, w, draw, lds_req, lane);
        F64_CLOCK(6);
        F64_MARK("G_scatter_sky_end");

        // ---- scatter (material.rs) and sky (camera.rs:279-281) ---------------------------------------------
        // Every operation is the reference's, in the reference's order, lane by lane; what is shared is the code.  A
        // wave nearly always holds lanes of every kind at once, so each IEEE square root and each division by a common
        // divisor that the kinds need is written once, for whichever lanes need it at that point:
        //   first normalisation  v / sqrt(|v|^2): the sky's and the Dialectric's unit direction (vec3.rs:120-122), the
        //                        Lambertian's random unit vector (vec3.rs:82), the Metal's unit reflection (material.rs:75);
        //   second square root:  the Metal's random unit vector (then divided through), the Dialectric's sin_theta
        //                        (material.rs:109);
        //   third square root:   refract's parallel part (vec3.rs:107).
        bool fresh = false;
        {
            const bool lam = scatters && mk == DRAY_MAT_LAMBERTIAN, met = scatters && mk == DRAY_MAT_METAL;
            const bool die = scatters && mk == DRAY_MAT_DIELECTRIC;
            Hit<double> rec = {}; // sphere.rs:49-55
            if (scatters) {
                if (is_box) box_record<double>(ro, rd, closest, best, rec);
                else fill_hit<double>(ka->geom, best, ro, rd, closest, rec);
            }
            const D2* mrow = (const D2*)(ka->mat + 4 * (size_t)(scatters ? rec.index : 0));
            D2 m01 = {0., 0.}, m23 = {0., 0.};
            if (met || die) { m01 = mrow[0]; m23 = mrow[1]; }
            V3<double> pr = {0., 0., 0.}; // the accepted attempt's point in the cube (vec3.rs:78)
            double lensq = 1.;
            if (lam || met) {
                const uint32_t ww[4] = {got.a, got.b, got.c, 0u};
                unit_vector_attempt<double>(ww, pr, lensq);
            }
            // first normalisation
            V3<double> va = rd;
            if (lam) va = pr;
            if (met) va = reflect(rd, rec.normal); // material.rs:74
            V3<double> ua = {0., 0., 0.};
            if (sky || lam || met || die) ua = vdiv(va, vsqrt(vlen2(va)));
            // what each kind makes of it, up to its next square root
            V3<double> sd = rd, so = ro; // Transparent (material.rs:25-31): the incoming ray, unchanged
            bool scattered = scatters && (mk == DRAY_MAT_TRANSPARENT);
            double cos_theta = 0., ri = 1., x2nd = 1.;
            if (scatters && mk != DRAY_MAT_TRANSPARENT) so = rec.p;
            if (lam) { // material.rs:46-56
                V3<double> dir = vadd(rec.normal, ua);
                const double tiny = 1e-8; // vec3.rs:42-45
                if (fabs(dir.x) < tiny && fabs(dir.y) < tiny && fabs(dir.z) < tiny) dir = rec.normal;
                sd = dir;
                scattered = true;
            }
            if (met) x2nd = lensq;
            if (die) { // material.rs:103-109; the table holds 1 / ior beside ior (an IEEE quotient is the same wherever it is made)
                ri = rec.front_face ? m01.x : m23.y;
                cos_theta = fmin(vdot(vneg(ua), rec.normal), 1.0);
                x2nd = 1.0 - cos_theta * cos_theta;
            }
            // second square root
            double s2nd = 0.;
            if (met || die) s2nd = vsqrt(x2nd);
            if (met) { // material.rs:75-80
                const V3<double> uv = vdiv(pr, s2nd);
                sd = vadd(ua, vscale(uv, m23.y));
                scattered = vdot(sd, rec.normal) > 0.;
            }
            bool refracts = false;
            V3<double> perp = {0., 0., 0.};
            if (die) { // material.rs:110-121
                bool reflects = ri * s2nd > 1.0;
                if (!reflects) { // the uniform is drawn only when refraction is possible
                    // material.rs:93-98; r0 = ((1 - ri) / (1 + ri))^2 comes from the table, for the face that was hit
                    const double r0 = rec.front_face ? m01.y : m23.x;
                    const double x = 1.0 - cos_theta;
                    const double xx = x * x;
                    const double x5 = x * (xx * xx); // powi(5)
                    reflects = r0 + (1.0 - r0) * x5 > u01<double>(w[0]);
                }
                refracts = !reflects;
                if (reflects) sd = reflect(ua, rec.normal);
                else perp = vscale(vadd(ua, vscale(rec.normal, cos_theta)), ri); // vec3.rs:105-106 (the same cosine)
                scattered = true;
            }
            // third square root
            if (refracts) {
                const double k = -(vsqrt(fabs(1.0 - vlen2(perp)))); // vec3.rs:107
                sd = vadd(perp, vscale(rec.normal, k));
            }
            if (scatters) {
                if (scattered) {
                    const bool white = mk == DRAY_MAT_DIELECTRIC || mk == DRAY_MAT_TRANSPARENT; // attenuation (1, 1, 1)
                    if constexpr (iterative) {
                        V3<double> al = white ? V3<double>{1., 1., 1.}
                                              : V3<double>{ka->mat[4 * rec.index], ka->mat[4 * rec.index + 1], ka->mat[4 * rec.index + 2]};
                        att = vmul(att, al); // camera.rs:130
                    } else if (!white) {
                        stack[sp++] = (uint32_t)rec.index; // a product with (1,1,1) is exact, so those are skipped
                    }
                    ro = so;
                    rd = sd;
                    depth -= 1;
                    fresh = depth > 0; // with no depth left the sample ends black at the lane's next turn
                } else {
                    depth = 0; // camera.rs:276: absorbed, the sample ends black at the lane's next turn
                }
            }
            // ---- samples that end here: their colour goes to its place in the wave's ring ---------------------------
            F64_CLOCK(7);
            if (ended) {
                V3<double> c = {0., 0., 0.};
                if (sky) {
                    const double a = 0.5 * (ua.y + 1.0);
                    c = vadd(vscale(V3<double>{1.0, 1.0, 1.0}, 1.0 - a), vscale(V3<double>{0.5, 0.7, 1.0}, a));
                }
                if constexpr (iterative) {
                    c = vmul(att, c); // camera.rs:121,135,142
                } else {
                    // camera.rs:274, innermost product first; four albedos are fetched at a time
                    auto four = [&]() {
                        const int32_t n4 = sp < 4 ? sp : 4;
                        uint32_t q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) q[u] = stack[sp - 1 - (u < n4 ? u : 0)];
                        V3<double> al[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) al[u] = {ka->mat[4 * q[u]], ka->mat[4 * q[u] + 1], ka->mat[4 * q[u] + 2]};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < n4) c = vmul(al[u], c);
                        sp -= n4;
                    };
                    if (sp > 0) { // (the first four, all that most samples have, without a loop's copies of the colour)
                        four();
                        while (sp > 0) four();
                    }
                }
                RingEntry* e = ring + (end_item & (kF64RingEntries - 1u));
                e->rg = D2{0. + c.x, 0. + c.y}; // camera.rs:223-225: pixel_color = (0, 0, 0) + sample
                e->b = 0. + c.z;
            }
        }

        F64_CLOCK(2);
        F64_MARK("H_camera_ray");
        // ---- camera ray of a new sample: camera.rs:238-262 --------------------------------------------------
        if (camera) {
            double px = 0., py = 0.;
            unit_disk_attempt<double>(got.a, got.b, px, py);
            const uint32_t j = pw / W, i = pw - j * W;
            get_ray_with<double>(ka->cam, (int32_t)i, (int32_t)j, w, px, py, ro, rd);
            att = {1., 1., 1.};
            depth = ka->cam.max_depth;
            sp = 0;
            camera = false;
            fresh = depth > 0;
        }

        F64_CLOCK(8);
        F64_MARK("I_new_ray");
        // ---- a new ray: the wide spheres in f64, the f32 view of the ray, the loose spheres, the first cell ------
        if (fresh) {
            my_segments++;
            closest = INFINITY;
            best = 0x7fffffff;
            cd = {INFINITY, NAN, kNoCand, kNoCand};
            wk.e = wk.e_end = 0u;
            walking = false;
            paused = false;
            const double a = vlen2(rd);
            const double omax = fmax(fabs(ro.x), fmax(fabs(ro.y), fabs(ro.z)));
            // Not for the f32 search: rays that are not finite (NaN fails every comparison), directions whose squared
            // length leaves the middle of the f64 range, origins so far from the scene that the move to the ball would
            // lose the ray in f64 rounding, scenes of a size f32 has no room for.  Those take the scan of the whole list.
            fallback = !(omax <= ka->reach && fabs(rd.x) < 1e300 && fabs(rd.y) < 1e300 && fabs(rd.z) < 1e300 && a >= 1e-200 &&
                         a <= 1e200 && ka->filter_ok != 0);
            if (!fallback) {
                // unit direction (v_rsq_f64 is good to 2^-23; one Newton step makes it 2^-40), and the origin moved along
                // the ray up to the ball of the narrow spheres
                double il = __builtin_amdgcn_rsq(a);
                il = il * __builtin_fma(-0.5 * a, il * il, 1.5);
                const double len = a * il;
                const V3<double> u = {rd.x * il, rd.y * il, rd.z * il};
                const V3<double> cb = {ka->ball[0], ka->ball[1], ka->ball[2]};
                const double hb = __builtin_fma(cb.z - ro.z, u.z, __builtin_fma(cb.y - ro.y, u.y, (cb.x - ro.x) * u.x));
                const double s0 = fmax(hb - ka->ball[3], 0.);
                const V3<double> om = {__builtin_fma(s0, u.x, ro.x), __builtin_fma(s0, u.y, ro.y), __builtin_fma(s0, u.z, ro.z)};
                const V3<double> mb = vsub(cb, om);
                const bool passes_by = __builtin_fma(mb.z, mb.z, __builtin_fma(mb.y, mb.y, mb.x * mb.x)) > ka->ball_far2;
                RayConst rc;
                Filter32& flt = rc.f;
                flt.ox = (float)om.x; flt.oy = (float)om.y; flt.oz = (float)om.z;
                flt.ux = (float)u.x; flt.uy = (float)u.y; flt.uz = (float)u.z;
                const float tm = (float)__builtin_fma(0.001, len, -s0);
                flt.tm_lo = fmaf(-4.76837158203125e-07f, fabsf(tm), tm);
                flt.tm_hi = fmaf(4.76837158203125e-07f, fabsf(tm), tm);
                const float on = fmaf(0.001953125f, (float)s0, __builtin_amdgcn_sqrtf(fmaf(flt.oz, flt.oz, fmaf(flt.oy, flt.oy, flt.ox * flt.ox))) * 1.000001f);
                filter_consts(flt, on, ka->narrow_cmax);
                { // the spheres every ray meets, rows and list positions through scalar loads: the wide ones (their
                  // |oc|^2 - r^2 cancels in f32: bracketed from f64 values), then the loose ones (narrow spheres kept out of the grid)
                    bool over = false;
                    const DRAY_CONSTANT double* ag = as_constant(ka->rg.always_geom64);
                    const DRAY_CONSTANT int32_t* ai = as_constant(ka->rg.always_index);
                    const int32_t n_wide = ka->rg.n_wide;
                    for (int32_t q = 0; q < n_wide; ++q)
                        over = over | wide_test(flt.tm_lo, flt.tm_hi, on, ro, u, s0, ag[4 * q], ag[4 * q + 1], ag[4 * q + 2], ag[4 * q + 3],
                                                (uint32_t)ai[q], cd);
                    const DRAY_CONSTANT V4f* lg = as_constant((const V4f*)ka->rg.grid.loose_geom);
                    const DRAY_CONSTANT int32_t* li = as_constant(ka->rg.grid.loose_index);
                    const int32_t n_loose = ka->rg.grid.n_loose;
                    for (int32_t q = 0; q < n_loose; ++q) {
                        const V4f row = lg[q];
                        over = over | filter_test(flt, row.x, row.y, row.z, row.w, (uint32_t)li[q], cd);
                    }
                    if (over) fallback = true; // more than two of them at once may hold the closest root: the whole list decides
                }
                if (!fallback && !passes_by && ka->rg.grid.n[0] > 0) { // the ray's first cell (3-D DDA along the f32 ray)
                    const float o[3] = {flt.ox, flt.oy, flt.oz}, d[3] = {flt.ux, flt.uy, flt.uz};
                    float t_in = 0.f, t_out = INFINITY, bmax = 0.f;
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        // A direction parallel to the axis' planes walks as one that crosses them after 10^12 cell sizes:
                        // from inside the slab every plane is ~10^12 away and never the next one, from outside
                        // t_in > t_out.  (No compares, selects or branches in the set-up of an axis.)
                        const float da = copysignf(fmaxf(fabsf(d[ax]), 1e-12f), d[ax]);
                        const float inv = __builtin_amdgcn_rcpf(da);
                        rc.A[ax] = ka->rg.grid.cell[ax] * inv;
                        const float t0 = (ka->rg.grid.lo[ax] - o[ax]) * inv, t1 = (ka->rg.grid.hi[ax] - o[ax]) * inv;
                        rc.B[ax] = t0 + fmaxf(rc.A[ax], 0.f); // leaving cell k towards larger indices happens at plane k + 1
                        t_in = fmaxf(t_in, fminf(t0, t1));
                        t_out = fminf(t_out, fmaxf(t0, t1));
                        bmax = fmaxf(bmax, fmaxf(fabsf(t0), fabsf(rc.B[ax])));
                    }
                    rc.slack = bmax * 3.814697265625e-6f; // 2^-18 of the largest term the boundary times are formed from
                    if (t_in <= t_out + rc.slack && !(cd.ub < t_in - rc.slack)) {
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            const float at = (fmaf(d[ax], t_in, o[ax]) - ka->rg.grid.lo[ax]) * ka->rg.grid.inv_cell[ax];
                            wk.c[ax] = fminf(fmaxf(floorf(at), 0.f), (float)(ka->rg.grid.n[ax] - 1));
                        }
                        const uint32_t ci = (uint32_t)fmaf(fmaf(wk.c[2], (float)ka->rg.grid.n[1], wk.c[1]), (float)ka->rg.grid.n[0], wk.c[0]);
                        uint32_t unused;
                        const uint2 cellrec = cell_of(ci, unused);
                        wk.e = cellrec.x;
                        wk.e_end = cellrec.x + cellrec.y;
                        walking = true;
                        lds_ray[wv][0][lane] = V4f{rc.A[0], rc.A[1], rc.A[2], rc.slack};
                        lds_ray[wv][1][lane] = V4f{rc.B[0], rc.B[1], rc.B[2], on};
                        lds_ray[wv][2][lane] = V4f{flt.ox, flt.oy, flt.oz, flt.tm_lo};
                        lds_ray[wv][3][lane] = V4f{flt.ux, flt.uy, flt.uz, flt.tm_hi};
                    }
                }
            }
        }
        F64_CLOCK(9);
        F64_MARK("Z_loop_edge");
    }
#ifdef DRAY_F64_CLOCKS
    if (lane == 0)
        for (int q = 0; q < 32; ++q)
            if (clk[q]) atomicAdd(&dray_f64_clocks[q], clk[q]);
#endif

    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) my_segments += __shfl_down(my_segments, off, 64);
    if (lane == 0 && ka->counters) {
        atomicAdd(&ka->counters->segments, my_segments);
        atomicAdd(&ka->counters->samples, (unsigned long long)retired); // every sample handed out has been retired
    }
}

} // namespace

bool plan_render_f64_stream(const CamDev<double>& cam, const Partition& part, const SceneDev& scene, int device,
                            F64StreamPlan* plan) {
    const long long npix = (long long)part.n_rows * cam.W;
    if (npix <= 0 || npix > 0x7fffffffLL || cam.spp <= 0) return false;
    if (!ref_grid_of(scene, true).use) return false;
    // compute units of the device and resident blocks per unit: asked once per device
    struct DeviceFacts { int cus = 0, occ = 0; };
    static DeviceFacts facts[64];
    static std::mutex facts_mu;
    int cus = 256, occ = 2;
    {
        std::lock_guard<std::mutex> lock(facts_mu);
        DeviceFacts local;
        DeviceFacts& f = (device >= 0 && device < 64) ? facts[device] : local;
        if (f.cus == 0) {
            int v = 0;
            f.cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) ? v : 256;
            f.occ = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, render_f64_stream_kernel<false, false>, 256, 0) == hipSuccess && v > 0) ? v : 2;
        }
        cus = f.cus;
        occ = f.occ;
    }
    plan->block = 256;
    plan->grid = cus * occ;
    // the compact copy of the grid in LDS, where as many workgroups as without it fit a compute unit with theirs and the
    // packed fields hold the grid's numbers (DRAY_GRID_LDS=0: the tables are read from memory; tests, tuning runs)
    plan->lds_bytes = 0;
    {
        const size_t n_cells = (size_t)scene.grid.n[0] * (size_t)scene.grid.n[1] * (size_t)scene.grid.n[2];
        const size_t bytes = (size_t)scene.n * 16u + n_cells * 4u + ((scene.grid_entries * 2u + 15u) & ~(size_t)15u);
        const char* env = std::getenv("DRAY_GRID_LDS");
        int occ_lds = 0;
        if (!(env && env[0] == '0') && scene.n_boxes == 0 && scene.n <= 65535 && scene.grid_entries < (1u << kLds64BeginBits) &&
            scene.grid_max_cell_entries < (1u << kLds64CountBits) && bytes <= 28 * 1024 && n_cells > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_lds, render_f64_stream_kernel<false, true>, 256, bytes) == hipSuccess &&
            occ_lds >= occ)
            plan->lds_bytes = bytes;
    }
    // a chunk holds at least 256 samples: one returning atomic per chunk on the queue's counter, which serves
    // some 80 M of them per second
    const uint32_t spp = (uint32_t)cam.spp;
    plan->chunk_pixels = (256u + spp - 1u) / spp;
    const long long chunks = (npix + plan->chunk_pixels - 1) / plan->chunk_pixels;
    const long long blocks_needed = (chunks + 3) / 4; // a wave needs a chunk to have anything to do
    if (blocks_needed < plan->grid) plan->grid = (int32_t)blocks_needed;
    plan->ring_bytes = (size_t)plan->grid * 4u * kF64RingEntries * sizeof(RingEntry);
    return true;
}

hipError_t launch_render_f64_stream(const CamDev<double>& cam, const SceneDev& scene, const Partition& part,
                                    const RngKey& key, uint32_t flags, const F64StreamPlan& plan, uint32_t* d_color,
                                    double* d_raw, int32_t* d_count, DeviceCounters* d_counters, void* rings,
                                    hipStream_t stream) {
    if (!rings || !d_counters) return hipErrorInvalidValue;
    F64Args args{};
    args.cam = cam;
    args.geom = scene.geom64; args.mat = scene.mat64; args.kind = scene.kind; args.n = scene.n;
    args.rg = ref_grid_of(scene, true);
    args.part = part; args.key = key;
    args.color = d_color; args.raw = d_raw; args.count = d_count; args.counters = d_counters;
    args.rings = (RingEntry*)rings; args.chunk_pixels = plan.chunk_pixels;
    args.n_cells = scene.grid.n[0] * scene.grid.n[1] * scene.grid.n[2];
    args.n_entries = (int32_t)scene.grid_entries;
    args.narrow_geom = (const V4f*)scene.f32.narrow_geom;
    args.narrow_index = scene.f32.narrow_index;
    args.n_narrow_padded = scene.f32.n_narrow_padded;
    { // the ball of the narrow spheres, and what the f32 search may be trusted with
        const double radius = (double)scene.f32.bound[3];
        double size = 0.;
        for (int q = 0; q < 3; ++q) {
            args.ball[q] = (double)scene.f32.bound[q];
            size = std::fmax(size, std::fabs(args.ball[q]));
        }
        args.ball[3] = radius;
        args.narrow_cmax = (float)((std::sqrt(args.ball[0] * args.ball[0] + args.ball[1] * args.ball[1] + args.ball[2] * args.ball[2]) + radius) * 1.000001);
        size += radius;
        args.ball_far2 = 2.1 * radius * radius;
        // an origin up to 2^20 scene sizes away is moved up to the ball with an error far inside the search's brackets
        args.reach = std::isfinite(size) ? 1048576. * size : INFINITY;
        args.filter_ok = (!std::isfinite(radius) || (size >= 1e-12 && size <= 1e12)) ? 1 : 0;
    }
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)plan.grid), dim3(plan.block), plan.lds_bytes, stream, args);
    };
    const bool iterative = (flags & DRAY_FLAG_ITERATIVE_PRODUCT) != 0;
    if (plan.lds_bytes != 0) iterative ? launch(render_f64_stream_kernel<true, true>) : launch(render_f64_stream_kernel<false, true>);
    else iterative ? launch(render_f64_stream_kernel<true, false>) : launch(render_f64_stream_kernel<false, false>);
    return hipGetLastError();
}

#ifdef DRAY_F64_CLOCKS
// Diagnostic builds only: reads and clears the stream kernel's section clocks.
extern "C" int dray_debug_f64_clocks(unsigned long long* out) {
    unsigned long long zero[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dray_f64_clocks), sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(dray_f64_clocks), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

} // namespace dray
