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

// Both rejection samplers for all lanes of a wave (wave-uniform control flow; every lane calls).  `w` is block 0 of
// the lane's stream, `draw` what the lane needs.  Acceptance is decided in f64 with the reference's expressions.
__device__ __forceinline__ Drawn coop_draw64(const Stream& st, const uint32_t (&w)[4], uint32_t draw, uint4* req, int lane) {
    Drawn res = {w[0], w[1], w[2]};
    bool pending = false;
    V3<double> p;
    double l2, qx, qy;
    if (draw == kDraw64UnitVector) {
        pending = !unit_vector_attempt<double>(w, p, l2);
    } else if (draw == kDraw64UnitDisk) { // attempt 0 of the aperture sample: uniforms 2, 3 of stage 0
        res = {w[2], w[3], 0u};
        pending = !unit_disk_attempt<double>(w[2], w[3], qx, qy);
    }
    uint32_t base = 1u; // next block index, the same for every pending lane
    for (unsigned long long pm = __ballot(pending); pm != 0ull; pm = __ballot(pending)) {
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
        const bool ok_vector = unit_vector_attempt<double>(wb, p, l2);
        // a block after the first holds two aperture attempts: uniforms 4b, 4b+1 and 4b+2, 4b+3
        const bool ok_a = unit_disk_attempt<double>(wb[0], wb[1], qx, qy), ok_b = unit_disk_attempt<double>(wb[2], wb[3], qx, qy);
        const bool disk = r.w == kDraw64UnitDisk;
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
    }
    return res;
}

// A lane's walk through the grid, kept across turns of the main loop.
struct Walk64 {
    double A[3], B[3]; // time at which the ray leaves cell k of an axis: k * A + B
    double slack;
    int32_t c[3];      // current cell
    uint32_t e, e_end; // entries of the current cell still to test
    int32_t steps;
    double closest;    // smallest accepted root so far, and its sphere's list position
    int32_t best;
};

// One sphere (centre, radius, position in the caller's list) against the walk's closest hit so far.
__device__ __forceinline__ void test64(V3<double> center, double radius, int32_t k, V3<double> ro, V3<double> rd, Walk64& w) {
    double root;
    if (sphere_root_of<double>(center, radius, ro, rd, root) && (root < w.closest || (root == w.closest && k < w.best))) {
        w.closest = root;
        w.best = k;
    }
}

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

// One finished sample in a wave's ring: the colour that joins the pixel's sum (camera.rs:223-225: 0 + sample).
struct RingEntry {
    double r, g, b;
    double unused; // entries are 32 bytes apart
};
static_assert(sizeof(RingEntry) == 32, "ring entries are 32 bytes apart");
constexpr uint32_t kFifoSlots = 16; // chunks a wave may have grabbed and not begun to retire

#ifndef DRAY_F64_WAVES_PER_EU
#define DRAY_F64_WAVES_PER_EU 4
#endif

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
};

// The copy of the grid a workgroup of the LDS kernel stages at its entry, so compact that four workgroups' copies fit
// a compute unit: the spheres' rows (cx, cy, cz, radius: the f64 geometry table as it is, by list position), then a
// word per cell -- first entry (12 bits) | entry count (4 bits) | list position of the first entry's sphere (16 bits)
// --, then the entries as 16-bit list positions.  A lane entering a cell has its first sphere from the cell's word, a
// lane staying in its cell reads the entry table: two dependent reads per step, both from LDS.
constexpr uint32_t kLds64BeginBits = 12u, kLds64CountBits = 4u;

// Hides where a pointer came from, so that reads through it are not moved across this point.
template <typename T>
__device__ __forceinline__ const DRAY_CONSTANT T* opaque(const DRAY_CONSTANT T* p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (const DRAY_CONSTANT T*)v;
}

template <bool ITERATIVE, bool LDSGRID>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DRAY_F64_WAVES_PER_EU, DRAY_F64_WAVES_PER_EU)))
void render_f64_stream_kernel(F64Args) {
    const DRAY_CONSTANT F64Args* ka = (const DRAY_CONSTANT F64Args*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ double lds_geom[]; // LDSGRID: rows of four doubles by list position, then the cell words, then the entries
    if constexpr (LDSGRID) {
        const int32_t n4 = 4 * ka->n, n_cells = ka->n_cells, n_entries = ka->n_entries;
        const double* geom = ka->geom;
        const uint2* cells = ka->rg.grid.cells;
        const int32_t* entry_index = ka->rg.entry_index;
        uint32_t* lds_cells = (uint32_t*)(lds_geom + n4);
        uint16_t* lds_entries = (uint16_t*)(lds_cells + n_cells);
        for (int32_t k = threadIdx.x; k < n4; k += blockDim.x) lds_geom[k] = geom[k];
        for (int32_t k = threadIdx.x; k < n_cells; k += blockDim.x) {
            const uint2 rec = cells[k];
            const uint32_t first = rec.y != 0u ? (uint32_t)entry_index[rec.x] : 0u;
            lds_cells[k] = rec.x | (rec.y << kLds64BeginBits) | (first << (kLds64BeginBits + kLds64CountBits));
        }
        for (int32_t k = threadIdx.x; k < n_entries; k += blockDim.x) lds_entries[k] = (uint16_t)entry_index[k];
        __syncthreads(); // the only block-wide wait: once, before any work is drawn
    }
    // the cell's record: first entry, entry count, and (LDSGRID) the first entry's sphere
    auto cell_of = [&](size_t ci, uint32_t& first_sphere) -> uint2 {
        if constexpr (LDSGRID) {
            const uint32_t rec = ((const uint32_t*)(lds_geom + 4 * ka->n))[ci];
            first_sphere = rec >> (kLds64BeginBits + kLds64CountBits);
            return make_uint2(rec & ((1u << kLds64BeginBits) - 1u), (rec >> kLds64BeginBits) & ((1u << kLds64CountBits) - 1u));
        } else {
            first_sphere = 0u;
            return ka->rg.grid.cells[ci];
        }
    };
    __shared__ uint4 lds_req[4][64];           // coop_draw64: streams handed to server lanes, one set of slots per wave
    __shared__ uint32_t lds_fifo[4][kFifoSlots]; // first pixel of the chunks the wave has grabbed, oldest first
    __shared__ double lds_stage[4][64][3];       // the colours of the samples being retired
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    const uint32_t W = (uint32_t)ka->cam.W, spp = (uint32_t)ka->cam.spp, npix = (uint32_t)ka->part.n_rows * W;
    constexpr bool iterative = ITERATIVE; // how a sample's attenuations are multiplied (camera.rs:130 or :274)
    RingEntry* ring = ka->rings + (size_t)(blockIdx.x * 4u + (uint32_t)wv) * kF64RingEntries;
    const int32_t max_steps = ka->rg.grid.n[0] + ka->rg.grid.n[1] + ka->rg.grid.n[2] + 4;
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
    bool dead = false;     // the path was absorbed: the sample ends black at the lane's next turn
    bool fallback = false; // the closest hit of this ray comes from the scan of the whole list
    Walk64 wk = {};
    wk.closest = INFINITY;
    wk.best = 0x7fffffff;

    // wave state (the same in every lane)
    uint32_t issued = 0u, retired = 0u;       // samples of the wave's stream handed out / added to their pixels
    uint32_t grabbed = 0u, begun = 0u;        // chunks taken from the queue / chunks whose retirement has begun
    uint32_t iss_p0 = 0u, iss_q = 0u, iss_n = 0u;     // the chunk being handed out: first pixel, next sample, samples
    uint32_t ret_p0 = 0u, ret_dp = 0u, ret_np = 0u, ret_s = 0u; // the chunk being retired: first pixel, pixel, pixels; sample
    size_t ret_out = 0;
    double acc = 0.;                          // running sum of the pixel being retired: lane 0 red, 1 green, 2 blue
    unsigned long long n_samples = 0ull;
    bool exhausted = false;

#ifdef DRAY_F64_CLOCKS
    unsigned long long clk[32] = {};
    unsigned long long clk_t = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        ka = opaque(ka);
        F64_COUNT(16, 1);                              // turns
        F64_COUNT(17, __popcll(__ballot(has)));        // lanes carrying a sample
        // ---- walk steps, while enough lanes are on their way --------------------------------------------------
        {
            const int quorum = exhausted ? 1 : DRAY_F64_WALK_QUORUM;
            do { // at least one step a turn: fewer lanes than a quorum may be all the wave has
                F64_COUNT(18, 1);                              // walk steps
                F64_COUNT(19, __popcll(__ballot(walking)));    // lanes walking at them
                if (walking) {
                    uint32_t sphere = 0u; // LDSGRID: list position of the sphere this step tests
                    bool entered = false;
                    if (wk.e == wk.e_end) { // this cell is done: stop, or step into the next one
                        const double tx = (double)wk.c[0] * wk.A[0] + wk.B[0], ty = (double)wk.c[1] * wk.A[1] + wk.B[1],
                                     tz = (double)wk.c[2] * wk.A[2] + wk.B[2];
                        const double t_exit = fmin(tx, fmin(ty, tz));
                        const bool stop = wk.closest < t_exit * (1. - 1e-9) - wk.slack; // nothing beyond this cell can be closer
                        const int ax = (tx <= ty && tx <= tz) ? 0 : (ty <= tz ? 1 : 2);
                        const double Aax = ax == 0 ? wk.A[0] : (ax == 1 ? wk.A[1] : wk.A[2]);
                        const int32_t dir = Aax < 0. ? -1 : 1;
                        wk.c[0] += ax == 0 ? dir : 0;
                        wk.c[1] += ax == 1 ? dir : 0;
                        wk.c[2] += ax == 2 ? dir : 0;
                        const int32_t moved = ax == 0 ? wk.c[0] : (ax == 1 ? wk.c[1] : wk.c[2]);
                        const int32_t limit = ax == 0 ? ka->rg.grid.n[0] : (ax == 1 ? ka->rg.grid.n[1] : ka->rg.grid.n[2]);
                        const bool inside = moved >= 0 && moved < limit;
                        bool go = inside && !stop;
                        if (go && ++wk.steps > max_steps) { // cannot happen on a straight line: take the whole list
                            fallback = true;
                            go = false;
                        }
                        walking = go;
                        const uint2 cellrec = cell_of(go ? ((size_t)wk.c[2] * ka->rg.grid.n[1] + wk.c[1]) * ka->rg.grid.n[0] + wk.c[0] : 0, sphere);
                        wk.e = cellrec.x;
                        wk.e_end = go ? cellrec.x + cellrec.y : cellrec.x;
                        entered = true;
                    }
                    if constexpr (LDSGRID) {
                        if (!entered && wk.e != wk.e_end) // (beside the other lanes' cell read: the two do not depend on one another)
                            sphere = ((const uint16_t*)((const uint32_t*)(lds_geom + 4 * ka->n) + ka->n_cells))[wk.e];
                        if (wk.e != wk.e_end) {
                            const double* eg = lds_geom + 4 * (size_t)sphere;
                            test64(V3<double>{eg[0], eg[1], eg[2]}, eg[3], (int32_t)sphere, ro, rd, wk);
                            ++wk.e;
                        }
                    } else if (wk.e != wk.e_end) {
                        const double* eg = ka->rg.entry_geom64 + 4 * (size_t)wk.e;
                        test64(V3<double>{eg[0], eg[1], eg[2]}, eg[3], ka->rg.entry_index[wk.e], ro, rd, wk);
                        ++wk.e;
                    }
                }
            } while (__popcll(__ballot(walking)) >= quorum);
        }

        F64_CLOCK(0);
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
                n_ok = (int)min(oldest, 64u);
            }
            if (n_ok == 0) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the lanes' stores to the ring have completed
            const RingEntry* e = ring + ((retired + (uint32_t)lane) & (kF64RingEntries - 1u));
            // (lanes past n_ok read entries nobody uses.)  The colours go through LDS to lanes 0, 1, 2, which add up
            // one channel each, sample after sample: one load and one addition per sample
            lds_stage[wv][lane][0] = e->r;
            lds_stage[wv][lane][1] = e->g;
            lds_stage[wv][lane][2] = e->b;
            const int ch = lane < 3 ? lane : 2;
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
                    place_of(ret_p0 + ret_dp, ret_out, word);
                    acc = ka->raw ? ka->raw[3 * ret_out + (size_t)ch] : 0.;
                }
                const int take = min(n_ok - k, (int)(spp - ret_s));
#pragma unroll 8
                for (int t = 0; t < take; ++t) acc += lds_stage[wv][k + t][ch]; // colors.rs:38
                k += take;
                ret_s += (uint32_t)take;
                if (ret_s == spp) { // the pixel is complete
                    const double a0 = readlane_f64(acc, 0), a1 = readlane_f64(acc, 1), a2 = readlane_f64(acc, 2);
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
            retired += (uint32_t)n_ok;
            n_samples += (unsigned long long)n_ok;
        }

        F64_CLOCK(3);
        // ---- lanes whose walk is over: the hit (hittable_list.rs:60-74 / sphere.rs:49-55), or the sky -----------
        // (which sphere or box and how far is all that is settled here; point and normal are made where the scatter
        // needs them, after the wave's draws, so that they do not sit in registers through those)
        const bool ready = has && !walking && !camera;
        bool scatters = false, ended = false, is_box = false;
        uint32_t mk = 0xFFu;
        V3<double> c = {0., 0., 0.};
        if (ready) {
            if (depth <= 0 || dead) { // camera.rs:265-267 / :120-123, or absorbed (camera.rs:276): black
                ended = true;
            } else {
                if (fallback) { // the reference's own scan, hittable_list.rs:65-70
                    wk.closest = INFINITY;
                    wk.best = 0x7fffffff;
                    for (int32_t k = 0; k < ka->n; ++k) {
                        double root;
                        if (sphere_root<double>(ka->geom, k, ro, rd, root) && root < wk.closest) {
                            wk.closest = root;
                            wk.best = k;
                        }
                    }
                }
                is_box = boxes_closest<double>(BoxList{ka->rg.box64, ka->rg.box_index, ka->rg.n_boxes}, ro, rd, wk.closest, wk.best);
                if (wk.best != 0x7fffffff) {
                    mk = ka->kind[wk.best];
                    scatters = true;
                } else {
                    V3<double> ud = vunit(rd); // camera.rs:279-281
                    double a = 0.5 * (ud.y + 1.0);
                    c = vadd(vscale(V3<double>{1.0, 1.0, 1.0}, 1.0 - a), vscale(V3<double>{0.5, 0.7, 1.0}, a));
                    ended = true;
                }
            }
        }

        F64_CLOCK(1);
        F64_COUNT(20, __popcll(__ballot(ready)));      // lanes shading
        // ---- samples that end here: their colour goes to its place in the wave's ring ---------------------------
        if (ended) {
            if constexpr (iterative) {
                c = vmul(att, c); // camera.rs:121,135,142
            } else {
                while (sp > 0) { // camera.rs:274, innermost product first; four albedos are fetched at a time
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
                }
            }
            RingEntry* e = ring + (item & (kF64RingEntries - 1u));
            e->r = 0. + c.x; // camera.rs:223-225: pixel_color = (0, 0, 0) + sample
            e->g = 0. + c.y;
            e->b = 0. + c.z;
            dead = false;
            has = false;
        }

        F64_CLOCK(2);
        // ---- refill: free lanes take the next samples of the stream (camera.rs:217-224) ------------------------
        {
            bool need = !has;
            unsigned long long want = __ballot(need);
            while (want != 0ull) {
                if (iss_q == iss_n && !exhausted) { // the next chunk of pixels
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
                        iss_n = min(ka->chunk_pixels, npix - lo) * spp;
                    }
                }
                if (exhausted) break;
                const uint32_t room = kF64RingEntries - (issued - retired); // a ring's length ahead of the oldest at most
                const uint32_t avail = min(iss_n - iss_q, room);
                const uint32_t wanted = (uint32_t)__popcll(want);
                const uint32_t take = wanted < avail ? wanted : avail;
                if (take == 0u) break;
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
                issued += take;
                want = __ballot(need);
            }
        }
        if (__ballot(has) == 0ull && retired == issued) break; // nothing in flight, nothing left to take

        F64_CLOCK(4);
        // ---- the first uniforms of the next step, and the rejection samplers, by the whole wave -----------------
        const bool draws = scatters || camera;
        const Stream st = {ka->key.k0, ka->key.k1, pw, ka->key.sample_offset + smp, camera ? 0u : (uint32_t)(ka->cam.max_depth - depth + 1)};
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (draws) st.block(0, w);
        const uint32_t draw = (scatters && (mk == DRAY_MAT_LAMBERTIAN || mk == DRAY_MAT_METAL)) ? (uint32_t)kDraw64UnitVector
                              : (camera && ka->cam.defocus != 0)                                      ? (uint32_t)kDraw64UnitDisk
                                                                                                  : (uint32_t)kDraw64None;
        F64_CLOCK(5);
        const Drawn got = coop_draw64(st, w, draw, lds_req[wv], lane);
        F64_CLOCK(6);

        // ---- scatter: material.rs -------------------------------------------------------------------------
        bool fresh = false;
        if (scatters) {
            Hit<double> rec; // sphere.rs:49-55
            if (is_box) box_record<double>(ro, rd, wk.closest, wk.best, rec);
            else fill_hit<double>(ka->geom, wk.best, ro, rd, wk.closest, rec);
            V3<double> so = ro, sd;
            const bool scattered = scatter_with<double>(
                mk, ka->mat + 4 * rec.index, rd, rec,
                [&]() { // vec3.rs:82 for the accepted attempt
                    const uint32_t ww[4] = {got.a, got.b, got.c, 0u};
                    V3<double> p;
                    double lensq;
                    unit_vector_attempt<double>(ww, p, lensq);
                    return vdiv(p, vsqrt(lensq));
                },
                [&]() { return u01<double>(w[0]); }, so, sd);
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
                dead = true; // camera.rs:276
            }
        }

        F64_CLOCK(7);
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
        // ---- a new ray: the spheres every ray meets, then its first cell ------------------------------------------
        if (fresh) {
            my_segments++;
            wk.closest = INFINITY;
            wk.best = 0x7fffffff;
            wk.e = wk.e_end = 0u;
            wk.steps = 0;
            fallback = false;
            walking = false;
            const double o[3] = {ro.x, ro.y, ro.z}, d[3] = {rd.x, rd.y, rd.z};
            bool finite = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) finite = finite && fabs(o[a]) < 1e300 && fabs(d[a]) < 1e300; // false for NaN too
            if (!finite) {
                fallback = true;
            } else {
                {
                    const DRAY_CONSTANT double* ag = as_constant(ka->rg.always_geom64);
                    const DRAY_CONSTANT int32_t* ai = as_constant(ka->rg.always_index);
                    // the wide and the loose spheres (scalar loads), four at a time so that their chains of f64
                    // operations run side by side
                    const int32_t n_always = ka->rg.n_always;
                    int32_t q = 0;
                    for (; q + 4 <= n_always; q += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            test64(V3<double>{ag[4 * (q + u)], ag[4 * (q + u) + 1], ag[4 * (q + u) + 2]}, ag[4 * (q + u) + 3],
                                   ai[q + u], ro, rd, wk);
                    }
                    for (; q < n_always; ++q)
                        test64(V3<double>{ag[4 * q], ag[4 * q + 1], ag[4 * q + 2]}, ag[4 * q + 3], ai[q], ro, rd, wk);
                }
                if (ka->rg.grid.n[0] > 0) {
                    double t_in = 0., t_out = INFINITY, bmax = 0.;
                    bool inside_slabs = true;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (fabs(d[a]) > 1e-250) {
                            const double inv = 1.0 / d[a];
                            wk.A[a] = (double)ka->rg.grid.cell[a] * inv;
                            const double t0 = ((double)ka->rg.grid.lo[a] - o[a]) * inv, t1 = ((double)ka->rg.grid.hi[a] - o[a]) * inv;
                            wk.B[a] = d[a] >= 0. ? t0 + wk.A[a] : t0; // leaving cell k towards larger indices happens at plane k + 1
                            t_in = fmax(t_in, fmin(t0, t1));
                            t_out = fmin(t_out, fmax(t0, t1));
                            bmax = fmax(bmax, fmax(fabs(t0), fabs(wk.B[a])));
                        } else { // parallel to the axis' planes: never crosses one; misses the grid if outside its slab
                            wk.A[a] = 0.;
                            wk.B[a] = INFINITY;
                            if (o[a] < (double)ka->rg.grid.lo[a] || o[a] > (double)ka->rg.grid.hi[a]) inside_slabs = false;
                        }
                    }
                    wk.slack = bmax * 1e-9;
                    if (inside_slabs && t_in <= t_out + wk.slack) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const double at = floor((o[a] + d[a] * t_in - (double)ka->rg.grid.lo[a]) * (double)ka->rg.grid.inv_cell[a]);
                            wk.c[a] = (int32_t)fmin(fmax(at, 0.), (double)(ka->rg.grid.n[a] - 1));
                        }
                        uint32_t unused;
                        const uint2 cellrec = cell_of(((size_t)wk.c[2] * ka->rg.grid.n[1] + wk.c[1]) * ka->rg.grid.n[0] + wk.c[0], unused);
                        wk.e = cellrec.x;
                        wk.e_end = cellrec.x + cellrec.y;
                        walking = true;
                    }
                }
            }
        }
        F64_CLOCK(9);
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
        atomicAdd(&ka->counters->samples, n_samples);
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
        const size_t bytes = (size_t)scene.n * 32u + n_cells * 4u + ((scene.grid_entries * 2u + 15u) & ~(size_t)15u);
        const char* env = std::getenv("DRAY_GRID_LDS");
        int occ_lds = 0;
        if (!(env && env[0] == '0') && scene.n_boxes == 0 && scene.n <= 65535 && scene.grid_entries < (1u << kLds64BeginBits) &&
            scene.grid_max_cell_entries < (1u << kLds64CountBits) && bytes <= 48 * 1024 &&
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
