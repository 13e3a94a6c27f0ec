// C ABI of libdray_hip.so (declared in include/dray.h): argument checking, the
// host-side camera set-up, scene upload and launch sequencing.  Host code only;
// the kernels live in render_ref.hip, render_f64.hip and render_f32.hip.  Build with
// -ffp-contract=off: the camera maths below must round like the reference's
// plain f64 expressions (src/raytracer/camera.rs:169-206).
#pragma clang fp contract(off)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "dray_internal.h"

using namespace dray;

// What one call on a scene needs for itself: counters (the trace kernels' work-queue head among them), the three
// events its timings come from, and what its statistics cannot be read off the device.  A scene keeps the last
// kCallSlots calls' slots, so that a caller may enqueue several calls and ask for their statistics afterwards.
constexpr int kCallSlots = 16;
struct CallSlot {
    DeviceCounters* d_counters = nullptr; // this slot's counters (or, for the allocation-free call, the static workspace's)
    DeviceCounters* own_counters = nullptr;
    DeviceCounters* h_counters = nullptr; // what the call counted, in page-locked host memory the device writes to
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr}; // trace begins, trace ends, results are in the caller's buffers
    uint64_t ticket = 0;                            // number of the call that holds the slot (1, 2, ...); 0: none yet
    dray_stats base{};                              // the fields known on the host
    bool samples_from_frame = false;                // f32 mode without depth: no kernel counts the samples
    std::chrono::steady_clock::time_point t_begin;
};

// Where an f32 trace kernel runs: fixed-point colour sums of its own and a stream of its own.  A scene has two and
// gives them to its calls in turn, so that the trace kernel of a call -- it reads nothing but the scene -- may start
// while the previous call's kernel is still running out, instead of after its last wave.
struct TraceSpace {
    unsigned long long* d_accum = nullptr;
    size_t accum_capacity = 0;    // in u64 words
    hipStream_t side = nullptr;
    hipEvent_t free_after = nullptr; // recorded when the call that last used the space has read its sums
};

struct dray_scene {
    int32_t device = 0;
    SceneDev dev;
    DeviceCounters* d_counter_block = nullptr; // kCallSlots counters, 256 bytes apart
    CallSlot slots[kCallSlots];
    uint64_t calls = 0;                        // calls that have taken a slot
    TraceSpace space[2];
    int next_space = 0;
    DeviceCounters* h_counter_block = nullptr; // the slots' host-side counters
    void* d_rings = nullptr;               // the f64 stream kernel's rings of finished samples (one per resident wave)
    size_t rings_bytes = 0;
    // Calls on one scene take turns on the host (`mu`, for the length of the enqueueing), and a call enqueued on
    // another stream than the one before it first makes its stream wait for that call's last event: the calls
    // of a scene reach the caller's buffers in the order they were made.
    std::mutex mu;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;
};

// The last kernel of a call: the counters go to the slot's place in page-locked host memory, where collect_call reads
// them once the call's last event has come.
__global__ void counters_to_host_kernel(const DeviceCounters* __restrict__ d, DeviceCounters* __restrict__ h) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        h->segments = d->segments;
        h->samples = d->samples;
        h->next_item = d->next_item;
        h->walk_steps = d->walk_steps;
    }
}
static hipError_t counters_to_host(const CallSlot* slot, hipStream_t stream) {
    hipLaunchKernelGGL(counters_to_host_kernel, dim3(1), dim3(64), 0, stream, slot->d_counters, slot->h_counters);
    return hipGetLastError();
}

// Workspace of the allocation-free entry point (dray_render_device_noalloc): the f32 kernel's fixed-point sums
// and the counters of one render, in device memory that belongs to the code object -- no hipMalloc on any call.
// Every GPU the library runs on holds its own copy of the array (25 MiB) until the process ends.  In host code
// the array's name is only a handle for hipGetSymbolAddress: kernels and memsets are given the per-device
// address that call returns (noalloc_workspace below), never the name.
constexpr size_t kNoAllocMaxPixels = DRAY_NOALLOC_MAX_PIXELS;
constexpr size_t kNoAllocCounterWords = 32;                              // 256 B: the DeviceCounters, padded
constexpr size_t kNoAllocWords = kNoAllocCounterWords + 3 * kNoAllocMaxPixels; // counters, then three sum planes
static __device__ __align__(256) unsigned long long scratch[kNoAllocWords];

namespace {

thread_local char g_detail[256] = "";

int hip_fail(hipError_t e, const char* what) {
    (void)hipGetLastError(); // the runtime remembers the error; the next kernel launch's check must not find it
    snprintf(g_detail, sizeof g_detail, "%s: %s", what, hipGetErrorString(e));
    return DRAY_ERR_HIP;
}

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return hip_fail(e__, #expr); \
    } while (0)

// RAII: make `device` current for the scope, restore the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

struct D3 {
    double x, y, z;
};
D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
D3 scale(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
D3 divs(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
D3 neg(D3 a) { return {-a.x, -a.y, -a.z}; }
double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
D3 unit(D3 a) { return divs(a, std::sqrt(dot(a, a))); }
D3 ld(const double* p) { return {p[0], p[1], p[2]}; }
void stv(double* p, D3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

double degrees_to_radians(double d) { return d * 3.14159265358979323846264338327950288 / 180.0; } // prelude.rs:17-20

// the list entries the library has a device form for: spheres of the five materials, and bounding boxes
bool entry_ok(const dray_sphere& e) {
    return e.shape == DRAY_SHAPE_BOX || (e.shape == DRAY_SHAPE_SPHERE && e.kind <= DRAY_MAT_TRANSPARENT);
}

bool camera_ok(const dray_camera* cam) {
    return cam && cam->image_width >= 1 && std::isfinite(cam->aspect_ratio) && cam->aspect_ratio > 0.;
}

// Philox4x32-10 on the host, for the scene recipe only.
void philox_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct SceneStream {
    uint32_t key[2];
    uint32_t n = 0;
    double next() {
        const uint32_t ctr[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, n >> 2};
        uint32_t w[4];
        philox_host(ctr, key, w);
        const double u = (double)(w[n & 3u] >> 8) * (1.0 / 16777216.0);
        ++n;
        return u;
    }
    double range(double lo, double hi) { return lo + (hi - lo) * next(); } // prelude.rs:27-30
};

void put_sphere(dray_sphere* o, double x, double y, double z, double r, uint32_t kind, double a0, double a1, double a2,
                double f) {
    std::memset(o, 0, sizeof *o);
    o->center[0] = x; o->center[1] = y; o->center[2] = z;
    o->radius = r;
    o->kind = kind;
    o->albedo[0] = a0; o->albedo[1] = a1; o->albedo[2] = a2;
    o->fuzz_or_ior = f;
}

template <typename T>
CamDev<T> make_cam(const dray_camera& cam, const dray_camera_derived& d) {
    CamDev<T> c{};
    for (int q = 0; q < 3; ++q) {
        c.center[q] = (T)d.center[q];
        c.p00[q] = (T)d.pixel00_loc[q];
        c.du[q] = (T)d.pixel_delta_u[q];
        c.dv[q] = (T)d.pixel_delta_v[q];
        c.ddu[q] = (T)d.defocus_disk_u[q];
        c.ddv[q] = (T)d.defocus_disk_v[q];
    }
    c.W = cam.image_width;
    c.H = d.image_height;
    c.spp = cam.samples_per_pixel;
    c.max_depth = cam.max_depth;
    c.defocus = cam.defocus_angle <= 0. ? 0 : 1; // camera.rs:247
    return c;
}

constexpr double kWideRadius = 32.0; // spheres at least this large are intersected in f64 by the f32 kernel

void scene_free(dray_scene* s) {
    if (!s) return;
    DeviceGuard g(s->device);
    SceneDev& d = s->dev;
    (void)hipFree(d.geom64); (void)hipFree(d.mat64); (void)hipFree(d.kind); (void)hipFree(d.kind32);
    (void)hipFree(d.geom32); (void)hipFree(d.mat32);
    (void)hipFree(d.narrow_geom); (void)hipFree(d.filter_xy); (void)hipFree(d.filter_zn);
    (void)hipFree(d.narrow_index); (void)hipFree(d.wide_index);
    (void)hipFree(d.grid_cells); (void)hipFree(d.grid_entry_geom); (void)hipFree(d.grid_entry_pos); (void)hipFree(d.grid_loose);
    (void)hipFree(d.box64); (void)hipFree(d.box_index);
    (void)hipFree(d.loose_geom); (void)hipFree(d.loose_index); (void)hipFree(d.wide_geom64); (void)hipFree(d.wide_axial);
    (void)hipFree(d.entry_geom64); (void)hipFree(d.entry_index); (void)hipFree(d.always_geom64); (void)hipFree(d.always_index);
    (void)hipFree(s->d_counter_block); (void)hipFree(s->d_rings);
    for (TraceSpace& t : s->space) {
        (void)hipFree(t.d_accum);
        if (t.side) (void)hipStreamDestroy(t.side);
    }
    if (s->h_counter_block) (void)hipHostFree(s->h_counter_block);
    for (CallSlot& c : s->slots)
        for (auto& e : c.ev)
            if (e) (void)hipEventDestroy(e);
    delete s;
}

// Host-side build of the uniform grid the f32 kernel's grid scan walks.
struct GridHost {
    GridDev dev{};
    std::vector<uint2> cells;
    std::vector<float> entry_geom;
    std::vector<int32_t> entry_pos, loose;
};

constexpr int kGridMaxAxis = 256;      // cells per axis
constexpr size_t kGridSpanLimit = 512; // a sphere overlapping more cells than this stays out of the grid

// ng: narrow_geom rows (cx, cy, cz, r*r) of the first n_narrow positions.
void build_grid(const std::vector<float>& ng, int32_t n_narrow, GridHost& g) {
    g = GridHost();
    if (n_narrow <= 0) return;
    std::vector<double> rad(n_narrow);
    std::vector<char> odd(n_narrow, 0); // kept out of the grid: not finite (the exact test never accepts those), or outsized
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int32_t n_regular = 0;
    std::vector<double> finite_rad;
    for (int32_t p = 0; p < n_narrow; ++p) {
        rad[p] = std::sqrt((double)ng[4 * p + 3]) * (1.0 + 1e-6);
        if (!std::isfinite(rad[p]) || !std::isfinite(ng[4 * p]) || !std::isfinite(ng[4 * p + 1]) ||
            !std::isfinite(ng[4 * p + 2]))
            odd[p] = 1;
        else
            finite_rad.push_back(rad[p]);
    }
    // A few spheres much larger than the rest would stretch the grid box over space the others leave empty and be
    // listed in many cells: rays would step through all of it.  They are tested by every ray instead (a test costs
    // less than one step), and the grid is laid over the others.
    if (finite_rad.size() >= 16) {
        double ratio = 3.0;
        int max_big = 8;
        if (const char* env = std::getenv("DRAY_GRID_BIG_RATIO")) ratio = std::atof(env);
        if (const char* env = std::getenv("DRAY_GRID_BIG_MAX")) max_big = std::atoi(env);
        std::nth_element(finite_rad.begin(), finite_rad.begin() + finite_rad.size() / 2, finite_rad.end());
        const double limit = ratio * finite_rad[finite_rad.size() / 2];
        int n_big = 0;
        for (int32_t p = 0; p < n_narrow; ++p) n_big += (!odd[p] && rad[p] > limit) ? 1 : 0;
        if (ratio > 0. && n_big <= max_big)
            for (int32_t p = 0; p < n_narrow; ++p)
                if (!odd[p] && rad[p] > limit) odd[p] = 1;
    }
    for (int32_t p = 0; p < n_narrow; ++p) {
        if (odd[p]) continue;
        ++n_regular;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::fmin(lo[a], (double)ng[4 * p + a] - rad[p]);
            hi[a] = std::fmax(hi[a], (double)ng[4 * p + a] + rad[p]);
        }
    }
    if (n_regular == 0) { // nothing to walk: every sphere goes to the loose list
        for (int32_t p = 0; p < n_narrow; ++p) g.loose.push_back(p);
        g.dev.n_loose = (int32_t)g.loose.size();
        return;
    }
    double ext[3], biggest = 0., reach = 0.;
    for (int a = 0; a < 3; ++a) {
        biggest = std::fmax(biggest, hi[a] - lo[a]);
        reach = std::fmax(reach, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    // every box below is inflated by far more than the f32 rounding of a walk or of the exact test can move a point
    const double pad = 1e-3 * biggest + 1e-5 * reach + 1e-6;
    for (int a = 0; a < 3; ++a) {
        lo[a] -= pad;
        hi[a] += pad;
        ext[a] = hi[a] - lo[a];
    }
    // cells aimed at per sphere; DRAY_GRID_CELLS_PER_SPHERE overrides the default for tuning runs
    double density = 2.0; // measured on the 485-sphere frame: 0.5 -> 28.1 ms, 1 -> 23.1, 2 -> 22.0, 4 -> 23.2, 8 -> 24.9 (200 spp)
    if (const char* env = std::getenv("DRAY_GRID_CELLS_PER_SPHERE")) {
        const double v = std::atof(env);
        if (v > 0.01 && v < 1000.) density = v;
    }
    double target = density * n_narrow;
    for (int attempt = 0;; ++attempt) {
        int32_t n[3];
        const double scale = std::cbrt(target / (ext[0] * ext[1] * ext[2]));
        for (int a = 0; a < 3; ++a) {
            const double want = std::floor(ext[a] * scale + 0.5);
            n[a] = (int32_t)std::fmin(std::fmax(want, 1.), (double)kGridMaxAxis);
        }
        double cell[3], margin = INFINITY;
        for (int a = 0; a < 3; ++a) {
            cell[a] = ext[a] / n[a];
            margin = std::fmin(margin, cell[a]);
        }
        margin = margin / 256. + 1e-5 * reach; // a sphere is listed wherever its ball grown by this overlaps
        const size_t n_cells = (size_t)n[0] * n[1] * n[2];
        std::vector<uint32_t> count(n_cells, 0u);
        std::vector<char> is_loose(odd);
        size_t total = 0;
        auto for_cells = [&](int32_t p, auto&& visit) { // cells the grown ball of sphere p overlaps
            const double r = rad[p] + margin;
            int32_t c0[3], c1[3];
            for (int a = 0; a < 3; ++a) {
                c0[a] = (int32_t)std::fmax(0., std::floor(((double)ng[4 * p + a] - r - lo[a]) / cell[a]));
                c1[a] = (int32_t)std::fmin((double)n[a] - 1., std::floor(((double)ng[4 * p + a] + r - lo[a]) / cell[a]));
            }
            for (int32_t z = c0[2]; z <= c1[2]; ++z)
                for (int32_t y = c0[1]; y <= c1[1]; ++y)
                    for (int32_t x = c0[0]; x <= c1[0]; ++x) {
                        const int32_t c[3] = {x, y, z};
                        double d2 = 0.;
                        for (int a = 0; a < 3; ++a) {
                            const double b0 = lo[a] + c[a] * cell[a], b1 = b0 + cell[a], v = (double)ng[4 * p + a];
                            const double d = v < b0 ? b0 - v : (v > b1 ? v - b1 : 0.);
                            d2 += d * d;
                        }
                        if (d2 <= r * r) visit(((size_t)z * n[1] + y) * n[0] + x);
                    }
        };
        for (int32_t p = 0; p < n_narrow; ++p) {
            if (is_loose[p]) continue;
            size_t span = 0;
            for_cells(p, [&](size_t) { ++span; });
            if (span > kGridSpanLimit) {
                is_loose[p] = 1;
                continue;
            }
            for_cells(p, [&](size_t c) { ++count[c]; });
            total += span;
        }
        // many entries per sphere mean the cells are too fine for the spheres' sizes: coarsen and try again
        if (total > 24 * (size_t)n_narrow + 1024 && attempt < 12 && n_cells > 1) {
            target /= 4.;
            continue;
        }
        g.cells.resize(n_cells);
        uint32_t at = 0;
        for (size_t c = 0; c < n_cells; ++c) {
            g.cells[c] = make_uint2(at, 0u);
            at += count[c];
        }
        g.entry_geom.resize(4 * (size_t)total);
        g.entry_pos.resize(total);
        for (int32_t p = 0; p < n_narrow; ++p) { // ascending p: a cell's entries come out in insertion order
            if (is_loose[p]) {
                g.loose.push_back(p);
                continue;
            }
            for_cells(p, [&](size_t c) {
                const size_t e = (size_t)g.cells[c].x + g.cells[c].y++;
                for (int q = 0; q < 4; ++q) g.entry_geom[4 * e + q] = ng[4 * p + q];
                g.entry_pos[e] = p;
            });
        }
        for (int a = 0; a < 3; ++a) {
            g.dev.lo[a] = (float)lo[a];
            g.dev.hi[a] = (float)hi[a];
            g.dev.n[a] = n[a];
            g.dev.cell[a] = (float)cell[a];
            g.dev.inv_cell[a] = (float)(1.0 / cell[a]);
        }
        g.dev.n_loose = (int32_t)g.loose.size();
        return;
    }
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return DRAY_OK;
}

// Splits the caller's list into the spheres the f32 kernel scans in f32 ("narrow": exact-table rows cx, cy, cz, r*r
// and their positions in the caller's list) and those it intersects in f64 ("wide").
void classify_spheres(const dray_sphere* sp, size_t n, std::vector<float>& ng, std::vector<int32_t>& nidx,
                      std::vector<int32_t>& widx) {
    for (size_t k = 0; k < n; ++k) {
        if (sp[k].shape == DRAY_SHAPE_BOX) continue; // a bounding box: in neither sphere list
        const double r = std::fmax(sp[k].radius, 0.); // sphere.rs:14
        if (r >= kWideRadius) {
            widx.push_back((int32_t)k);
        } else {
            nidx.push_back((int32_t)k);
            ng.push_back((float)sp[k].center[0]); ng.push_back((float)sp[k].center[1]); ng.push_back((float)sp[k].center[2]);
            ng.push_back((float)(r * r));
        }
    }
}

int grid_inspect_impl(const dray_sphere* sp, size_t n, dray_grid_info* info, uint32_t* cell_begin, int32_t* entry_sphere,
                      int32_t* loose_sphere) {
    if (!info || (!sp && n) || n > 0x7fffffffu) return DRAY_ERR_INVALID_ARG;
    for (size_t k = 0; k < n; ++k) // the lists dray_scene_create takes, and no others
        if (!entry_ok(sp[k])) return DRAY_ERR_UNSUPPORTED;
    std::vector<float> ng;
    std::vector<int32_t> nidx, widx;
    classify_spheres(sp, n, ng, nidx, widx);
    GridHost g;
    build_grid(ng, (int32_t)nidx.size(), g);
    std::memset(info, 0, sizeof *info);
    for (int a = 0; a < 3; ++a) {
        info->lo[a] = g.dev.lo[a];
        info->hi[a] = g.dev.hi[a];
        info->n[a] = g.dev.n[a];
    }
    info->n_narrow = (int32_t)nidx.size();
    info->n_wide = (int32_t)widx.size();
    info->n_loose = (int32_t)g.loose.size();
    info->n_entries = g.entry_pos.size();
    for (size_t c = 0; c < g.cells.size(); ++c) {
        if (g.cells[c].y > info->max_cell_entries) info->max_cell_entries = g.cells[c].y;
        if (cell_begin) cell_begin[c] = g.cells[c].x;
    }
    if (cell_begin) cell_begin[g.cells.size()] = (uint32_t)g.entry_pos.size();
    if (entry_sphere)
        for (size_t e = 0; e < g.entry_pos.size(); ++e) entry_sphere[e] = nidx[g.entry_pos[e]];
    if (loose_sphere)
        for (size_t k = 0; k < g.loose.size(); ++k) loose_sphere[k] = nidx[g.loose[k]];
    return DRAY_OK;
}

int scene_create_impl(const dray_sphere* sp, size_t n, int32_t device, dray_scene** out) {
    if (!out) return DRAY_ERR_INVALID_ARG;
    *out = nullptr;
    if ((!sp && n) || n > 0x7fffffffu) return DRAY_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return DRAY_ERR_NO_DEVICE;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return DRAY_ERR_INVALID_ARG;
    for (size_t k = 0; k < n; ++k)
        if (!entry_ok(sp[k])) return DRAY_ERR_UNSUPPORTED;

    DeviceGuard g(device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    dray_scene* s = new (std::nothrow) dray_scene();
    if (!s) return DRAY_ERR_INTERNAL;
    s->device = device;
    SceneDev& d = s->dev;
    d.n = (int32_t)n;

    std::vector<double> g64(4 * n), m64(4 * n);
    std::vector<float> g32(4 * n), m32(4 * n), ng;
    std::vector<uint32_t> kind(n);
    std::vector<int32_t> nidx, widx;
    std::vector<double> box64;
    std::vector<int32_t> box_index;
    for (size_t k = 0; k < n; ++k) {
        if (sp[k].shape == DRAY_SHAPE_BOX) {
            // the box itself goes to the box table; its row in the sphere tables is a sphere that no ray hits (the
            // discriminant of a NaN centre is NaN, and every comparison with the root fails), its material
            // Transparent (bounding_box.rs:65,71)
            const dray_box& b = reinterpret_cast<const dray_box&>(sp[k]);
            for (int q = 0; q < 3; ++q) box64.push_back(b.min[q]);
            for (int q = 0; q < 3; ++q) box64.push_back(b.max[q]);
            box_index.push_back((int32_t)k);
            for (int q = 0; q < 3; ++q) {
                g64[4 * k + q] = NAN; g32[4 * k + q] = NAN;
                m64[4 * k + q] = 0.; m32[4 * k + q] = 0.f;
            }
            g64[4 * k + 3] = 0.; g32[4 * k + 3] = 0.f; m64[4 * k + 3] = 0.; m32[4 * k + 3] = 0.f;
            kind[k] = DRAY_MAT_TRANSPARENT;
            continue;
        }
        const double r = std::fmax(sp[k].radius, 0.); // sphere.rs:14
        double f = sp[k].fuzz_or_ior;
        if (sp[k].kind == DRAY_MAT_METAL) f = f < 1.0 ? f : 1.0; // material.rs:67
        for (int q = 0; q < 3; ++q) {
            g64[4 * k + q] = sp[k].center[q];
            m64[4 * k + q] = sp[k].albedo[q];
            g32[4 * k + q] = (float)sp[k].center[q];
            m32[4 * k + q] = (float)sp[k].albedo[q];
        }
        g64[4 * k + 3] = r; m64[4 * k + 3] = f;
        g32[4 * k + 3] = (float)r; m32[4 * k + 3] = (float)f;
        kind[k] = sp[k].kind;
        if (sp[k].kind == DRAY_MAT_DIELECTRIC) {
            // A Dialectric has no albedo (material.rs:108: attenuation (1, 1, 1)), so the first three places of its f64
            // row hold what its scatter derives from ior alone, made here with the reference's operations (IEEE
            // quotients and products are the same on any machine): 1 / ior (material.rs:109) and reflectance's
            // r0 = ((1 - ri) / (1 + ri))^2 (material.rs:95-96) for ri = 1 / ior (front face) and ri = ior (back face).
            const double inv = 1.0 / f;
            double r0f = (1.0 - inv) / (1.0 + inv), r0b = (1.0 - f) / (1.0 + f);
            r0f = r0f * r0f;
            r0b = r0b * r0b;
            m64[4 * k + 0] = inv; m64[4 * k + 1] = r0f; m64[4 * k + 2] = r0b;
        }
    }
    classify_spheres(sp, n, ng, nidx, widx);
    d.n_narrow = (int32_t)nidx.size();
    d.n_wide = (int32_t)widx.size();
    while (nidx.size() % 8 != 0) { // padding entries can never be hit: their discriminant is -1e30 + h*h - |oc|^2 < 0
        nidx.push_back(-1);
        ng.push_back(0.f); ng.push_back(0.f); ng.push_back(0.f); ng.push_back(-1e30f);
    }
    d.n_narrow_padded = (int32_t)nidx.size();
    // Filter tables of the hybrid scan: per two consecutive narrow spheres (cx0, cy0, cx1, cy1) and
    // (cz0, nw0, cz1, nw1) with nw = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2), rounded towards listing (downwards).
    std::vector<float> fxy(2 * nidx.size()), fzn(2 * nidx.size());
    for (size_t p = 0; p < nidx.size(); ++p) {
        const float cx = ng[4 * p + 0], cy = ng[4 * p + 1], cz = ng[4 * p + 2], r2 = ng[4 * p + 3];
        float nwf = 1e30f; // padding: never listed
        if ((int32_t)p < d.n_narrow) {
            const double c2 = (double)cx * cx + (double)cy * cy + (double)cz * cz;
            const double nw = c2 - (double)r2 - 7.62939453125e-6 * (c2 + (double)r2);
            nwf = (float)nw;
            if ((double)nwf > nw) nwf = std::nextafterf(nwf, -INFINITY);
        }
        fxy[2 * p + 0] = cx; fxy[2 * p + 1] = cy;
        fzn[2 * p + 0] = cz; fzn[2 * p + 1] = nwf;
    }

    GridHost grid;
    build_grid(ng, d.n_narrow, grid);

    int rc = DRAY_OK;
    if ((rc = upload(&d.grid_cells, grid.cells)) || (rc = upload(&d.grid_entry_geom, grid.entry_geom)) ||
        (rc = upload(&d.grid_entry_pos, grid.entry_pos)) || (rc = upload(&d.grid_loose, grid.loose))) {
        scene_free(s);
        return rc;
    }
    d.grid = grid.dev;
    d.grid.cells = d.grid_cells;
    d.grid.entry_geom = (const float4*)d.grid_entry_geom;
    d.grid.entry_pos = d.grid_entry_pos;
    d.grid.loose = d.grid_loose;
    d.grid_entries = grid.entry_pos.size();
    for (const uint2& c : grid.cells) d.grid_max_cell_entries = c.y > d.grid_max_cell_entries ? c.y : d.grid_max_cell_entries;
    { // the grid's lists once more with list positions and f64 geometry, for the f64 stream kernel
        std::vector<double> eg(4 * grid.entry_pos.size());
        std::vector<int32_t> ei(grid.entry_pos.size());
        for (size_t e = 0; e < grid.entry_pos.size(); ++e) {
            ei[e] = nidx[grid.entry_pos[e]];
            for (int q = 0; q < 4; ++q) eg[4 * e + q] = g64[4 * (size_t)ei[e] + q];
        }
        std::vector<int32_t> ai(widx);
        for (int32_t p : grid.loose) ai.push_back(nidx[p]);
        std::vector<double> ag(4 * ai.size());
        for (size_t k = 0; k < ai.size(); ++k)
            for (int q = 0; q < 4; ++q) ag[4 * k + q] = g64[4 * (size_t)ai[k] + q];
        d.n_always = (int32_t)ai.size();
        // ... and for the f32 grid scan: the loose spheres' rows in fours, the wide spheres' f64 rows side by side
        std::vector<float> lg;
        std::vector<int32_t> li;
        for (int32_t p : grid.loose) {
            for (int q = 0; q < 4; ++q) lg.push_back(ng[4 * (size_t)p + q]);
            li.push_back(nidx[p]);
        }
        while (li.size() % 4 != 0) { // never hit: the discriminant is -1e30 + h*h - |oc|^2 < 0
            lg.push_back(0.f); lg.push_back(0.f); lg.push_back(0.f); lg.push_back(-1e30f);
            li.push_back(0x3fffffff);
        }
        std::vector<double> wg(4 * widx.size());
        for (size_t k = 0; k < widx.size(); ++k)
            for (int q = 0; q < 4; ++q) wg[4 * k + q] = g64[4 * (size_t)widx[k] + q];
        // the same rows for the f32 kernel's cancellation-free form (F32Tables::wide_axial)
        std::vector<float> wa(8 * widx.size());
        for (size_t k = 0; k < widx.size(); ++k) {
            const double* g = &g64[4 * (size_t)widx[k]];
            const bool axial = std::isfinite(g[0]) && std::isfinite(g[1]) && std::isfinite(g[2]) && std::isfinite(g[3]) &&
                               std::fabs(g[0]) <= g[3] / 32. && std::fabs(g[2]) <= g[3] / 32.;
            wa[8 * k + 0] = (float)g[0]; wa[8 * k + 1] = (float)g[1]; wa[8 * k + 2] = (float)g[2]; wa[8 * k + 3] = (float)(1.0 / g[3]);
            wa[8 * k + 4] = (float)(g[1] + g[3]); wa[8 * k + 5] = (float)(g[1] - g[3]); wa[8 * k + 6] = axial ? 1.f : 0.f; wa[8 * k + 7] = 0.f;
        }
        if ((rc = upload(&d.loose_geom, lg)) || (rc = upload(&d.loose_index, li)) || (rc = upload(&d.wide_geom64, wg)) ||
            (rc = upload(&d.wide_axial, wa))) {
            scene_free(s);
            return rc;
        }
        d.grid.loose_geom = (const float4*)d.loose_geom;
        d.grid.loose_index = d.loose_index;
        d.grid.n_loose_padded = (int32_t)li.size();
        if ((rc = upload(&d.entry_geom64, eg)) || (rc = upload(&d.entry_index, ei)) || (rc = upload(&d.always_geom64, ag)) ||
            (rc = upload(&d.always_index, ai))) {
            scene_free(s);
            return rc;
        }
    }
    // The throughput kernel renders Transparent (material.rs:22-31) as the absorber it amounts to: the ray it hands
    // back meets the same sphere again at every depth left and the sample comes out black.  (Only the statistics
    // differ: that kernel counts one segment where the reference-order kernels walk through the depths left.)
    std::vector<uint32_t> kind32(kind);
    for (uint32_t& k : kind32)
        if (k == DRAY_MAT_TRANSPARENT) k = DRAY_MAT_DEFAULT;
    d.n_boxes = (int32_t)box_index.size();
    if ((rc = upload(&d.box64, box64)) || (rc = upload(&d.box_index, box_index))) {
        scene_free(s);
        return rc;
    }
    if ((rc = upload(&d.geom64, g64)) || (rc = upload(&d.mat64, m64)) || (rc = upload(&d.kind, kind)) ||
        (rc = upload(&d.kind32, kind32)) ||
        (rc = upload(&d.geom32, g32)) || (rc = upload(&d.mat32, m32)) || (rc = upload(&d.narrow_geom, ng)) ||
        (rc = upload(&d.filter_xy, fxy)) || (rc = upload(&d.filter_zn, fzn)) ||
        (rc = upload(&d.narrow_index, nidx)) || (rc = upload(&d.wide_index, widx))) {
        scene_free(s);
        return rc;
    }
    d.f32.narrow_geom = (const float4*)d.narrow_geom;
    d.f32.filter_xy = (const float4*)d.filter_xy;
    d.f32.filter_zn = (const float4*)d.filter_zn;
    d.f32.narrow_index = d.narrow_index;
    d.f32.wide_index = d.wide_index;
    d.f32.geom64 = d.geom64;
    d.f32.wide_geom64 = d.wide_geom64;
    d.f32.wide_axial = (const float4*)d.wide_axial;
    d.grid.entry_index = d.entry_index;
    d.f32.geom32 = (const float4*)d.geom32;
    d.f32.mat32 = (const float4*)d.mat32;
    d.f32.kind = d.kind32;
    d.f32.n_narrow_padded = d.n_narrow_padded;
    d.f32.n_wide = d.n_wide;
    { // the ball the f32 kernel moves a far ray's origin up to: centre of the narrow spheres' box, farthest reach
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        auto finite = [&](int32_t p) {
            return std::isfinite(ng[4 * p]) && std::isfinite(ng[4 * p + 1]) && std::isfinite(ng[4 * p + 2]) &&
                   std::isfinite(ng[4 * p + 3]);
        };
        for (int32_t p = 0; p < d.n_narrow; ++p)
            if (finite(p))
                for (int q = 0; q < 3; ++q) {
                    lo[q] = std::fmin(lo[q], (double)ng[4 * p + q]);
                    hi[q] = std::fmax(hi[q], (double)ng[4 * p + q]);
                }
        if (lo[0] <= hi[0]) {
            double reach = 0.;
            for (int32_t p = 0; p < d.n_narrow; ++p)
                if (finite(p)) {
                    double d2 = 0.;
                    for (int q = 0; q < 3; ++q) {
                        const double v = (double)ng[4 * p + q] - 0.5 * (lo[q] + hi[q]);
                        d2 += v * v;
                    }
                    reach = std::fmax(reach, std::sqrt(d2) + std::sqrt((double)ng[4 * p + 3]));
                }
            for (int q = 0; q < 3; ++q) d.f32.bound[q] = (float)(0.5 * (lo[q] + hi[q]));
            d.f32.bound[3] = (float)(1.01 * reach + 1e-6 * (std::fabs(d.f32.bound[0]) + std::fabs(d.f32.bound[1]) + std::fabs(d.f32.bound[2])) + 1e-30);
        }
    }
    constexpr size_t kCounterStride = 256; // counters of calls in flight at once do not share a cache line
    static_assert(sizeof(DeviceCounters) <= kCounterStride, "counters outgrew their stride");
    hipError_t e = hipMalloc((void**)&s->d_counter_block, kCallSlots * kCounterStride);
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_counter_block, kCallSlots * sizeof(DeviceCounters), hipHostMallocDefault);
    for (int q = 0; q < kCallSlots && e == hipSuccess; ++q) {
        s->slots[q].own_counters = (DeviceCounters*)((char*)s->d_counter_block + q * kCounterStride);
        s->slots[q].h_counters = s->h_counter_block + q;
        for (auto& ev : s->slots[q].ev)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    for (TraceSpace& t : s->space)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&t.side, hipStreamNonBlocking);
    if (e != hipSuccess) {
        scene_free(s);
        return hip_fail(e, "scene workspace");
    }
    *out = s;
    return DRAY_OK;
}

// Per-device address of `scratch`, looked up on the first allocation-free call on that device.
constexpr int kMaxDevices = 64;
std::mutex g_noalloc_mu;
unsigned long long* g_noalloc_ptr[kMaxDevices] = {};

// `device` must be current.  Returns the device's copy of the workspace array, or null with g_detail set.
unsigned long long* noalloc_workspace(int device) {
    if (device < 0 || device >= kMaxDevices) {
        snprintf(g_detail, sizeof g_detail, "device %d is beyond the %d the allocation-free workspace table holds", device,
                 kMaxDevices);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_noalloc_mu);
    if (!g_noalloc_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hip_fail(e, "hipGetSymbolAddress(scratch)");
            return nullptr;
        }
        g_noalloc_ptr[device] = (unsigned long long*)p;
    }
    return g_noalloc_ptr[device];
}

// Takes the scene's next call slot (the scene's lock is held).  The slot's previous holder, kCallSlots calls ago, has
// to be over before its counters and events serve again.
int take_slot(dray_scene* scene, CallSlot** out) {
    const uint64_t ticket = ++scene->calls;
    CallSlot* slot = &scene->slots[ticket % kCallSlots];
    if (slot->ticket) HIP_TRY(hipEventSynchronize(slot->ev[2]));
    slot->ticket = ticket;
    slot->d_counters = slot->own_counters;
    slot->base = dray_stats{};
    slot->samples_from_frame = false;
    slot->t_begin = std::chrono::steady_clock::now();
    *out = slot;
    return DRAY_OK;
}

// Waits for the call that holds `ticket` and fills in its statistics.
int collect_call(dray_scene* scene, uint64_t ticket, dray_stats* stats) {
    if (!scene || !stats || ticket == 0) return DRAY_ERR_INVALID_ARG;
    DeviceGuard g(scene->device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    std::lock_guard<std::mutex> lock(scene->mu);
    CallSlot* slot = &scene->slots[ticket % kCallSlots];
    if (slot->ticket != ticket) {
        snprintf(g_detail, sizeof g_detail, "call %llu of this scene is not among its last %d", (unsigned long long)ticket, kCallSlots);
        return DRAY_ERR_INVALID_ARG;
    }
    HIP_TRY(hipEventSynchronize(slot->ev[2]));
    // (the call's last kernel has put the counters into host memory: no copy, which on the default stream would wait
    // for every render enqueued there since, and no stream of the scene's own for it -- a process has few hardware queues)
    const DeviceCounters hc = *slot->h_counters;
    float k_ms = 0.f, r_ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&k_ms, slot->ev[0], slot->ev[1]));
    HIP_TRY(hipEventElapsedTime(&r_ms, slot->ev[1], slot->ev[2]));
    *stats = slot->base;
    stats->segments = hc.segments;
    stats->walk_steps = hc.walk_steps;
    if (!slot->samples_from_frame) stats->samples = hc.samples;
    stats->kernel_ms = k_ms;
    stats->resolve_ms = r_ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - slot->t_begin).count();
    return DRAY_OK;
}

// One render on one scene, in the steps its callers need apart: check the arguments (prepare), start the trace
// kernel, which reads only the scene (begin), then -- once the running sums it adds to are in place -- the kernel that
// reads and writes the caller's buffers (finish), and, after the work is done, the statistics (collect).  The job holds
// the scene's lock from begin to finish.
struct RenderJob {
    const dray_camera* cam = nullptr;
    dray_scene* scene = nullptr;
    dray_opts opts{};
    bool noalloc = false, empty = false;
    dray_camera_derived dv{};
    Partition part{};
    int32_t W = 0, H = 0;
    hipStream_t stream = nullptr;
    DeviceCounters* d_counters = nullptr;
    unsigned long long* d_accum = nullptr;
    CallSlot* slot = nullptr;
    TraceSpace* space = nullptr;
    uint64_t ticket = 0;
    dray_stats base{};
    F32Plan plan{};
    std::unique_lock<std::mutex> lock;
    std::chrono::steady_clock::time_point t_begin;

    // Returns a status; `empty` says that there is nothing to render (camera.rs:217: no samples, no writes).
    int prepare(const dray_camera* cam_in, dray_scene* scene_in, bool have_color, size_t buf_len, const dray_opts* opts_in,
                void* stream_v, bool noalloc_in, dray_stats* stats) {
        t_begin = std::chrono::steady_clock::now();
        cam = cam_in; scene = scene_in; noalloc = noalloc_in;
        if (!camera_ok(cam) || !scene || !have_color) return DRAY_ERR_INVALID_ARG;
        if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
        if (opts.mode > DRAY_MODE_F32_PLAIN || opts.variant < 0 || opts.variant > DRAY_SCAN_GRID) return DRAY_ERR_INVALID_ARG;
        if (opts.row_step == 0) opts.row_step = 1;
        if (opts.row_step < 0 || opts.row_begin < 0) return DRAY_ERR_INVALID_ARG;
        if (opts.device >= 0 && opts.device != scene->device) return DRAY_ERR_INVALID_ARG;
        int rc = dray_camera_initialize(cam, &dv);
        if (rc) return rc;
        W = cam->image_width; H = dv.image_height;
        part.row_begin = opts.row_begin;
        part.row_step = opts.row_step;
        part.n_rows = opts.row_begin < H ? (H - opts.row_begin + opts.row_step - 1) / opts.row_step : 0;
        part.compact = (opts.flags & DRAY_FLAG_COMPACT_ROWS) ? 1 : 0;
        const size_t need = part.compact ? (size_t)part.n_rows * (size_t)W : (size_t)W * (size_t)H;
        if (buf_len < need) return DRAY_ERR_BUFFER_TOO_SMALL;
        const bool exact_product = opts.mode != DRAY_MODE_F32 && !(opts.flags & DRAY_FLAG_ITERATIVE_PRODUCT);
        if (exact_product && cam->max_depth > kExactMaxDepth) return DRAY_ERR_UNSUPPORTED;
        base = dray_stats{};
        base.image_height = H;
        base.rows_rendered = part.n_rows;
        base.n_spheres = scene->dev.n;
        base.n_wide_spheres = opts.mode == DRAY_MODE_F32 ? scene->dev.n_wide : 0;
        if (stats) *stats = base;
        empty = cam->samples_per_pixel <= 0 || part.n_rows == 0;
        stream = (hipStream_t)stream_v;
        // what this call cannot serve is refused here, before it has waited on anything or taken a call slot
        if (noalloc && (size_t)part.n_rows * (size_t)W > kNoAllocMaxPixels) {
            snprintf(g_detail, sizeof g_detail, "%d rows x %d pixels exceed the %zu pixels of the allocation-free workspace",
                     part.n_rows, W, kNoAllocMaxPixels);
            return DRAY_ERR_TOO_LARGE;
        }
        if (opts.mode == DRAY_MODE_F32 && opts.variant == DRAY_SCAN_HYBRID &&
            (size_t)scene->dev.n_narrow_padded * 8u > kHybridLdsLimit) {
            snprintf(g_detail, sizeof g_detail, "the hybrid scan stages 8 bytes per sphere in LDS: %d spheres exceed its %zu bytes",
                     scene->dev.n_narrow_padded, kHybridLdsLimit);
            return DRAY_ERR_UNSUPPORTED;
        }
        return DRAY_OK;
    }

    // Takes a call slot, clears the workspace and, in the f32 mode, enqueues the trace kernel.
    int begin() {
        DeviceGuard g(scene->device);
        if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
        lock = std::unique_lock<std::mutex>(scene->mu); // calls take turns (see dray_scene)
        if (scene->last_done && scene->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, scene->last_done, 0));
        scene->last_stream = stream;
        int rc = take_slot(scene, &slot);
        if (rc) return rc;
        ticket = slot->ticket;
        slot->base = base;
        slot->t_begin = t_begin;
        slot->samples_from_frame = opts.mode == DRAY_MODE_F32 && cam->max_depth <= 0;
        if (slot->samples_from_frame)
            slot->base.samples = (uint64_t)part.n_rows * (uint64_t)W * (uint64_t)cam->samples_per_pixel;

        // the counters and the f32 kernel's sums: the scene's own (the sums grown when a frame needs more), or the
        // library's static array for the entry point that may not allocate
        if (noalloc) {
            unsigned long long* ws = noalloc_workspace(scene->device);
            if (!ws) return DRAY_ERR_HIP;
            slot->d_counters = (DeviceCounters*)ws;
            d_accum = ws + kNoAllocCounterWords;
        }
        d_counters = slot->d_counters;
        static_assert(sizeof(DeviceCounters) <= kNoAllocCounterWords * sizeof(unsigned long long), "counters outgrew their slot");

        if (opts.mode != DRAY_MODE_F32) {
            HIP_TRY(hipMemsetAsync(d_counters, 0, sizeof(DeviceCounters), stream));
            return DRAY_OK;
        }
        const CamDev<float> c = make_cam<float>(*cam, dv);
        plan = plan_render_f32(c, part, opts.pixels_per_pool, opts.variant, scene->dev, scene->device);
        plan.no_refill = (opts.flags & DRAY_FLAG_NO_REFILL) ? 1 : 0;
        // The trace kernel runs on the stream of one of the scene's two trace spaces (the allocation-free call: on
        // the caller's stream, over the static workspace); the caller's stream picks up when it has ended.
        hipStream_t trace_stream = stream;
        if (!noalloc) {
            space = &scene->space[scene->next_space];
            scene->next_space ^= 1;
            if (plan.accum_words > space->accum_capacity) {
                if (space->d_accum) HIP_TRY(hipFree(space->d_accum)); // waits for the device: no earlier call still reads it
                space->d_accum = nullptr;
                space->accum_capacity = 0;
                HIP_TRY(hipMalloc((void**)&space->d_accum, plan.accum_words * sizeof(unsigned long long)));
                space->accum_capacity = plan.accum_words;
            }
            d_accum = space->d_accum;
            trace_stream = space->side;
            if (space->free_after) HIP_TRY(hipStreamWaitEvent(trace_stream, space->free_after, 0)); // its last user has read the sums
        }
        HIP_TRY(hipMemsetAsync(d_counters, 0, sizeof(DeviceCounters), trace_stream));
        HIP_TRY(hipEventRecord(slot->ev[0], trace_stream));
        const RngKey key = {(uint32_t)opts.seed, (uint32_t)(opts.seed >> 32), opts.sample_offset};
        // The trace kernel stores every pixel's sums once (render_f32.hip), so the planes are not cleared for it.
        if (cam->max_depth > 0)
            HIP_TRY(launch_render_f32(c, scene->dev, part, key, plan, d_accum, d_counters, trace_stream));
        else // camera.rs:265: with no depth every sample is black and no ray is traced
            HIP_TRY(hipMemsetAsync(d_accum, 0, plan.accum_words * sizeof(unsigned long long), trace_stream));
        HIP_TRY(hipEventRecord(slot->ev[1], trace_stream));
        if (trace_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, slot->ev[1], 0));
        return DRAY_OK;
    }

    // Enqueues the kernel that reads and writes the caller's buffers (after `inputs_ready`, if given: an event recorded
    // on another stream once the buffers hold what the render adds to), then lets go of the scene.
    int finish(uint32_t* d_color, double* d_raw, int32_t* d_count, hipEvent_t inputs_ready) {
        DeviceGuard g(scene->device);
        if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
        const RngKey key = {(uint32_t)opts.seed, (uint32_t)(opts.seed >> 32), opts.sample_offset};
        if (inputs_ready) HIP_TRY(hipStreamWaitEvent(stream, inputs_ready, 0));
        if (opts.mode == DRAY_MODE_F32) {
            HIP_TRY(launch_resolve_f32(make_cam<float>(*cam, dv), part, d_accum, d_color, d_raw, d_count, stream));
            HIP_TRY(counters_to_host(slot, stream));
            HIP_TRY(hipEventRecord(slot->ev[2], stream));
            if (space) space->free_after = slot->ev[2];
        } else {
            HIP_TRY(hipEventRecord(slot->ev[0], stream));
            if (opts.mode == DRAY_MODE_F64) {
                const CamDev<double> c = make_cam<double>(*cam, dv);
                // Three organisations, one result: every sphere of the list per ray, one thread per pixel (scalar); the
                // grid walk, one thread per pixel (hybrid); the grid walk as a stream of samples (auto, grid), which
                // keeps finished samples in rings of its own -- so the entry point that may not allocate runs by pixel
                int how = opts.variant == DRAY_SCAN_SCALAR ? kRefWholeList
                          : (opts.variant == DRAY_SCAN_HYBRID || noalloc) ? kRefGridByPixel : kRefGridAsStream;
                F64StreamPlan sp{};
                if (how == kRefGridAsStream && !plan_render_f64_stream(c, part, scene->dev, scene->device, &sp)) how = kRefGridByPixel;
                if (how == kRefGridAsStream && sp.ring_bytes > scene->rings_bytes) {
                    if (scene->d_rings) HIP_TRY(hipFree(scene->d_rings)); // waits for the device: no earlier call still uses it
                    scene->d_rings = nullptr;
                    scene->rings_bytes = 0;
                    if (hipMalloc(&scene->d_rings, sp.ring_bytes) == hipSuccess) {
                        scene->rings_bytes = sp.ring_bytes;
                    } else { // no room for the rings: the kernel that needs no workspace renders the same values
                        (void)hipGetLastError();
                        scene->d_rings = nullptr;
                        how = kRefGridByPixel;
                    }
                }
                if (how == kRefGridAsStream) {
                    HIP_TRY(launch_render_f64_stream(c, scene->dev, part, key, opts.flags, sp, d_color, d_raw, d_count,
                                                     d_counters, scene->d_rings, stream));
                } else {
                    HIP_TRY(launch_render_ref_f64(c, scene->dev, part, key, opts.flags, how, d_color, d_raw, d_count,
                                                  d_counters, stream));
                }
            }
            else
                HIP_TRY(launch_render_ref_f32(make_cam<float>(*cam, dv), scene->dev, part, key, opts.flags, d_color, d_raw,
                                              d_count, d_counters, stream));
            HIP_TRY(hipEventRecord(slot->ev[1], stream));
            HIP_TRY(counters_to_host(slot, stream));
            HIP_TRY(hipEventRecord(slot->ev[2], stream));
        }
        scene->last_done = slot->ev[2];
        lock.unlock();
        return DRAY_OK;
    }

    void abandon() { // a step failed: let go of the scene
        if (lock.owns_lock()) lock.unlock();
    }

    // Waits for the job and fills in what the device counted.
    int collect(dray_stats* stats) { return collect_call(scene, ticket, stats); }
};

int render_device_impl(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw, int32_t* d_count,
                       size_t buf_len, const dray_opts* opts_in, void* stream_v, dray_stats* stats,
                       bool noalloc = false, uint64_t* ticket = nullptr) {
    if (ticket) *ticket = 0;
    RenderJob job;
    int rc = job.prepare(cam, scene, d_color != nullptr, buf_len, opts_in, stream_v, noalloc, stats);
    if (rc || job.empty) return rc;
    if ((rc = job.begin()) || (rc = job.finish(d_color, d_raw, d_count, nullptr))) {
        job.abandon();
        return rc;
    }
    if (ticket) *ticket = job.ticket;
    return stats ? job.collect(stats) : DRAY_OK;
}

int render_host_impl(const dray_camera* cam, const dray_sphere* spheres, size_t n, uint32_t* color, double* raw,
                     int32_t* count, size_t buf_len, const dray_opts* opts_in, dray_stats* stats) {
    const auto t_begin = std::chrono::steady_clock::now();
    if (!camera_ok(cam) || !color || (!spheres && n)) return DRAY_ERR_INVALID_ARG;
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    if (opts.row_step == 0) opts.row_step = 1;
    if (opts.row_step < 0 || opts.row_begin < 0) return DRAY_ERR_INVALID_ARG;
    dray_camera_derived dv;
    int rc = dray_camera_initialize(cam, &dv);
    if (rc) return rc;
    const size_t W = (size_t)cam->image_width, H = (size_t)dv.image_height;
    const bool compact = (opts.flags & DRAY_FLAG_COMPACT_ROWS) != 0;
    const size_t n_rows = (size_t)opts.row_begin < H ? (H - opts.row_begin + opts.row_step - 1) / opts.row_step : 0;
    const size_t used = compact ? n_rows * W : W * H; // entries past this are never touched (main.rs:69-73)
    if (buf_len < used) return DRAY_ERR_BUFFER_TOO_SMALL;

    dray_scene* scene = nullptr;
    rc = scene_create_impl(spheres, n, opts.device, &scene);
    if (rc) return rc;
    opts.device = scene->device;
    DeviceGuard g(scene->device);
    uint32_t* d_color = nullptr;
    double* d_raw = nullptr;
    int32_t* d_count = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_color); (void)hipFree(d_raw); (void)hipFree(d_count);
        scene_free(scene);
    };
    auto fail = [&](hipError_t e, const char* what) { int r = hip_fail(e, what); cleanup(); return r; };
    const size_t alloc = used ? used : 1;
    hipError_t e;
    if ((e = hipMalloc((void**)&d_color, alloc * sizeof(uint32_t))) != hipSuccess) return fail(e, "hipMalloc color");
    if ((e = hipMemcpy(d_color, color, used * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload color");
    if (raw) {
        if ((e = hipMalloc((void**)&d_raw, alloc * 3 * sizeof(double))) != hipSuccess) return fail(e, "hipMalloc raw");
        if ((e = hipMemcpy(d_raw, raw, used * 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload raw");
    }
    if (count) {
        if ((e = hipMalloc((void**)&d_count, alloc * sizeof(int32_t))) != hipSuccess) return fail(e, "hipMalloc count");
        if ((e = hipMemcpy(d_count, count, used * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload count");
    }
    dray_stats local{};
    rc = render_device_impl(cam, scene, d_color, d_raw, d_count, used, &opts, nullptr, &local);
    if (rc == DRAY_OK) {
        if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "hipDeviceSynchronize");
        if ((e = hipMemcpy(color, d_color, used * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download color");
        if (raw && (e = hipMemcpy(raw, d_raw, used * 3 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download raw");
        if (count && (e = hipMemcpy(count, d_count, used * sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download count");
    }
    cleanup();
    if (stats) {
        *stats = local;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return rc;
}

int resolve_device(const dray_opts& opts, int* device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return DRAY_ERR_NO_DEVICE;
    *device = opts.device;
    if (*device < 0) HIP_TRY(hipGetDevice(device));
    if (*device >= ndev) return DRAY_ERR_INVALID_ARG;
    return DRAY_OK;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// A render context: everything the one-call route (host buffers in, host buffers out) would otherwise set up and tear
// down per call, kept between calls -- per share of the frame a scene (sphere tables, grid, workspace) on its device,
// the device-side colour / sum / count slabs, a stream and a pinned staging block.  Share k of G renders scanlines
// k, k + G, ... (whole scene replicated), as dray_render_multi describes.
struct CtxShare {
    int32_t device = 0;
    dray_scene* scene = nullptr;
    // One stream of its own: a GPU serves four hardware queues per process here, and with a second one in the context
    // the scene's two trace streams came to share a queue, so that a pass could not begin while the one before it ran
    // out (the headline frame in ten passes: 32.9 against 29.5 ms).  The sums go up on the scene's idle trace stream.
    hipStream_t compute = nullptr;
    hipEvent_t uploaded = nullptr;
    uint32_t* d_color = nullptr;
    double* d_raw = nullptr;
    int32_t* d_count = nullptr;
    size_t cap = 0; // pixels the three slabs hold
    char* stage = nullptr; // pinned host block: colour slab, sums slab, count slab
    size_t stage_bytes = 0;
    RenderJob job;
    dray_stats stats{};
    bool active = false;
    // a progressive render keeps three passes under way, each on a colour route of its own: route 0 is the slab and
    // the pinned block above, routes 1 and 2 have theirs here
    uint32_t* d_color_more[2] = {nullptr, nullptr};
    size_t cap_more[2] = {0, 0};
    char* stage_more[2] = {nullptr, nullptr};
    size_t stage_more_bytes[2] = {0, 0};
    hipEvent_t pass_done[3] = {nullptr, nullptr, nullptr}; // the pass's results have left the device
};
constexpr int kCtxRoutes = 3;

// One pass of a context's render between its two halves: enqueued, not yet handed to the caller.
struct CtxPass {
    std::vector<uint64_t> ticket; // per share: the scene's call (0: the share has no rows)
    int buf = 0;                   // which colour route it uses
    size_t W = 0, H = 0;
    bool staged_color = false, staged_raw = false, staged_count = false; // rows wait in pinned blocks for the scatter
};

struct dray_context {
    std::mutex mu; // one call at a time
    std::vector<CtxShare> shares;
    std::vector<dray_sphere> spheres; // the list the shares' scenes were built from
    bool have_scene = false;
    struct Pin { char* p; size_t bytes; bool owned; };
    // host ranges the copy engines reach directly: registered with the HIP runtime (dray_context_pin) or allocated
    // pinned by the context (dray_context_host_alloc, `owned`)
    std::vector<Pin> pins;
    int32_t res_W = 0, res_H = 0; // frame whose running sums the device slabs hold (0: none)
    bool res_raw = false, res_count = false;
};

namespace {

void context_free(dray_context* c) {
    if (!c) return;
    for (CtxShare& sh : c->shares) {
        DeviceGuard g(sh.device);
        if (sh.compute) (void)hipStreamSynchronize(sh.compute);
        (void)hipFree(sh.d_color); (void)hipFree(sh.d_raw); (void)hipFree(sh.d_count);
        if (sh.stage) (void)hipHostFree(sh.stage);
        for (auto& d : sh.d_color_more) (void)hipFree(d);
        for (auto& h : sh.stage_more)
            if (h) (void)hipHostFree(h);
        for (auto& e : sh.pass_done)
            if (e) (void)hipEventDestroy(e);
        if (sh.uploaded) (void)hipEventDestroy(sh.uploaded);
        if (sh.compute) (void)hipStreamDestroy(sh.compute);
        scene_free(sh.scene);
    }
    for (auto& pin : c->pins) {
        if (pin.owned) (void)hipHostFree(pin.p);
        else (void)hipHostUnregister(pin.p);
    }
    delete c;
}

int context_create_impl(const int32_t* devices, int32_t n_devices, dray_context** out) {
    if (!out) return DRAY_ERR_INVALID_ARG;
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return DRAY_ERR_NO_DEVICE;
    if (n_devices <= 0) { // every visible GPU
        n_devices = visible;
        devices = nullptr;
    }
    if (n_devices > 64) return DRAY_ERR_INVALID_ARG;
    dray_context* c = new (std::nothrow) dray_context();
    if (!c) return DRAY_ERR_INTERNAL;
    c->shares.resize((size_t)n_devices);
    for (int32_t k = 0; k < n_devices; ++k) {
        const int32_t d = devices ? devices[k] : k;
        if (d < 0 || d >= visible) {
            context_free(c);
            return DRAY_ERR_INVALID_ARG;
        }
        CtxShare& sh = c->shares[(size_t)k];
        sh.device = d;
        DeviceGuard g(d);
        hipError_t e = g.err;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sh.compute, hipStreamNonBlocking);
        for (auto& ev : sh.pass_done)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.uploaded, hipEventDisableTiming);
        if (e != hipSuccess) {
            const int rc = hip_fail(e, "context streams");
            context_free(c);
            return rc;
        }
    }
    *out = c;
    return DRAY_OK;
}

// Builds the shares' scenes from the list, unless they were built from this very list already.
int context_set_scene_locked(dray_context* c, const dray_sphere* spheres, size_t n) {
    if (!spheres && n) return DRAY_ERR_INVALID_ARG;
    if (c->have_scene && c->spheres.size() == n && (n == 0 || std::memcmp(c->spheres.data(), spheres, n * sizeof *spheres) == 0))
        return DRAY_OK;
    for (size_t k = 0; k < n; ++k)
        if (!entry_ok(spheres[k])) return DRAY_ERR_UNSUPPORTED;
    c->have_scene = false;
    for (CtxShare& sh : c->shares) {
        if (sh.compute) {
            DeviceGuard g(sh.device);
            (void)hipStreamSynchronize(sh.compute);
        }
        scene_free(sh.scene);
        sh.scene = nullptr;
    }
    for (CtxShare& sh : c->shares) {
        const int rc = scene_create_impl(spheres, n, sh.device, &sh.scene);
        if (rc) return rc;
    }
    c->spheres.assign(spheres, spheres + n);
    c->have_scene = true;
    c->res_W = c->res_H = 0; // the slabs' sums belong to renders of the previous list: not to be carried over silently
    return DRAY_OK;
}

// Is [p, p + bytes) covered by the registered ranges (page-aligned, disjoint, possibly adjacent)?
bool pinned(const dray_context* c, const void* p, size_t bytes) {
    const char* at = (const char*)p;
    const char* end = at + bytes;
    for (bool moved = true; at < end && moved;) {
        moved = false;
        for (const auto& pin : c->pins)
            if (at >= pin.p && at < pin.p + pin.bytes) {
                at = pin.p + pin.bytes;
                moved = true;
                break;
            }
    }
    return at >= end;
}

// Registers the pages of [host, host + bytes) that no earlier registration holds (two buffers of the caller may meet
// inside one page, and the runtime refuses to register a page twice).
int context_pin_locked(dray_context* c, void* host, size_t bytes) {
    const uintptr_t page = 4096;
    const uintptr_t lo = (uintptr_t)host & ~(page - 1), hi = ((uintptr_t)host + bytes + page - 1) & ~(page - 1);
    std::vector<std::pair<uintptr_t, uintptr_t>> todo{{lo, hi}};
    for (const auto& pin : c->pins) {
        const uintptr_t a = (uintptr_t)pin.p, b = a + pin.bytes;
        std::vector<std::pair<uintptr_t, uintptr_t>> rest;
        for (const auto& seg : todo) {
            if (b <= seg.first || a >= seg.second) { rest.push_back(seg); continue; }
            if (seg.first < a) rest.emplace_back(seg.first, a);
            if (b < seg.second) rest.emplace_back(b, seg.second);
        }
        todo.swap(rest);
    }
    for (const auto& seg : todo) {
        const hipError_t e = hipHostRegister((void*)seg.first, seg.second - seg.first, hipHostRegisterDefault);
        if (e != hipSuccess) return hip_fail(e, "hipHostRegister");
        c->pins.push_back({(char*)seg.first, (size_t)(seg.second - seg.first), false});
    }
    return DRAY_OK;
}

// image row k + r G of the caller's buffer <-> slab row r of share k; `elem` bytes per pixel
size_t share_rows(size_t H, int32_t k, int32_t G) { return (size_t)k < H ? (H - (size_t)k + (size_t)G - 1) / (size_t)G : 0; }
void scatter_rows(void* host, const char* stage, size_t elem, size_t W, size_t H, int32_t k, int32_t G) {
    char* h = (char*)host + (size_t)k * W * elem;
    const size_t row = W * elem, host_pitch = (size_t)G * row, n_rows = share_rows(H, k, G);
    for (size_t r = 0; r < n_rows; ++r) std::memcpy(h + r * host_pitch, stage + r * row, row);
}

#define CTX_TRY(expr)                                      \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return hip_fail(e__, #expr); \
    } while (0)

// First half of a context's render: everything is enqueued -- trace kernels, the sums' way up, the kernels that read
// them, the results' way down -- and `pass` says what is under way.  `buf` picks the colour route (1, 2: a slab and
// a pinned block of their own, for a pass enqueued while earlier ones have not been handed out; their colours are
// always staged).  `own_sums`: 0 none; a pass of several whose caller gave no raw / count keeps those sums in the context's
// slabs all the same, so that the picture is the picture of all passes so far -- 1: zeroed for the first pass, 2: carried on.
// Nothing here waits for the device.
int context_enqueue_locked(dray_context* c, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                           size_t buf_len, const dray_opts* opts_in, uint32_t sync, int buf, bool stage_color, int own_sums,
                           CtxPass* pass) {
    pass->ticket.assign(c->shares.size(), 0);
    pass->buf = buf;
    if (!camera_ok(cam) || !color) return DRAY_ERR_INVALID_ARG;
    if (!c->have_scene) {
        snprintf(g_detail, sizeof g_detail, "the context has no scene: call dray_context_set_scene first");
        return DRAY_ERR_INVALID_ARG;
    }
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    // the row partition is this call's to make
    if (opts.row_begin != 0 || opts.row_step > 1 || (opts.flags & DRAY_FLAG_COMPACT_ROWS)) return DRAY_ERR_INVALID_ARG;
    if (opts.mode > DRAY_MODE_F32_PLAIN || opts.variant < 0 || opts.variant > DRAY_SCAN_GRID) return DRAY_ERR_INVALID_ARG;
    if (sync & ~(uint32_t)(DRAY_SYNC_UPLOAD_SUMS | DRAY_SYNC_DOWNLOAD_SUMS)) return DRAY_ERR_INVALID_ARG;
    dray_camera_derived dv;
    int rc = dray_camera_initialize(cam, &dv);
    if (rc) return rc;
    const size_t W = (size_t)cam->image_width, H = (size_t)dv.image_height;
    pass->W = W;
    pass->H = H;
    if (buf_len < W * H) return DRAY_ERR_BUFFER_TOO_SMALL;
    const bool exact_product = opts.mode != DRAY_MODE_F32 && !(opts.flags & DRAY_FLAG_ITERATIVE_PRODUCT);
    if (exact_product && cam->max_depth > kExactMaxDepth) return DRAY_ERR_UNSUPPORTED;
    const int32_t G = (int32_t)c->shares.size();
    if (cam->samples_per_pixel <= 0) return DRAY_OK; // camera.rs:217: no samples, no writes
    const bool upload = (sync & DRAY_SYNC_UPLOAD_SUMS) != 0, download = (sync & DRAY_SYNC_DOWNLOAD_SUMS) != 0;
    if (!upload && (raw || count)) {
        // the sums stay where the previous call left them, on the device: it must have been this frame's call
        if (c->res_W != (int32_t)W || c->res_H != (int32_t)H || c->res_raw != (raw != nullptr) || c->res_count != (count != nullptr)) {
            snprintf(g_detail, sizeof g_detail, "no running sums of a %zu x %zu frame are resident: pass DRAY_SYNC_UPLOAD_SUMS", W, H);
            return DRAY_ERR_INVALID_ARG;
        }
    }
    c->res_W = c->res_H = 0; // until this call has gone through

    const bool direct_color = !stage_color && G > 0 && pinned(c, color, W * H * sizeof(uint32_t));
    const bool direct_raw = raw && pinned(c, raw, W * H * 3 * sizeof(double));
    const bool direct_count = count && pinned(c, count, W * H * sizeof(int32_t));
    pass->staged_color = !direct_color;
    pass->staged_raw = download && raw && !direct_raw;
    pass->staged_count = download && count && !direct_count;

    struct Fail { // on any failure: let go of the scenes, drain what was enqueued
        dray_context* c;
        bool armed = true;
        ~Fail() {
            if (!armed) return;
            for (CtxShare& sh : c->shares) {
                if (!sh.active) continue;
                sh.job.abandon();
                DeviceGuard g(sh.device);
                (void)hipStreamSynchronize(sh.compute);
                sh.active = false;
            }
        }
    } guard{c};

    // ---- set-up per share: slabs and staging large enough, the job checked ----------------------------------
    for (int32_t k = 0; k < G; ++k) {
        CtxShare& sh = c->shares[(size_t)k];
        sh.active = false;
        std::memset(&sh.stats, 0, sizeof sh.stats);
        const size_t n_rows = share_rows(H, k, G);
        if (n_rows == 0) continue;
        const size_t slab = n_rows * W;
        DeviceGuard g(sh.device);
        if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
        if (slab > sh.cap) {
            CTX_TRY(hipStreamSynchronize(sh.compute));
            (void)hipFree(sh.d_color); (void)hipFree(sh.d_raw); (void)hipFree(sh.d_count);
            sh.d_color = nullptr; sh.d_raw = nullptr; sh.d_count = nullptr; sh.cap = 0;
            CTX_TRY(hipMalloc((void**)&sh.d_color, slab * sizeof(uint32_t)));
            CTX_TRY(hipMalloc((void**)&sh.d_raw, slab * 3 * sizeof(double)));
            CTX_TRY(hipMalloc((void**)&sh.d_count, slab * sizeof(int32_t)));
            sh.cap = slab;
        }
        const size_t stage_need = slab * (sizeof(uint32_t) + 3 * sizeof(double) + sizeof(int32_t));
        if (!(direct_color && (!raw || direct_raw) && (!count || direct_count)) && stage_need > sh.stage_bytes) {
            CTX_TRY(hipStreamSynchronize(sh.compute));
            if (sh.stage) (void)hipHostFree(sh.stage);
            sh.stage = nullptr; sh.stage_bytes = 0;
            CTX_TRY(hipHostMalloc((void**)&sh.stage, stage_need, hipHostMallocDefault));
            sh.stage_bytes = stage_need;
        }
        if (buf > 0) { // a colour route of its own
            const int r = buf - 1;
            if (slab > sh.cap_more[r]) {
                CTX_TRY(hipStreamSynchronize(sh.compute));
                (void)hipFree(sh.d_color_more[r]);
                sh.d_color_more[r] = nullptr; sh.cap_more[r] = 0;
                CTX_TRY(hipMalloc((void**)&sh.d_color_more[r], slab * sizeof(uint32_t)));
                sh.cap_more[r] = slab;
            }
            if (slab * sizeof(uint32_t) > sh.stage_more_bytes[r]) {
                CTX_TRY(hipStreamSynchronize(sh.compute));
                if (sh.stage_more[r]) (void)hipHostFree(sh.stage_more[r]);
                sh.stage_more[r] = nullptr; sh.stage_more_bytes[r] = 0;
                CTX_TRY(hipHostMalloc((void**)&sh.stage_more[r], slab * sizeof(uint32_t), hipHostMallocDefault));
                sh.stage_more_bytes[r] = slab * sizeof(uint32_t);
            }
        }
        dray_opts o = opts;
        o.device = sh.device;
        o.row_begin = k;
        o.row_step = G;
        o.flags |= DRAY_FLAG_COMPACT_ROWS;
        rc = sh.job.prepare(cam, sh.scene, true, slab, &o, sh.compute, false, &sh.stats);
        if (rc) return rc;
        sh.active = !sh.job.empty;
    }
    // ---- every share's trace kernel starts before any host-side copying ----------------------------------------
    for (CtxShare& sh : c->shares)
        if (sh.active && (rc = sh.job.begin())) return rc;

    auto rows = [&](CtxShare& sh, int32_t k, void* dev, void* host, size_t elem, bool to_device, bool direct, char* stage,
                    hipStream_t stream) -> hipError_t {
        const size_t n_rows = share_rows(H, k, G);
        char* h = (char*)host + (size_t)k * W * elem;
        const size_t row = W * elem, host_pitch = (size_t)G * row;
        const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        if (direct) { // registered memory: the copy engine reads or writes the caller's rows itself
            if (G == 1) return to_device ? hipMemcpyAsync(dev, h, n_rows * row, kind, stream) : hipMemcpyAsync(h, dev, n_rows * row, kind, stream);
            return to_device ? hipMemcpy2DAsync(dev, row, h, host_pitch, row, n_rows, kind, stream)
                             : hipMemcpy2DAsync(h, host_pitch, dev, row, row, n_rows, kind, stream);
        }
        if (to_device) { // gather the rows into the pinned block, one copy from there
            for (size_t r = 0; r < n_rows; ++r) std::memcpy(stage + r * row, h + r * host_pitch, row);
            return hipMemcpyAsync(dev, stage, n_rows * row, kind, stream);
        }
        return hipMemcpyAsync(stage, dev, n_rows * row, kind, stream); // scattered to the rows once the pass is done
    };
    // ---- the sums the render adds to go up while the trace kernels run; then the kernels that need them ---------
    for (int32_t k = 0; k < G; ++k) {
        CtxShare& sh = c->shares[(size_t)k];
        if (!sh.active) continue;
        DeviceGuard g(sh.device);
        const size_t slab = (size_t)sh.job.part.n_rows * W;
        char* st_raw = sh.stage ? sh.stage + slab * sizeof(uint32_t) : nullptr;
        char* st_count = sh.stage ? st_raw + slab * 3 * sizeof(double) : nullptr;
        uint32_t* d_color = buf > 0 ? sh.d_color_more[buf - 1] : sh.d_color;
        char* st_color = buf > 0 ? sh.stage_more[buf - 1] : sh.stage;
        // The way up: on the stream of the scene's other trace space, which has nothing to do until the next call's
        // trace kernel (a copy enqueued behind this call's own kernels would be made by a kernel too, and wait with
        // them for the compute units the trace kernel holds).
        hipStream_t up = sh.scene->space[sh.scene->next_space].side;
        bool sent = false;
        if (upload && raw) { CTX_TRY(rows(sh, k, sh.d_raw, raw, 3 * sizeof(double), true, direct_raw, st_raw, up)); sent = true; }
        if (upload && count) { CTX_TRY(rows(sh, k, sh.d_count, count, sizeof(int32_t), true, direct_count, st_count, up)); sent = true; }
        if (sent) CTX_TRY(hipEventRecord(sh.uploaded, up));
        const bool own_raw = own_sums != 0 && !raw, own_count = own_sums != 0 && !count;
        if (own_sums == 1 && own_raw) CTX_TRY(hipMemsetAsync(sh.d_raw, 0, slab * 3 * sizeof(double), sh.compute));
        if (own_sums == 1 && own_count) CTX_TRY(hipMemsetAsync(sh.d_count, 0, slab * sizeof(int32_t), sh.compute));
        if ((rc = sh.job.finish(d_color, raw || own_raw ? sh.d_raw : nullptr, count || own_count ? sh.d_count : nullptr,
                                sent ? sh.uploaded : nullptr)))
            return rc;
        pass->ticket[(size_t)k] = sh.job.ticket;
        // results come back on the compute stream, behind the kernels
        CTX_TRY(rows(sh, k, d_color, color, sizeof(uint32_t), false, direct_color, st_color, sh.compute));
        if (download && raw) CTX_TRY(rows(sh, k, sh.d_raw, raw, 3 * sizeof(double), false, direct_raw, st_raw, sh.compute));
        if (download && count) CTX_TRY(rows(sh, k, sh.d_count, count, sizeof(int32_t), false, direct_count, st_count, sh.compute));
        CTX_TRY(hipEventRecord(sh.pass_done[buf], sh.compute));
        sh.active = false;
    }
    guard.armed = false;
    c->res_W = (int32_t)W; c->res_H = (int32_t)H; c->res_raw = raw != nullptr; c->res_count = count != nullptr;
    return DRAY_OK;
}

// Second half: waits for the pass, hands the staged rows out, adds the shares' statistics up.
int context_complete_locked(dray_context* c, const CtxPass& pass, uint32_t* color, double* raw, int32_t* count, dray_stats* stats) {
    const int32_t G = (int32_t)c->shares.size();
    const size_t W = pass.W, H = pass.H;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->image_height = (int32_t)H;
        stats->n_spheres = (int32_t)c->spheres.size();
    }
    for (int32_t k = 0; k < G; ++k) {
        CtxShare& sh = c->shares[(size_t)k];
        if ((size_t)k >= pass.ticket.size() || pass.ticket[(size_t)k] == 0) continue;
        DeviceGuard g(sh.device);
        const hipError_t e = hipEventSynchronize(sh.pass_done[pass.buf]);
        if (e != hipSuccess) {
            c->res_W = c->res_H = 0;
            return hip_fail(e, "hipEventSynchronize");
        }
        const size_t slab = share_rows(H, k, G) * W;
        if (pass.staged_color) scatter_rows(color, pass.buf > 0 ? sh.stage_more[pass.buf - 1] : sh.stage, sizeof(uint32_t), W, H, k, G);
        if (pass.staged_raw) scatter_rows(raw, sh.stage + slab * sizeof(uint32_t), 3 * sizeof(double), W, H, k, G);
        if (pass.staged_count) scatter_rows(count, sh.stage + slab * (sizeof(uint32_t) + 3 * sizeof(double)), sizeof(int32_t), W, H, k, G);
        dray_stats st{};
        const int rc = collect_call(sh.scene, pass.ticket[(size_t)k], &st);
        if (rc) return rc;
        if (stats) {
            stats->segments += st.segments; stats->samples += st.samples; stats->walk_steps += st.walk_steps;
            stats->rows_rendered += st.rows_rendered;
            stats->kernel_ms = std::fmax(stats->kernel_ms, st.kernel_ms);   // the shares run side by side
            stats->resolve_ms = std::fmax(stats->resolve_ms, st.resolve_ms);
            if (st.n_wide_spheres > stats->n_wide_spheres) stats->n_wide_spheres = st.n_wide_spheres;
        }
    }
    return DRAY_OK;
}
#undef CTX_TRY

// The whole frame into host buffers: see dray_context_render in dray.h.
int context_render_locked(dray_context* c, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                          size_t buf_len, const dray_opts* opts_in, uint32_t sync, dray_stats* stats) {
    const auto t_begin = std::chrono::steady_clock::now();
    CtxPass pass;
    int rc = context_enqueue_locked(c, cam, color, raw, count, buf_len, opts_in, sync, 0, false, 0, &pass);
    if (rc == DRAY_OK) rc = context_complete_locked(c, pass, color, raw, count, stats);
    else if (stats) std::memset(stats, 0, sizeof *stats);
    if (rc == DRAY_OK && stats)
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

// A frame's samples in passes, two passes under way at a time: see dray_context_render_progressive in dray.h.
int context_render_progressive_locked(dray_context* c, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                                      size_t buf_len, const dray_opts* opts_in, int32_t samples_per_pass,
                                      dray_present_fn present, void* user, dray_stats* stats) {
    const auto t_begin = std::chrono::steady_clock::now();
    if (!camera_ok(cam) || samples_per_pass <= 0) return DRAY_ERR_INVALID_ARG;
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    const int32_t total = cam->samples_per_pixel;
    const int32_t n_pass = total <= 0 ? 1 : (total + samples_per_pass - 1) / samples_per_pass;
    dray_stats sum{};
    CtxPass under_way[kCtxRoutes];
    dray_camera pass_cam[kCtxRoutes];
    auto enqueue = [&](int32_t p) {
        const int r = p % kCtxRoutes;
        pass_cam[r] = *cam;
        const int32_t first = p * samples_per_pass;
        pass_cam[r].samples_per_pixel = total <= 0 ? total : (total - first < samples_per_pass ? total - first : samples_per_pass);
        dray_opts o = opts;
        o.sample_offset = opts.sample_offset + (uint32_t)first;
        const uint32_t sync = (p == 0 ? DRAY_SYNC_UPLOAD_SUMS : 0u) | (p == n_pass - 1 ? DRAY_SYNC_DOWNLOAD_SUMS : 0u);
        return context_enqueue_locked(c, &pass_cam[r], color, raw, count, buf_len, &o, sync, r, n_pass > 1,
                                      n_pass > 1 ? (p == 0 ? 1 : 2) : 0, &under_way[r]);
    };
    // Three passes are under way at a time: while pass p's picture is handed out, pass p + 1 is on the GPU (its trace
    // kernel holds the compute units, and pass p's small last kernel completes as that one runs out) and pass p + 2 is
    // enqueued behind it, so that the GPU does not wait for this thread between passes.
    int rc = DRAY_OK;
    for (int32_t p = 0; rc == DRAY_OK && p < n_pass && p < kCtxRoutes; ++p) rc = enqueue(p);
    for (int32_t p = 0; rc == DRAY_OK && p < n_pass; ++p) {
        dray_stats st{};
        rc = context_complete_locked(c, under_way[p % kCtxRoutes], color, raw, count, &st);
        if (rc) break;
        if (p == 0) sum = st;
        else {
            sum.segments += st.segments; sum.samples += st.samples; sum.walk_steps += st.walk_steps;
            sum.kernel_ms += st.kernel_ms; sum.resolve_ms += st.resolve_ms;
        }
        const int32_t done = (p + 1) * samples_per_pass < total ? (p + 1) * samples_per_pass : total;
        if (present && total > 0) present(user, p, done); // camera.rs:217: no samples, nothing to show
        if (p + kCtxRoutes < n_pass) rc = enqueue(p + kCtxRoutes); // its colour route is the one just handed out
    }
    if (rc != DRAY_OK) { // whatever is still under way must not outlive the buffers it writes
        for (CtxShare& sh : c->shares) {
            DeviceGuard g(sh.device);
            (void)hipStreamSynchronize(sh.compute);
        }
        c->res_W = c->res_H = 0;
        return rc;
    }
    if (stats) {
        *stats = sum;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return DRAY_OK;
}

// The contexts behind dray_render and dray_render_multi: one per list of devices, kept until dray_release_cache or
// the end of the process, so that a second call with the same sphere list finds scene, grid and buffers in place.
// A cached context is shared between the cache and the calls that use it: a call keeps the one it looked up alive for
// its own length, so a context dropped from the cache (eviction, dray_release_cache) dies when its last user lets go.
std::mutex g_cache_mu;
std::vector<std::pair<std::vector<int32_t>, std::shared_ptr<dray_context>>> g_cache;

int cached_render(const std::vector<int32_t>& devs, const dray_camera* cam, const dray_sphere* spheres, size_t n,
                  uint32_t* color, double* raw, int32_t* count, size_t buf_len, const dray_opts* opts, dray_stats* stats) {
    const auto t_begin = std::chrono::steady_clock::now();
    std::shared_ptr<dray_context> held;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        for (auto& e : g_cache)
            if (e.first == devs) held = e.second;
        if (!held) {
            if (g_cache.size() >= 8) g_cache.erase(g_cache.begin()); // a handful of device lists at most: drop the oldest
            dray_context* made = nullptr;
            const int rc = context_create_impl(devs.data(), (int32_t)devs.size(), &made);
            if (rc) return rc;
            held = std::shared_ptr<dray_context>(made, context_free);
            g_cache.emplace_back(devs, held);
        }
    }
    dray_context* c = held.get();
    std::lock_guard<std::mutex> lock(c->mu); // (released before `held`, declared earlier, lets go of the context)
    // argument errors come before any device work, as they always did
    if (!camera_ok(cam) || !color || (!spheres && n)) return DRAY_ERR_INVALID_ARG;
    int rc = context_set_scene_locked(c, spheres, n);
    if (rc) return rc;
    rc = context_render_locked(c, cam, color, raw, count, buf_len, opts, DRAY_SYNC_UPLOAD_SUMS | DRAY_SYNC_DOWNLOAD_SUMS, stats);
    if (rc == DRAY_OK && stats)
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

// Camera::render over several GPUs from one process.
int render_multi_impl(const dray_camera* cam, const dray_sphere* spheres, size_t n, uint32_t* color, double* raw,
                      int32_t* count, size_t buf_len, const dray_opts* opts_in, const int32_t* devices, int32_t n_devices,
                      dray_stats* stats) {
    if (!camera_ok(cam) || !color || (!spheres && n)) return DRAY_ERR_INVALID_ARG;
    if (opts_in) { // the row partition is this call's to make
        if (opts_in->row_begin != 0 || opts_in->row_step > 1 || (opts_in->flags & DRAY_FLAG_COMPACT_ROWS)) return DRAY_ERR_INVALID_ARG;
        if (opts_in->mode > DRAY_MODE_F32_PLAIN || opts_in->variant < 0 || opts_in->variant > DRAY_SCAN_GRID) return DRAY_ERR_INVALID_ARG;
    }
    for (size_t k = 0; k < n; ++k)
        if (!entry_ok(spheres[k])) return DRAY_ERR_UNSUPPORTED;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return DRAY_ERR_NO_DEVICE;
    std::vector<int32_t> devs;
    if (n_devices <= 0) n_devices = visible;              // every visible GPU
    if (n_devices > 64) return DRAY_ERR_INVALID_ARG;
    for (int32_t k = 0; k < n_devices; ++k) {
        const int32_t d = devices ? devices[k] : k;
        if (d < 0 || d >= visible) return DRAY_ERR_INVALID_ARG;
        devs.push_back(d);
    }
    return cached_render(devs, cam, spheres, n, color, raw, count, buf_len, opts_in, stats);
}

// Camera::render on one GPU.  A call that names a row partition of its own takes the plain route (render_host_impl).
int render_one_impl(const dray_camera* cam, const dray_sphere* spheres, size_t n, uint32_t* color, double* raw,
                    int32_t* count, size_t buf_len, const dray_opts* opts_in, dray_stats* stats) {
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    if (opts.row_begin != 0 || (opts.row_step != 0 && opts.row_step != 1) || (opts.flags & DRAY_FLAG_COMPACT_ROWS))
        return render_host_impl(cam, spheres, n, color, raw, count, buf_len, opts_in, stats);
    if (!camera_ok(cam) || !color || (!spheres && n)) return DRAY_ERR_INVALID_ARG;
    int device = -1;
    const int rc = resolve_device(opts, &device);
    if (rc) return rc;
    opts.device = -1; // the context's share names the device
    return cached_render(std::vector<int32_t>{device}, cam, spheres, n, color, raw, count, buf_len, &opts, stats);
}


int camera_rays_device_impl(const dray_camera* cam, const dray_pixel_index* d_px, size_t n, dray_path_entry* d_out,
                            const dray_opts* opts_in, void* stream_v) {
    if (!camera_ok(cam) || (n && (!d_px || !d_out))) return DRAY_ERR_INVALID_ARG;
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    dray_camera_derived dv;
    int rc = dray_camera_initialize(cam, &dv);
    if (rc) return rc;
    int device = -1;
    if ((rc = resolve_device(opts, &device))) return rc;
    DeviceGuard g(device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    const RngKey key = {(uint32_t)opts.seed, (uint32_t)(opts.seed >> 32), opts.sample_offset};
    HIP_TRY(launch_camera_rays(make_cam<double>(*cam, dv), key, d_px, n, d_out, (hipStream_t)stream_v));
    return DRAY_OK;
}

int path_step_device_impl(dray_scene* scene, int32_t image_width, dray_path_entry* d_entries, size_t n,
                          const dray_opts* opts_in, void* stream_v, bool count, uint64_t* ticket = nullptr) {
    if (!scene || image_width < 1 || (n && !d_entries)) return DRAY_ERR_INVALID_ARG;
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    if (opts.device >= 0 && opts.device != scene->device) return DRAY_ERR_INVALID_ARG;
    DeviceGuard g(scene->device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    const RngKey key = {(uint32_t)opts.seed, (uint32_t)(opts.seed >> 32), opts.sample_offset};
    hipStream_t stream = (hipStream_t)stream_v;
    // DRAY_MODE_F32: the step at the throughput kernel's precision (a list with bounding boxes takes the f64 step)
    const bool f32 = opts.mode == DRAY_MODE_F32 && scene->dev.n_boxes == 0;
    auto launch = [&](DeviceCounters* counters) {
        return f32 ? launch_path_step_f32(scene->dev, key, image_width, d_entries, n, counters, stream)
                   : launch_path_step(scene->dev, key, image_width, opts.variant != DRAY_SCAN_SCALAR, d_entries, n, counters, stream);
    };
    if (count) { // only the counting form touches the scene's workspace; the plain form reads the sphere tables alone
        std::lock_guard<std::mutex> lock(scene->mu);
        if (scene->last_done && scene->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, scene->last_done, 0));
        scene->last_stream = stream;
        CallSlot* slot = nullptr;
        const int rc = take_slot(scene, &slot);
        if (rc) return rc;
        slot->base.n_spheres = scene->dev.n;
        HIP_TRY(hipMemsetAsync(slot->d_counters, 0, sizeof(DeviceCounters), stream));
        HIP_TRY(hipEventRecord(slot->ev[0], stream));
        HIP_TRY(launch(slot->d_counters));
        HIP_TRY(hipEventRecord(slot->ev[1], stream));
        HIP_TRY(counters_to_host(slot, stream));
        HIP_TRY(hipEventRecord(slot->ev[2], stream));
        scene->last_done = slot->ev[2];
        if (ticket) *ticket = slot->ticket;
        return DRAY_OK;
    }
    HIP_TRY(launch(nullptr));
    return DRAY_OK;
}

// RAII device buffer for the host-buffer variants
template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

int camera_rays_host_impl(const dray_camera* cam, const dray_pixel_index* px, size_t n, dray_path_entry* out,
                          const dray_opts* opts_in) {
    if (!camera_ok(cam) || (n && (!px || !out))) return DRAY_ERR_INVALID_ARG;
    dray_opts opts;
    if (opts_in) opts = *opts_in; else dray_opts_default(&opts);
    int device = -1, rc = resolve_device(opts, &device);
    if (rc) return rc;
    DeviceGuard g(device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    if (n == 0) return DRAY_OK;
    DevBuf<dray_pixel_index> d_px;
    DevBuf<dray_path_entry> d_out;
    HIP_TRY(d_px.alloc(n));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(hipMemcpy(d_px.p, px, n * sizeof *px, hipMemcpyHostToDevice));
    opts.device = device;
    if ((rc = camera_rays_device_impl(cam, d_px.p, n, d_out.p, &opts, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof *out, hipMemcpyDeviceToHost));
    return DRAY_OK;
}

int path_step_host_impl(dray_scene* scene, int32_t image_width, dray_path_entry* entries, size_t n,
                        const dray_opts* opts_in, dray_stats* stats) {
    const auto t_begin = std::chrono::steady_clock::now();
    if (!scene || image_width < 1 || (n && !entries)) return DRAY_ERR_INVALID_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    DeviceGuard g(scene->device);
    if (g.err != hipSuccess) return hip_fail(g.err, "hipSetDevice");
    if (n == 0) return DRAY_OK;
    DevBuf<dray_path_entry> d_e;
    HIP_TRY(d_e.alloc(n));
    HIP_TRY(hipMemcpy(d_e.p, entries, n * sizeof *entries, hipMemcpyHostToDevice));
    uint64_t ticket = 0;
    const int rc = path_step_device_impl(scene, image_width, d_e.p, n, opts_in, nullptr, true, &ticket);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(entries, d_e.p, n * sizeof *entries, hipMemcpyDeviceToHost));
    if (stats) {
        const int rc2 = collect_call(scene, ticket, stats);
        if (rc2) return rc2;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return DRAY_OK;
}

} // namespace

// ------------------------------------------------------------------ C ABI ----

extern "C" {

const char* dray_version(void) { return "dray-hip 0.1.0 (gfx950)"; }

const char* dray_strerror(int status) {
    switch (status) {
    case DRAY_OK: return "ok";
    case DRAY_ERR_INVALID_ARG: return "invalid argument";
    case DRAY_ERR_BUFFER_TOO_SMALL: return "buffer too small for image_width x image_height";
    case DRAY_ERR_UNSUPPORTED: return "unsupported scene object, material or depth";
    case DRAY_ERR_NO_DEVICE: return "no HIP device";
    case DRAY_ERR_HIP: return g_detail[0] ? g_detail : "HIP runtime error";
    case DRAY_ERR_INTERNAL: return "internal error";
    case DRAY_ERR_TOO_LARGE: return g_detail[0] ? g_detail : "frame larger than the allocation-free workspace";
    default: return "unknown status";
    }
}

int dray_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    return n;
}

void dray_opts_default(dray_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->seed = 1;
    o->mode = DRAY_MODE_F32;
    o->device = -1;
    o->row_step = 1;
}

void dray_camera_default(dray_camera* cam) { // camera.rs:147-163
    if (!cam) return;
    std::memset(cam, 0, sizeof *cam);
    cam->aspect_ratio = 1.0;
    cam->image_width = 100;
    cam->samples_per_pixel = 10;
    cam->max_depth = 10;
    cam->vfov = 90.;
    cam->lookat[2] = -1.;
    cam->vup[1] = 1.;
    cam->defocus_angle = 0.;
    cam->focus_dist = 10.;
}

int dray_camera_initialize(const dray_camera* cam, dray_camera_derived* out) { // camera.rs:169-206
    if (!camera_ok(cam) || !out) return DRAY_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const double hh = (double)cam->image_width / cam->aspect_ratio;
    int32_t H = hh >= 2147483647.0 ? 2147483647 : (int32_t)hh; // `as i32` saturates
    if (H < 1) H = 1;
    out->image_height = H;
    out->pixel_samples_scale = 1.0 / (double)cam->samples_per_pixel;
    const D3 center = ld(cam->lookfrom);
    const double theta = degrees_to_radians(cam->vfov);
    const double h = std::tan(theta / 2.);
    const double viewport_height = 2. * h * cam->focus_dist;
    const double viewport_width = viewport_height * ((double)cam->image_width / (double)H);
    const D3 w = unit(sub(ld(cam->lookfrom), ld(cam->lookat)));
    const D3 u = unit(cross(ld(cam->vup), w));
    const D3 v = cross(w, u);
    const D3 viewport_u = scale(u, viewport_width);
    const D3 viewport_v = scale(neg(v), viewport_height);
    const D3 du = divs(viewport_u, (double)cam->image_width);
    const D3 dvv = divs(viewport_v, (double)H);
    const D3 upper_left = sub(sub(sub(center, scale(w, cam->focus_dist)), divs(viewport_u, 2.)), divs(viewport_v, 2.));
    const D3 p00 = add(upper_left, scale(add(du, dvv), 0.5));
    const double defocus_radius = cam->focus_dist * std::tan(degrees_to_radians(cam->defocus_angle / 2.));
    stv(out->center, center);
    stv(out->pixel00_loc, p00);
    stv(out->pixel_delta_u, du);
    stv(out->pixel_delta_v, dvv);
    stv(out->u, u); stv(out->v, v); stv(out->w, w);
    stv(out->defocus_disk_u, scale(u, defocus_radius));
    stv(out->defocus_disk_v, scale(v, defocus_radius));
    return DRAY_OK;
}

int dray_render(const dray_camera* cam, const dray_sphere* spheres, size_t n_spheres, uint32_t* color, double* raw,
                int32_t* count, size_t buf_len, const dray_opts* opts, dray_stats* stats) {
    try {
        return render_one_impl(cam, spheres, n_spheres, color, raw, count, buf_len, opts, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_render_multi(const dray_camera* cam, const dray_sphere* spheres, size_t n_spheres, uint32_t* color, double* raw,
                      int32_t* count, size_t buf_len, const dray_opts* opts, const int32_t* devices, int32_t n_devices,
                      dray_stats* stats) {
    try {
        return render_multi_impl(cam, spheres, n_spheres, color, raw, count, buf_len, opts, devices, n_devices, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

void dray_release_cache(void) {
    try {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        g_cache.clear(); // a context still in use by a call on another thread is freed when that call returns
    } catch (...) {
    }
}

int dray_context_create(const int32_t* devices, int32_t n_devices, dray_context** out) {
    try {
        return context_create_impl(devices, n_devices, out);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

void dray_context_destroy(dray_context* ctx) {
    try {
        if (ctx) { std::lock_guard<std::mutex> busy(ctx->mu); }
        context_free(ctx);
    } catch (...) {
    }
}

int dray_context_set_scene(dray_context* ctx, const dray_sphere* spheres, size_t n_spheres) {
    try {
        if (!ctx) return DRAY_ERR_INVALID_ARG;
        std::lock_guard<std::mutex> lock(ctx->mu);
        return context_set_scene_locked(ctx, spheres, n_spheres);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_context_render(dray_context* ctx, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                        size_t buf_len, const dray_opts* opts, uint32_t sync, dray_stats* stats) {
    try {
        if (!ctx) return DRAY_ERR_INVALID_ARG;
        std::lock_guard<std::mutex> lock(ctx->mu);
        return context_render_locked(ctx, cam, color, raw, count, buf_len, opts, sync, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_context_render_progressive(dray_context* ctx, const dray_camera* cam, uint32_t* color, double* raw, int32_t* count,
                                    size_t buf_len, const dray_opts* opts, int32_t samples_per_pass, dray_present_fn present,
                                    void* user, dray_stats* stats) {
    try {
        if (!ctx) return DRAY_ERR_INVALID_ARG;
        std::lock_guard<std::mutex> lock(ctx->mu);
        return context_render_progressive_locked(ctx, cam, color, raw, count, buf_len, opts, samples_per_pass, present, user, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_context_pin(dray_context* ctx, void* host, size_t bytes) {
    try {
        if (!ctx || !host || bytes == 0) return DRAY_ERR_INVALID_ARG;
        std::lock_guard<std::mutex> lock(ctx->mu);
        return context_pin_locked(ctx, host, bytes);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_context_host_alloc(dray_context* ctx, size_t bytes, void** out) {
    try {
        if (!ctx || !out || bytes == 0) return DRAY_ERR_INVALID_ARG;
        *out = nullptr;
        std::lock_guard<std::mutex> lock(ctx->mu);
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
        std::memset(p, 0, bytes);
        ctx->pins.push_back({(char*)p, bytes, true});
        *out = p;
        return DRAY_OK;
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

void dray_context_unpin(dray_context* ctx) {
    try {
        if (!ctx) return;
        std::lock_guard<std::mutex> lock(ctx->mu);
        for (CtxShare& sh : ctx->shares) {
            DeviceGuard g(sh.device);
            if (sh.compute) (void)hipStreamSynchronize(sh.compute);
        }
        std::vector<dray_context::Pin> keep;
        for (auto& pin : ctx->pins) {
            if (pin.owned) keep.push_back(pin); // the context's own allocations live until it is destroyed
            else (void)hipHostUnregister(pin.p);
        }
        ctx->pins.swap(keep);
    } catch (...) {
    }
}

int dray_scene_create(const dray_sphere* spheres, size_t n_spheres, int32_t device, dray_scene** out) {
    try {
        return scene_create_impl(spheres, n_spheres, device, out);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

void dray_scene_destroy(dray_scene* scene) {
    try {
        scene_free(scene);
    } catch (...) {
    }
}

int dray_render_device(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw, int32_t* d_count,
                       size_t buf_len, const dray_opts* opts, void* stream, dray_stats* stats) {
    try {
        return render_device_impl(cam, scene, d_color, d_raw, d_count, buf_len, opts, stream, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_render_device_async(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw, int32_t* d_count,
                             size_t buf_len, const dray_opts* opts, void* stream, uint64_t* ticket) {
    try {
        return render_device_impl(cam, scene, d_color, d_raw, d_count, buf_len, opts, stream, nullptr, false, ticket);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_scene_collect(dray_scene* scene, uint64_t ticket, dray_stats* stats) {
    try {
        return collect_call(scene, ticket, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_render_device_noalloc(const dray_camera* cam, dray_scene* scene, uint32_t* d_color, double* d_raw,
                               int32_t* d_count, size_t buf_len, const dray_opts* opts, void* stream, dray_stats* stats) {
    try {
        return render_device_impl(cam, scene, d_color, d_raw, d_count, buf_len, opts, stream, stats, true);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_camera_rays(const dray_camera* cam, const dray_pixel_index* px, size_t n, dray_path_entry* out,
                     const dray_opts* opts) {
    try {
        return camera_rays_host_impl(cam, px, n, out, opts);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_path_step(dray_scene* scene, int32_t image_width, dray_path_entry* entries, size_t n, const dray_opts* opts,
                   dray_stats* stats) {
    try {
        return path_step_host_impl(scene, image_width, entries, n, opts, stats);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_camera_rays_device(const dray_camera* cam, const dray_pixel_index* d_px, size_t n, dray_path_entry* d_out,
                            const dray_opts* opts, void* stream) {
    try {
        return camera_rays_device_impl(cam, d_px, n, d_out, opts, stream);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_path_step_device(dray_scene* scene, int32_t image_width, dray_path_entry* d_entries, size_t n,
                          const dray_opts* opts, void* stream) {
    try {
        return path_step_device_impl(scene, image_width, d_entries, n, opts, stream, false);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_grid_inspect(const dray_sphere* spheres, size_t n_spheres, dray_grid_info* info, uint32_t* cell_begin,
                      int32_t* entry_sphere, int32_t* loose_sphere) {
    try {
        return grid_inspect_impl(spheres, n_spheres, info, cell_begin, entry_sphere, loose_sphere);
    } catch (...) {
        return DRAY_ERR_INTERNAL;
    }
}

int dray_scene_three_spheres(dray_sphere* out, size_t cap) {
    if (!out || cap < 4) return DRAY_ERR_BUFFER_TOO_SMALL;
    put_sphere(&out[0], 0., -100.5, -1., 100., DRAY_MAT_LAMBERTIAN, 0.8, 0.8, 0.0, 0.);
    put_sphere(&out[1], 0., 0., -1.2, 0.5, DRAY_MAT_LAMBERTIAN, 0.1, 0.2, 0.5, 0.);
    put_sphere(&out[2], -1., 0., -1., 0.5, DRAY_MAT_DIELECTRIC, 0., 0., 0., 1.5);
    put_sphere(&out[3], 1., 0., -1., 0.5, DRAY_MAT_METAL, 0.8, 0.6, 0.2, 1.0);
    return 4;
}

int dray_scene_recipe(uint64_t seed, int32_t lo, int32_t hi, double p_diffuse, double p_metal, double fuzz_max,
                      dray_sphere* out, size_t cap) { // main.rs:13-50, generalised
    if (!out || cap < 1) return DRAY_ERR_BUFFER_TOO_SMALL;
    if (hi < lo) return DRAY_ERR_INVALID_ARG;
    SceneStream rs;
    rs.key[0] = (uint32_t)seed;
    rs.key[1] = (uint32_t)(seed >> 32);
    size_t n = 0;
    put_sphere(&out[n++], 0., -1000., 0., 1000., DRAY_MAT_LAMBERTIAN, 0.5, 0.5, 0.5, 0.);
    for (int a = lo; a < hi; ++a)
        for (int b = lo; b < hi; ++b) {
            const double choose_mat = rs.next();
            const double cx = (double)a + 0.9 * rs.next();
            const double cz = (double)b + 0.9 * rs.next();
            const D3 dlt = sub(D3{cx, 0.2, cz}, D3{4., 0.2, 0.});
            if (std::sqrt(dot(dlt, dlt)) > 0.9) {
                if (n >= cap) return DRAY_ERR_BUFFER_TOO_SMALL;
                if (choose_mat < p_diffuse) {
                    const double r0 = rs.next(), r1 = rs.next(), r2 = rs.next();
                    const double s0 = rs.next(), s1 = rs.next(), s2 = rs.next();
                    put_sphere(&out[n++], cx, 0.2, cz, 0.2, DRAY_MAT_LAMBERTIAN, r0 * s0, r1 * s1, r2 * s2, 0.);
                } else if (choose_mat < p_diffuse + p_metal) {
                    const double a0 = rs.range(0.5, 1.), a1 = rs.range(0.5, 1.), a2 = rs.range(0.5, 1.);
                    const double fuzz = rs.range(0., fuzz_max);
                    put_sphere(&out[n++], cx, 0.2, cz, 0.2, DRAY_MAT_METAL, a0, a1, a2, fuzz);
                } else {
                    put_sphere(&out[n++], cx, 0.2, cz, 0.2, DRAY_MAT_DIELECTRIC, 0., 0., 0., 1.5);
                }
            }
        }
    if (n + 3 > cap) return DRAY_ERR_BUFFER_TOO_SMALL;
    put_sphere(&out[n++], 0., 1., 0., 1.0, DRAY_MAT_DIELECTRIC, 0., 0., 0., 1.5);
    put_sphere(&out[n++], -4., 1., 0., 1.0, DRAY_MAT_LAMBERTIAN, 0.4, 0.2, 0.1, 0.);
    put_sphere(&out[n++], 4., 1., 0., 1.0, DRAY_MAT_METAL, 0.7, 0.6, 0.5, 0.0);
    return (int)n;
}

int dray_scene_final(uint64_t seed, dray_sphere* out, size_t cap) {
    return dray_scene_recipe(seed, -11, 11, 0.8, 0.15, 0.5, out, cap);
}

int dray_write_ppm(const char* path, const uint32_t* color, int32_t width, int32_t height, int32_t binary) {
    if (!path || !color || width <= 0 || height <= 0) return DRAY_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, binary ? "wb" : "w");
    if (!f) return DRAY_ERR_INVALID_ARG;
    bool ok = std::fprintf(f, "%s\n%d %d\n255\n", binary ? "P6" : "P3", width, height) > 0;
    const size_t n = (size_t)width * (size_t)height;
    if (binary) {
        std::vector<unsigned char> row((size_t)width * 3u);
        for (int32_t j = 0; j < height && ok; ++j) {
            for (int32_t i = 0; i < width; ++i) {
                const uint32_t c = color[(size_t)j * (size_t)width + (size_t)i];
                row[3 * i + 0] = (unsigned char)(c >> 16);
                row[3 * i + 1] = (unsigned char)(c >> 8);
                row[3 * i + 2] = (unsigned char)c;
            }
            ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    } else {
        for (size_t k = 0; k < n && ok; ++k)
            ok = std::fprintf(f, "%u %u %u\n", (color[k] >> 16) & 255u, (color[k] >> 8) & 255u, color[k] & 255u) > 0;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? DRAY_OK : DRAY_ERR_INTERNAL;
}

} // extern "C"
