// Reference-order building blocks of the f64 (and plain-f32) render kernels for gfx950: every arithmetic step is
// written in the operation order of the Rust reference, so that the f64 instantiations can be compared with the CPU
// oracle bit for bit.  Shared by render_ref.hip (one thread per pixel, the path step) and render_f64.hip (the frame
// as a stream of samples).  Include only from files built with -ffp-contract=off.
#pragma once
#pragma clang fp contract(off)

#include "dray_device.h"
#include "dray_internal.h"

namespace dray {
namespace {


template <typename T>
struct V3 {
    T x, y, z;
};

// vec3.rs:131-257, element-wise, no fused operations.
template <typename T> __device__ __forceinline__ V3<T> vadd(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> vsub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> vmul(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename T> __device__ __forceinline__ V3<T> vscale(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
// Three quotients by one divisor (vec3.rs:229-238) -- and, wherever a ray is tested against several spheres, the
// roots' division by |d|^2 (sphere.rs:38,41).  An IEEE f64 division on gfx950 is a sequence the compiler writes out:
// both operands scaled if their exponents are extreme (v_div_scale), y = v_rcp_f64 of the divisor refined by two
// Newton steps, q0 = x y, e = fma(-s, q0, x), q = fma(e, y, q0) (v_div_fmas), special cases patched (v_div_fixup).
// For operands whose exponents lie in a middle band nothing is scaled and nothing patched, so y depends on the
// divisor alone: it is formed once and every quotient costs three operations, with the bits the division yields
// (tools/ubench_div.hip compares the two on the GPU).  Outside the band -- zero, subnormal, huge, infinite, NaN --
// the division itself is taken.
__device__ __forceinline__ bool div_band(double x) { // 2^-323 <= |x| < 2^328: no scaling for any pair in the band
    const uint32_t e = ((uint32_t)__double2hiint(x) >> 20) & 0x7ffu;
    return e - 700u <= 650u;
}
__device__ __forceinline__ double refined_rcp(double s) {
    const double r0 = __builtin_amdgcn_rcp(s);
    const double e0 = __builtin_fma(-s, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-s, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
__device__ __forceinline__ double div_by(double x, double s, double y) { // x / s for y = refined_rcp(s), both in the band
    const double q0 = x * y;
    const double e = __builtin_fma(-s, q0, x);
    return __builtin_fma(e, y, q0);
}
__device__ __forceinline__ bool div_band4(double a, double b, double c, double d) { // all four in the band (no branches)
    auto off = [](double x) { return (((uint32_t)__double2hiint(x) >> 20) & 0x7ffu) - 700u; };
    return max(max(off(a), off(b)), max(off(c), off(d))) <= 650u;
}
template <typename T> __device__ __forceinline__ V3<T> vdiv(V3<T> a, T s) {
    if constexpr (sizeof(T) == sizeof(double)) {
        if (div_band4(s, a.x, a.y, a.z)) {
            const double y = refined_rcp(s);
            return {div_by(a.x, s, y), div_by(a.y, s, y), div_by(a.z, s, y)};
        }
    }
    return {a.x / s, a.y / s, a.z / s};
}
template <typename T> __device__ __forceinline__ V3<T> vneg(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> __device__ __forceinline__ T vdot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vec3.rs:64-66
template <typename T> __device__ __forceinline__ T vlen2(V3<T> a) { return vdot(a, a); }
template <typename T> __device__ __forceinline__ T rsqrt_exact(T x);
template <> __device__ __forceinline__ double rsqrt_exact<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float rsqrt_exact<float>(float x) { return sqrtf(x); }
template <typename T> __device__ __forceinline__ T vsqrt(T x) { return rsqrt_exact<T>(x); }
template <typename T> __device__ __forceinline__ V3<T> vunit(V3<T> a) { return vdiv(a, vsqrt(vlen2(a))); } // vec3.rs:120-122

template <typename T> __device__ __forceinline__ V3<T> reflect(V3<T> v, V3<T> n) { // vec3.rs:97-102
    V3<T> b = vscale(n, vdot(v, n));
    return vsub(v, vscale(b, (T)2.0));
}

template <typename T> __device__ __forceinline__ V3<T> refract(V3<T> uv, V3<T> n, T eta) { // vec3.rs:104-109
    T cos_theta = fmin(vdot(vneg(uv), n), (T)1.0);
    V3<T> perp = vscale(vadd(uv, vscale(n, cos_theta)), eta);
    T k = -(vsqrt(fabs((T)1.0 - vlen2(perp))));
    return vadd(perp, vscale(n, k));
}

template <typename T> __device__ __forceinline__ T tiny_len2();
template <> __device__ __forceinline__ double tiny_len2<double>() { return 1e-160; } // vec3.rs:80
template <> __device__ __forceinline__ float tiny_len2<float>() { return 1e-30f; }   // f32 stand-in for 1e-160

// vec3.rs:76-84.  Attempt a uses uniforms 4a, 4a+1, 4a+2 of the stage = words 0..2 of block a.
// One attempt: the point in the cube, its squared length, and whether the reference would keep it (:80).
template <typename T> __device__ __forceinline__ bool unit_vector_attempt(const uint32_t (&w)[4], V3<T>& p, T& lensq) {
    p.x = (T)-1. + ((T)1. - (T)-1.) * u01<T>(w[0]); // prelude.rs:27-30
    p.y = (T)-1. + ((T)1. - (T)-1.) * u01<T>(w[1]);
    p.z = (T)-1. + ((T)1. - (T)-1.) * u01<T>(w[2]);
    lensq = vlen2(p);
    return tiny_len2<T>() <= lensq && lensq <= (T)1.;
}
template <typename T> __device__ __forceinline__ V3<T> random_unit_vector(const Stream& st) {
    for (uint32_t a = 0;; ++a) {
        uint32_t w[4];
        st.block(a, w);
        V3<T> p;
        T lensq;
        if (unit_vector_attempt<T>(w, p, lensq)) return vdiv(p, vsqrt(lensq));
    }
}

// vec3.rs:111-118, one attempt: two consecutive uniforms of stage 0.
template <typename T> __device__ __forceinline__ bool unit_disk_attempt(uint32_t wx, uint32_t wy, T& px, T& py) {
    px = (T)-1. + ((T)1. - (T)-1.) * u01<T>(wx);
    py = (T)-1. + ((T)1. - (T)-1.) * u01<T>(wy);
    return px * px + py * py + (T)0. * (T)0. < (T)1.;
}

template <typename T> struct Hit {
    V3<T> p, normal;
    T t;
    bool front_face;
    int32_t index;
};

// What the closest-hit query reads: the sphere list, and for the f64 kernels the uniform grid the scene builder
// lays over the list (dray_internal.h) as a way to skip spheres the ray does not come near.
struct RefGrid {
    GridDev grid;                // null cells: no grid
    const int32_t* narrow_index; // grid / loose position -> position in the caller's list
    const int32_t* wide_index;   // spheres kept out of the grid for their size
    int32_t n_wide;
    int32_t use;                 // 0: scan the whole list
    // list entries that are bounding boxes: min xyz, max xyz, and the entry's position in the list
    const double* box64;
    const int32_t* box_index;
    int32_t n_boxes;
    // the stream kernel's copies of the lists, with the f64 geometry beside the list positions (SceneDev)
    const double* entry_geom64;
    const int32_t* entry_index;
    const double* always_geom64;
    const int32_t* always_index;
    int32_t n_always;
};

// BoundingBox::hit, bounding_box.rs:50-76, up to the choice of the root, for an upper bound of +infinity (a smaller
// bound accepts the same root or none, as for a sphere: t_max >= t_min).  As written there: only axes 0 and 1 are
// looked at (:53), a plane's distance is `min - origin / direction` (:55-56), and f64::max / f64::min return the
// other operand when one is NaN, as fmax / fmin do.
template <typename T>
__device__ __forceinline__ bool box_root(const double* __restrict__ box, V3<T> ro, V3<T> rd, T& root) {
    const T tmin = (T)0.001;
    T t_min = -(T)INFINITY, t_max = (T)INFINITY;
    const T o[2] = {ro.x, ro.y}, d[2] = {rd.x, rd.y};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (d[a] != (T)0.0) {
            T t0 = (T)box[a] - o[a] / d[a];
            T t1 = (T)box[3 + a] - o[a] / d[a];
            t_min = fmax(t_min, fmin(t0, t1));
            t_max = fmin(t_max, fmax(t0, t1));
        }
    }
    if (!(t_max >= t_min)) return false;
    root = t_min;
    if (!(tmin < root)) { // interval.rs:28-30, strict
        root = t_max;
        if (!(tmin < root)) return false;
    }
    return root < (T)INFINITY;
}

// The boxes of the list against the closest sphere hit so far: smallest root, on equal roots the earlier entry.
// A box hit sets t and p (bounding_box.rs:63-64); its material reads neither normal nor face.
struct BoxList {
    const double* box64;
    const int32_t* box_index;
    int32_t n_boxes;
};
template <typename T>
__device__ __forceinline__ bool boxes_closest(const BoxList bl, V3<T> ro, V3<T> rd, T& closest, int32_t& best) {
    bool any = false;
    for (int32_t q = 0; q < bl.n_boxes; ++q) {
        const int32_t k = bl.box_index[q];
        T root;
        if (box_root<T>(bl.box64 + 6 * q, ro, rd, root) && (root < closest || (root == closest && k < best))) {
            closest = root;
            best = k;
            any = true;
        }
    }
    return any;
}
// The record of a box hit: t and p are set (bounding_box.rs:63-64); its material reads neither normal nor face.
template <typename T>
__device__ __forceinline__ void box_record(V3<T> ro, V3<T> rd, T closest, int32_t best, Hit<T>& rec) {
    rec.t = closest;
    rec.p = vadd(ro, vscale(rd, closest)); // ray.rs:17-19
    rec.normal = {(T)0., (T)0., (T)0.};
    rec.front_face = false;
    rec.index = best;
}
template <typename T>
__device__ __forceinline__ bool boxes_hit(const BoxList bl, V3<T> ro, V3<T> rd, T closest, int32_t best, Hit<T>& rec) {
    const bool any = boxes_closest<T>(bl, ro, rd, closest, best);
    if (any) box_record<T>(ro, rd, closest, best, rec);
    return any;
}

// sphere.rs:28-48 for sphere k up to the choice of the root: the root the reference would accept for an upper bound
// of +infinity, or no root.  (With a smaller bound the reference accepts the same root or none: the far root is only
// tried when the near one is not above tmin, or not below the bound -- and then the far one is not below it either.)
template <typename T>
__device__ __forceinline__ bool sphere_root_of(V3<T> center, T radius, V3<T> ro, V3<T> rd, T& root) {
    const T tmin = (T)0.001;
    V3<T> oc = vsub(center, ro);
    T a = vlen2(rd);
    T h = vdot(rd, oc);
    T c = vlen2(oc) - radius * radius;
    T discriminant = h * h - a * c;
    if (discriminant < (T)0.) return false;
    T sqrtd = vsqrt(discriminant);
    root = (h - sqrtd) / a;
    if (!(tmin < root)) { // interval.rs:28-30, strict
        root = (h + sqrtd) / a;
        if (!(tmin < root)) return false;
    }
    return root < (T)INFINITY;
}
template <typename T>
__device__ __forceinline__ bool sphere_root(const T* __restrict__ geom, int32_t k, V3<T> ro, V3<T> rd, T& root) {
    return sphere_root_of<T>(V3<T>{geom[4 * k + 0], geom[4 * k + 1], geom[4 * k + 2]}, geom[4 * k + 3], ro, rd, root);
}

// sphere.rs:49-55 for the accepted root
template <typename T>
__device__ __forceinline__ void fill_hit(const T* __restrict__ geom, int32_t k, V3<T> ro, V3<T> rd, T root, Hit<T>& rec) {
    const V3<T> center = {geom[4 * k + 0], geom[4 * k + 1], geom[4 * k + 2]};
    const T radius = geom[4 * k + 3];
    rec.t = root;
    rec.p = vadd(ro, vscale(rd, root)); // ray.rs:17-19
    V3<T> on = vdiv(vsub(rec.p, center), radius);
    rec.front_face = vdot(rd, on) < (T)0.; // hittable.rs:16-22
    rec.normal = rec.front_face ? on : vscale(on, (T)-1.);
    rec.index = k;
}

// hittable_list.rs:60-74 around sphere.rs:28-57: insertion order, shrinking upper bound.
template <typename T>
__device__ __forceinline__ bool list_hit_linear(const T* __restrict__ geom, int32_t n, const RefGrid& rg, V3<T> ro,
                                                V3<T> rd, Hit<T>& rec) {
    T closest = (T)INFINITY;
    int32_t best = 0x7fffffff;
    for (int32_t k = 0; k < n; ++k) {
        T root;
        if (sphere_root(geom, k, ro, rd, root) && root < closest) { // strict: on equal t the earlier sphere stays
            closest = root;
            best = k;
        }
    }
    if (boxes_hit<T>(BoxList{rg.box64, rg.box_index, rg.n_boxes}, ro, rd, closest, best, rec)) return true;
    if (best == 0x7fffffff) return false;
    fill_hit(geom, best, ro, rd, closest, rec);
    return true;
}

// The same closest hit -- smallest root, on equal roots the sphere inserted first -- found by walking the ray
// through the uniform grid (3-D DDA in f64) and testing only the spheres listed in the cells it crosses, front to
// back, until the best root lies before the current cell's exit.  Every sphere is tested with the reference's own
// expressions (sphere_root), so root, point and normal are bit for bit those of the linear scan; the order in which
// spheres are met does not matter because the minimum is taken over (root, position).  The grid lists a sphere in
// every cell its ball grown by 1/256 of a cell overlaps, which f64 rounding of the walk cannot bridge; the wide and
// the loose spheres are tested for every ray.  A ray that is not finite, or a walk that takes more steps than a
// straight line can, falls back to the linear scan.
__device__ __forceinline__ bool list_hit_grid(const double* __restrict__ geom, int32_t n, const RefGrid& rg,
                                              V3<double> ro, V3<double> rd, Hit<double>& rec) {
    const GridDev& g = rg.grid;
    const double o[3] = {ro.x, ro.y, ro.z}, d[3] = {rd.x, rd.y, rd.z};
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) finite = finite && fabs(o[a]) < 1e300 && fabs(d[a]) < 1e300; // false for NaN too
    if (!finite) return list_hit_linear<double>(geom, n, rg, ro, rd, rec);

    double closest = INFINITY;
    int32_t best = 0x7fffffff;
    auto test = [&](int32_t k) {
        double root;
        if (sphere_root<double>(geom, k, ro, rd, root) && (root < closest || (root == closest && k < best))) {
            closest = root;
            best = k;
        }
    };
    for (int32_t q = 0; q < rg.n_wide; ++q) test(rg.wide_index[q]);
    for (int32_t q = 0; q < g.n_loose; ++q) test(rg.narrow_index[g.loose[q]]);

    if (g.n[0] > 0) {
        double A[3], B[3], t_in = 0., t_out = INFINITY, bmax = 0.;
        bool inside_slabs = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (fabs(d[a]) > 1e-250) {
                const double inv = 1.0 / d[a];
                A[a] = (double)g.cell[a] * inv;
                const double t0 = ((double)g.lo[a] - o[a]) * inv, t1 = ((double)g.hi[a] - o[a]) * inv;
                B[a] = d[a] >= 0. ? t0 + A[a] : t0; // leaving cell k towards larger indices happens at plane k + 1
                t_in = fmax(t_in, fmin(t0, t1));
                t_out = fmin(t_out, fmax(t0, t1));
                bmax = fmax(bmax, fmax(fabs(t0), fabs(B[a])));
            } else { // parallel to the axis' planes: never crosses one; misses the grid if outside its slab
                A[a] = 0.;
                B[a] = INFINITY;
                if (o[a] < (double)g.lo[a] || o[a] > (double)g.hi[a]) inside_slabs = false;
            }
        }
        const double slack = bmax * 1e-9;
        if (inside_slabs && t_in <= t_out + slack) {
            int32_t c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double at = floor((o[a] + d[a] * t_in - (double)g.lo[a]) * (double)g.inv_cell[a]);
                c[a] = (int32_t)fmin(fmax(at, 0.), (double)(g.n[a] - 1));
            }
            const int32_t max_steps = g.n[0] + g.n[1] + g.n[2] + 4;
            int32_t steps = 0;
            for (;; ++steps) { // (one action per turn -- test one sphere or leave the cell -- was tried: 15 % slower)
                if (steps > max_steps) return list_hit_linear<double>(geom, n, rg, ro, rd, rec);
                const uint2 cellrec = g.cells[((size_t)c[2] * g.n[1] + c[1]) * g.n[0] + c[0]];
                for (uint32_t e = cellrec.x; e < cellrec.x + cellrec.y; ++e) test(rg.narrow_index[g.entry_pos[e]]);
                const double tx = (double)c[0] * A[0] + B[0], ty = (double)c[1] * A[1] + B[1], tz = (double)c[2] * A[2] + B[2];
                const double t_exit = fmin(tx, fmin(ty, tz));
                if (closest < t_exit * (1. - 1e-9) - slack) break; // nothing beyond this cell can be closer
                const int ax = (tx <= ty && tx <= tz) ? 0 : (ty <= tz ? 1 : 2);
                c[ax] += A[ax] < 0. ? -1 : 1;
                if (c[ax] < 0 || c[ax] >= g.n[ax]) break;
            }
        }
    }
    if (boxes_hit<double>(BoxList{rg.box64, rg.box_index, rg.n_boxes}, ro, rd, closest, best, rec)) return true;
    if (best == 0x7fffffff) return false;
    fill_hit<double>(geom, best, ro, rd, closest, rec);
    return true;
}

template <typename T>
__device__ __forceinline__ bool list_hit(const T* __restrict__ geom, int32_t n, const RefGrid& rg, V3<T> ro, V3<T> rd,
                                         Hit<T>& rec) {
    if constexpr (sizeof(T) == sizeof(double)) {
        if (rg.use) return list_hit_grid(geom, n, rg, ro, rd, rec);
    }
    return list_hit_linear<T>(geom, n, rg, ro, rd, rec);
}

template <typename T> __device__ __forceinline__ T reflectance(T cosine, T ri) { // material.rs:93-98
    T r0 = ((T)1. - ri) / ((T)1. + ri);
    r0 = r0 * r0;
    T x = (T)1. - cosine;
    T x2 = x * x;
    T x5 = x * (x2 * x2); // powi(5)
    return r0 + ((T)1. - r0) * x5;
}

// material.rs scatter of the five material kinds.  Returns false when the ray is absorbed.  `so` comes in as the
// incoming ray's origin and goes out as the scattered ray's: the hit point, except for Transparent.  The two random
// draws a scatter may make come through callables: `unit_vector()` is random_unit_vector (vec3.rs:76-84),
// `uniform()` the dielectric's random_f64 (material.rs:114); each is called at most once, and only where the
// reference draws.
template <typename T, typename UnitVector, typename Uniform>
__device__ __forceinline__ bool scatter_with(uint32_t kind, const T* __restrict__ mat, V3<T> rd, const Hit<T>& rec,
                                             UnitVector&& unit_vector, Uniform&& uniform, V3<T>& so, V3<T>& sd) {
    if (kind == DRAY_MAT_TRANSPARENT) { // material.rs:25-31: the incoming ray, unchanged
        sd = rd;
        return true;
    }
    so = rec.p;
    if (kind == DRAY_MAT_LAMBERTIAN) { // material.rs:46-56
        V3<T> dir = vadd(rec.normal, unit_vector());
        const T s = (T)1e-8; // vec3.rs:42-45
        if (fabs(dir.x) < s && fabs(dir.y) < s && fabs(dir.z) < s) dir = rec.normal;
        sd = dir;
        return true;
    }
    if (kind == DRAY_MAT_METAL) { // material.rs:73-80
        V3<T> refl = reflect(rd, rec.normal);
        refl = vadd(vunit(refl), vscale(unit_vector(), mat[3]));
        sd = refl;
        return vdot(sd, rec.normal) > (T)0.;
    }
    if (kind == DRAY_MAT_DIELECTRIC) { // material.rs:103-122
        const T ior = mat[3];
        T ri = rec.front_face ? (T)1.0 / ior : ior;
        V3<T> ud = vunit(rd);
        T cos_theta = fmin(vdot(vneg(ud), rec.normal), (T)1.0);
        T sin_theta = vsqrt((T)1.0 - cos_theta * cos_theta);
        bool reflects = ri * sin_theta > (T)1.0;
        if (!reflects) reflects = reflectance(cos_theta, ri) > uniform(); // drawn only when refraction is possible
        sd = reflects ? reflect(ud, rec.normal) : refract(ud, rec.normal, ri);
        return true;
    }
    return false; // material.rs:13-19
}

// The same with the draws made on the spot from the stage's stream: the unit vector's attempts are blocks 0, 1, ...,
// the dielectric's uniform is word 0 of block 0.
template <typename T>
__device__ __forceinline__ bool scatter(uint32_t kind, const T* __restrict__ mat, V3<T> rd, const Hit<T>& rec,
                                        const Stream& st, V3<T>& so, V3<T>& sd) {
    return scatter_with<T>(
        kind, mat, rd, rec, [&]() { return random_unit_vector<T>(st); },
        [&]() {
            uint32_t w[4];
            st.block(0, w);
            return u01<T>(w[0]);
        },
        so, sd);
}

// camera.rs:238-262 from the draws already made: `w` = block 0 of the sample's stage-0 stream (uniforms 0, 1 are the
// offsets in the pixel), (px, py) = the accepted point of random_in_unit_disk (read only with defocus).
template <typename T, typename Cam> // Cam: a CamDev<T>, in whichever address space it lives
__device__ __forceinline__ void get_ray_with(const Cam& cam, int32_t i, int32_t j, const uint32_t (&w)[4], T px, T py,
                                             V3<T>& ro, V3<T>& rd) {
    const T ox = u01<T>(w[0]) - (T)0.5, oy = u01<T>(w[1]) - (T)0.5;
    const V3<T> p00 = {cam.p00[0], cam.p00[1], cam.p00[2]};
    const V3<T> du = {cam.du[0], cam.du[1], cam.du[2]}, dv = {cam.dv[0], cam.dv[1], cam.dv[2]};
    const V3<T> center = {cam.center[0], cam.center[1], cam.center[2]};
    V3<T> ps = vadd(vadd(p00, vscale(du, (T)i + ox)), vscale(dv, (T)j + oy));
    if (!cam.defocus) {
        ro = center;
    } else {
        const V3<T> ddu = {cam.ddu[0], cam.ddu[1], cam.ddu[2]}, ddv = {cam.ddv[0], cam.ddv[1], cam.ddv[2]};
        ro = vadd(vadd(center, vscale(ddu, px)), vscale(ddv, py));
    }
    rd = vsub(ps, ro);
}

// camera.rs:238-262 with the draws made on the spot
template <typename T>
__device__ __forceinline__ void get_ray(const CamDev<T>& cam, int32_t i, int32_t j, const Stream& st, V3<T>& ro,
                                        V3<T>& rd) {
    uint32_t w0[4], w[4];
    st.block(0, w0);
    T px = (T)0., py = (T)0.;
    if (cam.defocus) {
        // vec3.rs:111-118; attempt m uses uniforms 2+2m, 3+2m of stage 0
        w[0] = w0[0]; w[1] = w0[1]; w[2] = w0[2]; w[3] = w0[3];
        for (uint32_t m = 0;; ++m) {
            const uint32_t d0 = 2u + 2u * m;
            if (m > 0 && (d0 & 3u) == 0u) st.block(d0 >> 2, w);
            if (unit_disk_attempt<T>(w[d0 & 3u], w[(d0 & 3u) + 1u], px, py)) break;
        }
    }
    get_ray_with<T>(cam, i, j, w0, px, py, ro, rd);
}

inline RefGrid ref_grid_of(const SceneDev& scene, bool use) {
    RefGrid rg{};
    rg.grid = scene.grid;
    rg.narrow_index = scene.narrow_index;
    rg.wide_index = scene.wide_index;
    rg.n_wide = scene.n_wide;
    rg.use = (use && scene.narrow_index && scene.wide_index) ? 1 : 0;
    rg.box64 = scene.box64;
    rg.box_index = scene.box_index;
    rg.n_boxes = scene.n_boxes;
    rg.entry_geom64 = scene.entry_geom64;
    rg.entry_index = scene.entry_index;
    rg.always_geom64 = scene.always_geom64;
    rg.always_index = scene.always_index;
    rg.n_always = scene.n_always;
    return rg;
}

} // namespace
} // namespace dray
