/*
 * dray_oracle.c -- CPU restatement (plain C, f64) of the reference path tracer's
 * per-pixel render path.  TEST INFRASTRUCTURE ONLY (see dray_oracle.h): checker
 * for the HIP path; never shipped, never the thing measured except as the bench's
 * cpu_baseline leg.  PARITY UNPINNED beyond Vec3: the Rust reference cannot be
 * built here and its own tests cover only src/raytracer/vec3.rs:260-450.
 *
 * Build with -ffp-contract=off: every a*b+c below must round twice, as the
 * reference's plain f64 expressions do.
 *
 * Every function names the reference lines it follows.
 */
#include "dray_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ RNG -- */

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * Replaces rand::random (src/raytracer/prelude.rs:22-25), whose stream is OS-seeded. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* U[0,1) with 24 significant bits: exactly representable in f32 and f64, so an
 * f32 and an f64 consumer of one stream see the same uniforms.  Half-open like
 * rand::random::<f64>() (prelude.rs:24). */
double orc_u01(uint32_t word) { return (double)(word >> 8) * (1.0 / 16777216.0); }

double orc_rng_draw(orc_rng* rng, uint32_t draw_index) {
    rng->draws++;
    if (rng->kind == 1) {
        if (rng->script_pos >= rng->script_len) return 0.5; /* script exhausted: harmless mid value */
        return rng->script[rng->script_pos++];
    }
    uint32_t ctr[4] = {rng->pixel, rng->sample, rng->stage, draw_index >> 2};
    uint32_t out[4];
    orc_philox4x32_10(ctr, rng->key, out);
    return orc_u01(out[draw_index & 3u]);
}

/* prelude.rs:27-30 */
static double rng_range(orc_rng* rng, uint32_t draw_index, double lo, double hi) {
    return lo + (hi - lo) * orc_rng_draw(rng, draw_index);
}

/* ----------------------------------------------------------------- Vec3 -- */

void orc_v_add(const double* a, const double* b, double* o) { /* vec3.rs:153-162 */
    double x = a[0] + b[0], y = a[1] + b[1], z = a[2] + b[2];
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_sub(const double* a, const double* b, double* o) { /* vec3.rs:172-181 */
    double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_mul(const double* a, const double* b, double* o) { /* vec3.rs:191-200 */
    double x = a[0] * b[0], y = a[1] * b[1], z = a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_scale(const double* a, double s, double* o) { /* vec3.rs:202-219 (both operand orders) */
    double x = a[0] * s, y = a[1] * s, z = a[2] * s;
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_div(const double* a, double s, double* o) { /* vec3.rs:229-238: three divides, no reciprocal */
    double x = a[0] / s, y = a[1] / s, z = a[2] / s;
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_divv(const double* a, const double* b, double* o) { /* vec3.rs:240-249 */
    double x = a[0] / b[0], y = a[1] / b[1], z = a[2] / b[2];
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_neg(const double* a, double* o) { /* vec3.rs:131-137 */
    double x = -a[0], y = -a[1], z = -a[2];
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_min(const double* a, const double* b, double* o) { /* vec3.rs:48-54 */
    double x = fmin(a[0], b[0]), y = fmin(a[1], b[1]), z = fmin(a[2], b[2]);
    o[0] = x; o[1] = y; o[2] = z;
}
void orc_v_max(const double* a, const double* b, double* o) { /* vec3.rs:56-62 */
    double x = fmax(a[0], b[0]), y = fmax(a[1], b[1]), z = fmax(a[2], b[2]);
    o[0] = x; o[1] = y; o[2] = z;
}
double orc_v_dot(const double* a, const double* b) { /* vec3.rs:64-66, left to right */
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
void orc_v_cross(const double* a, const double* b, double* o) { /* vec3.rs:68-74 */
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
double orc_v_length_squared(const double* a) { return orc_v_dot(a, a); } /* vec3.rs:30-32 */
double orc_v_length(const double* a) { return sqrt(orc_v_length_squared(a)); } /* vec3.rs:26-28 */
void orc_v_unit(const double* a, double* o) { orc_v_div(a, orc_v_length(a), o); } /* vec3.rs:120-122 */
int orc_v_near_zero(const double* a) { /* vec3.rs:42-45 */
    const double s = 1e-8;
    return fabs(a[0]) < s && fabs(a[1]) < s && fabs(a[2]) < s;
}
void orc_v_reflect(const double* v, const double* n, double* o) { /* vec3.rs:97-102 */
    double b[3], b2[3];
    orc_v_scale(n, orc_v_dot(v, n), b);
    orc_v_scale(b, 2.0, b2);
    orc_v_sub(v, b2, o);
}
void orc_v_refract(const double* uv, const double* n, double eta, double* o) { /* vec3.rs:104-109 */
    double nuv[3], t[3], perp[3], par[3];
    orc_v_neg(uv, nuv);
    double cos_theta = fmin(orc_v_dot(nuv, n), 1.0);
    orc_v_scale(n, cos_theta, t);
    orc_v_add(uv, t, t);
    orc_v_scale(t, eta, perp);
    double k = -(sqrt(fabs(1.0 - orc_v_length_squared(perp))));
    orc_v_scale(n, k, par);
    orc_v_add(perp, par, o);
}

/* vec3.rs:76-84: rejection sampling in the cube, accepted iff 1e-160 <= |p|^2 <= 1. */
void orc_random_unit_vector(orc_rng* rng, double* o) {
    for (uint32_t attempt = 0;; ++attempt) {
        double p[3];
        p[0] = rng_range(rng, 4u * attempt + 0u, -1., 1.);
        p[1] = rng_range(rng, 4u * attempt + 1u, -1., 1.);
        p[2] = rng_range(rng, 4u * attempt + 2u, -1., 1.);
        double lensq = orc_v_length_squared(p);
        if (1e-160 <= lensq && lensq <= 1.) {
            orc_v_div(p, sqrt(lensq), o);
            return;
        }
        if (rng->kind == 1 && rng->script_pos >= rng->script_len) { /* scripted run ran dry */
            o[0] = 1.0; o[1] = 0.0; o[2] = 0.0;
            return;
        }
    }
}

/* vec3.rs:111-118: accepted iff |p|^2 < 1 (strict). */
void orc_random_in_unit_disk(orc_rng* rng, double* o) {
    for (uint32_t attempt = 0;; ++attempt) {
        double p[3];
        p[0] = rng_range(rng, 2u + 2u * attempt, -1., 1.);
        p[1] = rng_range(rng, 3u + 2u * attempt, -1., 1.);
        p[2] = 0.0;
        if (orc_v_length_squared(p) < 1.) {
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            return;
        }
        if (rng->kind == 1 && rng->script_pos >= rng->script_len) {
            o[0] = 0.0; o[1] = 0.0; o[2] = 0.0;
            return;
        }
    }
}

/* ------------------------------------------------------------- Interval -- */

int orc_surrounds(double lo, double hi, double x) { return lo < x && x < hi; } /* interval.rs:28-30 */
double orc_clamp(double lo, double hi, double x) {                             /* interval.rs:32-36 */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* ------------------------------------------------- Geometry & materials -- */

/* sphere.rs:28-57 and hittable.rs:16-22 */
int orc_sphere_hit(const orc_sphere* s, const double* ro, const double* rd, double tmin, double tmax, orc_hit* rec) {
    double oc[3];
    orc_v_sub(s->center, ro, oc);
    double a = orc_v_length_squared(rd);
    double h = orc_v_dot(rd, oc);
    double c = orc_v_length_squared(oc) - s->radius * s->radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (h - sqrtd) / a;
    if (!orc_surrounds(tmin, tmax, root)) {
        root = (h + sqrtd) / a;
        if (!orc_surrounds(tmin, tmax, root)) return 0;
    }
    rec->t = root;
    double td[3];
    orc_v_scale(rd, rec->t, td); /* ray.rs:17-19 */
    orc_v_add(ro, td, rec->p);
    double on[3];
    orc_v_sub(rec->p, s->center, on);
    orc_v_div(on, s->radius, on);
    rec->front_face = orc_v_dot(rd, on) < 0.;
    if (rec->front_face) {
        rec->normal[0] = on[0]; rec->normal[1] = on[1]; rec->normal[2] = on[2];
    } else {
        orc_v_scale(on, -1., rec->normal);
    }
    return 1;
}

/* BoundingBox::hit, bounding_box.rs:50-76.  The loop covers axes 0 and 1 only (:53), the plane distances are
 * `min - origin / direction` (:55-56; the division binds tighter than the subtraction), f64::max / f64::min return
 * the other operand when one is NaN (fmax / fmin do the same), and only t and p of the record are written: normal and
 * front_face stay what the record held (its material, Transparent, reads neither). */
int orc_box_hit(const orc_box* b, const double* ro, const double* rd, double tmin, double tmax, orc_hit* rec) {
    double t_min = -INFINITY, t_max = INFINITY;
    for (int a = 0; a < 2; ++a) {
        if (rd[a] != 0.0) {
            double t0 = b->min[a] - ro[a] / rd[a];
            double t1 = b->max[a] - ro[a] / rd[a];
            t_min = fmax(t_min, fmin(t0, t1));
            t_max = fmin(t_max, fmax(t0, t1));
        }
    }
    if (t_max >= t_min) {
        double t;
        if (orc_surrounds(tmin, tmax, t_min)) t = t_min;
        else if (orc_surrounds(tmin, tmax, t_max)) t = t_max;
        else return 0;
        rec->t = t;
        double td[3];
        orc_v_scale(rd, t, td); /* ray.rs:17-19 */
        orc_v_add(ro, td, rec->p);
        return 1;
    }
    return 0;
}

/* hittable_list.rs:60-74: insertion order, shrinking upper bound, strict compare
 * (on equal t the earlier object stays). */
int orc_list_hit(const orc_sphere* s, size_t n, const double* ro, const double* rd, double tmin, double tmax,
                 orc_hit* rec) {
    orc_hit tmp;
    memset(&tmp, 0, sizeof tmp);
    int hit_anything = 0;
    double closest = tmax;
    for (size_t k = 0; k < n; ++k) {
        const int hit = s[k].shape == ORC_SHAPE_BOX ? orc_box_hit((const orc_box*)&s[k], ro, rd, tmin, closest, &tmp)
                                                    : orc_sphere_hit(&s[k], ro, rd, tmin, closest, &tmp);
        if (hit) {
            hit_anything = 1;
            closest = tmp.t;
            tmp.index = (int32_t)k;
            *rec = tmp;
        }
    }
    return hit_anything;
}

/* material.rs:93-98; powi(5) written as the multiply chain x*((x*x)*(x*x)). */
double orc_reflectance(double cosine, double ri) {
    double r0 = (1. - ri) / (1. + ri);
    r0 = r0 * r0;
    double x = 1. - cosine;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1. - r0) * x5;
}

/* material.rs:13-19 (absorb), :44-57 (Lambertian), :71-81 (Metal), :101-123 (Dialectric).
 * Returns 1 if scattered.  The scattered origin is always the hit point. */
int orc_scatter(const orc_sphere* s, const double* rd, const orc_hit* rec, orc_rng* rng, double* att, double* so,
                double* sd) {
    so[0] = rec->p[0]; so[1] = rec->p[1]; so[2] = rec->p[2];
    switch (s->kind) {
    case ORC_MAT_LAMBERTIAN: {
        double ruv[3], dir[3];
        orc_random_unit_vector(rng, ruv);
        orc_v_add(rec->normal, ruv, dir);
        if (orc_v_near_zero(dir)) { dir[0] = rec->normal[0]; dir[1] = rec->normal[1]; dir[2] = rec->normal[2]; }
        sd[0] = dir[0]; sd[1] = dir[1]; sd[2] = dir[2];
        att[0] = s->albedo[0]; att[1] = s->albedo[1]; att[2] = s->albedo[2];
        return 1;
    }
    case ORC_MAT_METAL: {
        double refl[3], ruv[3], fz[3];
        orc_v_reflect(rd, rec->normal, refl);
        orc_v_unit(refl, refl);
        orc_random_unit_vector(rng, ruv); /* drawn even when fuzz is 0 */
        orc_v_scale(ruv, s->fuzz_or_ior, fz);
        orc_v_add(refl, fz, sd);
        att[0] = s->albedo[0]; att[1] = s->albedo[1]; att[2] = s->albedo[2];
        return orc_v_dot(sd, rec->normal) > 0.;
    }
    case ORC_MAT_DIELECTRIC: {
        att[0] = 1.0; att[1] = 1.0; att[2] = 1.0;
        double ri = rec->front_face ? 1.0 / s->fuzz_or_ior : s->fuzz_or_ior;
        double ud[3], nud[3];
        orc_v_unit(rd, ud);
        orc_v_neg(ud, nud);
        double cos_theta = fmin(orc_v_dot(nud, rec->normal), 1.0);
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        int cannot_refract = ri * sin_theta > 1.0;
        /* short-circuit: the uniform is consumed only when refraction is possible */
        if (cannot_refract || orc_reflectance(cos_theta, ri) > orc_rng_draw(rng, 0u))
            orc_v_reflect(ud, rec->normal, sd);
        else
            orc_v_refract(ud, rec->normal, ri, sd);
        return 1;
    }
    default:
        return 0;
    }
}

/* Material::scatter as the path loops call it, with the incoming ray's origin at hand: Transparent
 * (material.rs:22-31) hands the incoming ray back unchanged -- origin included -- with attenuation (1, 1, 1); every
 * other material scatters from the hit point (orc_scatter). */
static int scatter_from(const orc_sphere* s, const double* ro, const double* rd, const orc_hit* rec, orc_rng* rng,
                        double* att, double* so, double* sd) {
    if (s->kind == ORC_MAT_TRANSPARENT || s->shape == ORC_SHAPE_BOX) { /* a box's material: bounding_box.rs:65,71 */
        att[0] = 1.0; att[1] = 1.0; att[2] = 1.0;
        so[0] = ro[0]; so[1] = ro[1]; so[2] = ro[2];
        sd[0] = rd[0]; sd[1] = rd[1]; sd[2] = rd[2];
        return 1;
    }
    return orc_scatter(s, rd, rec, rng, att, so, sd);
}

/* --------------------------------------------------------------- Camera -- */

void orc_camera_default(orc_camera* cam) { /* camera.rs:147-163 */
    memset(cam, 0, sizeof *cam);
    cam->aspect_ratio = 1.0;
    cam->image_width = 100;
    cam->samples_per_pixel = 10;
    cam->max_depth = 10;
    cam->vfov = 90.;
    cam->lookfrom[0] = 0.; cam->lookfrom[1] = 0.; cam->lookfrom[2] = 0.;
    cam->lookat[0] = 0.; cam->lookat[1] = 0.; cam->lookat[2] = -1.;
    cam->vup[0] = 0.; cam->vup[1] = 1.; cam->vup[2] = 0.;
    cam->defocus_angle = 0.;
    cam->focus_dist = 10.;
}

static double degrees_to_radians(double degrees) { return degrees * 3.14159265358979323846264338327950288 / 180.0; } /* prelude.rs:17-20 */

void orc_camera_initialize(const orc_camera* cam, orc_camera_derived* d) { /* camera.rs:169-206 */
    memset(d, 0, sizeof *d);
    d->image_height = (int32_t)((double)cam->image_width / cam->aspect_ratio);
    if (d->image_height < 1) d->image_height = 1;
    d->pixel_samples_scale = 1.0 / (double)cam->samples_per_pixel;
    memcpy(d->center, cam->lookfrom, sizeof d->center);

    double theta = degrees_to_radians(cam->vfov);
    double h = tan(theta / 2.);
    double viewport_height = 2. * h * cam->focus_dist;
    double viewport_width = viewport_height * ((double)cam->image_width / (double)d->image_height);

    double t[3];
    orc_v_sub(cam->lookfrom, cam->lookat, t);
    orc_v_unit(t, d->w);
    orc_v_cross(cam->vup, d->w, t);
    orc_v_unit(t, d->u);
    orc_v_cross(d->w, d->u, d->v);

    double viewport_u[3], viewport_v[3], nv[3];
    orc_v_scale(d->u, viewport_width, viewport_u);
    orc_v_neg(d->v, nv);
    orc_v_scale(nv, viewport_height, viewport_v);

    orc_v_div(viewport_u, (double)cam->image_width, d->pixel_delta_u);
    orc_v_div(viewport_v, (double)d->image_height, d->pixel_delta_v);

    double fw[3], hu[3], hv[3], ul[3], sum[3];
    orc_v_scale(d->w, cam->focus_dist, fw);
    orc_v_div(viewport_u, 2., hu);
    orc_v_div(viewport_v, 2., hv);
    orc_v_sub(d->center, fw, ul);
    orc_v_sub(ul, hu, ul);
    orc_v_sub(ul, hv, ul);
    orc_v_add(d->pixel_delta_u, d->pixel_delta_v, sum);
    orc_v_scale(sum, 0.5, sum);
    orc_v_add(ul, sum, d->pixel00_loc);

    double defocus_radius = cam->focus_dist * tan(degrees_to_radians(cam->defocus_angle / 2.));
    orc_v_scale(d->u, defocus_radius, d->defocus_disk_u);
    orc_v_scale(d->v, defocus_radius, d->defocus_disk_v);

    d->viewport_height = viewport_height;
    d->viewport_width = viewport_width;
    d->defocus_radius = defocus_radius;
}

/* camera.rs:238-262: offsets first, then (if the aperture is open) the disk sample. */
void orc_get_ray(const orc_camera* cam, const orc_camera_derived* d, int32_t i, int32_t j, orc_rng* rng, double* ro,
                 double* rd) {
    double ox = orc_rng_draw(rng, 0u) - 0.5;
    double oy = orc_rng_draw(rng, 1u) - 0.5;
    double a[3], b[3], ps[3];
    orc_v_scale(d->pixel_delta_u, (double)i + ox, a);
    orc_v_scale(d->pixel_delta_v, (double)j + oy, b);
    orc_v_add(d->pixel00_loc, a, ps);
    orc_v_add(ps, b, ps);
    if (cam->defocus_angle <= 0.) {
        ro[0] = d->center[0]; ro[1] = d->center[1]; ro[2] = d->center[2];
    } else {
        double p[3], pu[3], pv[3];
        orc_random_in_unit_disk(rng, p);
        orc_v_scale(d->defocus_disk_u, p[0], pu);
        orc_v_scale(d->defocus_disk_v, p[1], pv);
        orc_v_add(d->center, pu, ro);
        orc_v_add(ro, pv, ro);
    }
    orc_v_sub(ps, ro, rd);
}

static void sky(const double* rd, double* color) { /* camera.rs:279-281 */
    double ud[3];
    orc_v_unit(rd, ud);
    double a = 0.5 * (ud[1] + 1.0);
    const double white[3] = {1.0, 1.0, 1.0}, blue[3] = {0.5, 0.7, 1.0};
    double w[3], b[3];
    orc_v_scale(white, 1.0 - a, w);
    orc_v_scale(blue, a, b);
    orc_v_add(w, b, color);
}

static void ray_color_rec(const orc_sphere* s, size_t n, const double* ro, const double* rd, int32_t depth,
                          uint32_t stage, orc_rng* rng, double* color, uint64_t* segments) { /* camera.rs:264-282 */
    if (depth <= 0) { color[0] = 0.; color[1] = 0.; color[2] = 0.; return; }
    orc_hit rec;
    memset(&rec, 0, sizeof rec);
    if (segments) (*segments)++;
    if (orc_list_hit(s, n, ro, rd, 0.001, INFINITY, &rec)) {
        double att[3], so[3], sd[3];
        rng->stage = stage;
        if (scatter_from(&s[rec.index], ro, rd, &rec, rng, att, so, sd)) {
            double inner[3];
            ray_color_rec(s, n, so, sd, depth - 1, stage + 1u, rng, inner, segments);
            orc_v_mul(att, inner, color);
            return;
        }
        color[0] = 0.; color[1] = 0.; color[2] = 0.;
        return;
    }
    sky(rd, color);
}

void orc_ray_color_recursive(const orc_sphere* s, size_t n, const double* ro, const double* rd, int32_t depth,
                             orc_rng* rng, double* color, uint64_t* segments) {
    ray_color_rec(s, n, ro, rd, depth, 1u, rng, color, segments);
}

/* camera.rs:118-144 applied until finished: explicit throughput, left-associated products. */
void orc_ray_color_iterative(const orc_sphere* s, size_t n, const double* ro_in, const double* rd_in, int32_t depth,
                             orc_rng* rng, double* color, uint64_t* segments) {
    double att[3] = {1., 1., 1.};
    double ro[3] = {ro_in[0], ro_in[1], ro_in[2]}, rd[3] = {rd_in[0], rd_in[1], rd_in[2]};
    const double zero[3] = {0., 0., 0.};
    uint32_t stage = 1u;
    for (;;) {
        if (depth <= 0) { orc_v_mul(att, zero, color); return; }
        orc_hit rec;
        memset(&rec, 0, sizeof rec);
        if (segments) (*segments)++;
        if (orc_list_hit(s, n, ro, rd, 0.001, INFINITY, &rec)) {
            double a2[3], so[3], sd[3];
            rng->stage = stage;
            if (scatter_from(&s[rec.index], ro, rd, &rec, rng, a2, so, sd)) {
                orc_v_mul(att, a2, att);
                memcpy(ro, so, sizeof ro);
                memcpy(rd, sd, sizeof rd);
                depth -= 1;
                stage += 1u;
                continue;
            }
            orc_v_mul(att, zero, color);
            return;
        }
        double sk[3];
        sky(rd, sk);
        orc_v_mul(att, sk, color);
        return;
    }
}

/* One call of ray_color_iteration, camera.rs:118-144, on the resumable path state of camera.rs:15-38.
 * The scatter draws from the Philox stream (pixel, sample, e->stage); a scatter moves the entry on to the next stage. */
void orc_ray_color_iteration(orc_path_entry* e, const orc_sphere* s, size_t n, uint64_t seed, int32_t image_width,
                             uint32_t sample_offset) {
    const double zero[3] = {0., 0., 0.};
    if (e->depth <= 0) { /* camera.rs:120-123 */
        orc_v_mul(e->attenuation, zero, e->color);
        e->finished = 1u; e->hit_object_or_stop = 1u;
        return;
    }
    orc_hit rec;
    memset(&rec, 0, sizeof rec);
    if (orc_list_hit(s, n, e->origin, e->direction, 0.001, INFINITY, &rec)) { /* camera.rs:126 */
        orc_rng rng;
        memset(&rng, 0, sizeof rng);
        rng.key[0] = (uint32_t)seed;
        rng.key[1] = (uint32_t)(seed >> 32);
        rng.pixel = (uint32_t)e->pixel_j * (uint32_t)image_width + (uint32_t)e->pixel_i;
        rng.sample = sample_offset + (uint32_t)e->pixel_sample_num;
        rng.stage = e->stage;
        double a2[3], so[3], sd[3];
        if (scatter_from(&s[rec.index], e->origin, e->direction, &rec, &rng, a2, so, sd)) { /* camera.rs:129-133 */
            orc_v_mul(e->attenuation, a2, e->attenuation);
            memcpy(e->origin, so, sizeof so);
            memcpy(e->direction, sd, sizeof sd);
            e->depth -= 1;
            e->stage += 1u;
            e->finished = 0u; e->hit_object_or_stop = 1u;
            return;
        }
        orc_v_mul(e->attenuation, zero, e->color); /* camera.rs:135-136 */
        e->finished = 1u; e->hit_object_or_stop = 1u;
        return;
    }
    double sk[3];
    sky(e->direction, sk); /* camera.rs:140-142 */
    orc_v_mul(e->attenuation, sk, e->color);
    e->finished = 1u; e->hit_object_or_stop = 0u;
}

/* CameraRayIterator::next for one pixel index (camera.rs:103-114) followed by RayColorEntry::new(ray, max_depth)
 * (camera.rs:23-31). */
void orc_camera_ray_entry(const orc_camera* cam, const orc_camera_derived* d, uint64_t seed, uint32_t sample_offset,
                          int32_t i, int32_t j, int32_t sample, orc_path_entry* e) {
    orc_rng rng;
    memset(&rng, 0, sizeof rng);
    rng.key[0] = (uint32_t)seed;
    rng.key[1] = (uint32_t)(seed >> 32);
    rng.pixel = (uint32_t)j * (uint32_t)cam->image_width + (uint32_t)i;
    rng.sample = sample_offset + (uint32_t)sample;
    rng.stage = 0u;
    memset(e, 0, sizeof *e);
    orc_get_ray(cam, d, i, j, &rng, e->origin, e->direction);
    e->attenuation[0] = e->attenuation[1] = e->attenuation[2] = 1.;
    e->depth = cam->max_depth;
    e->stage = 1u;
    e->pixel_i = i; e->pixel_j = j; e->pixel_sample_num = sample;
}

/* The two functions above over a batch (test convenience). px holds (pixel_i, pixel_j, pixel_sample_num) triples. */
void orc_camera_ray_entries(const orc_camera* cam, uint64_t seed, uint32_t sample_offset, const int32_t* px, size_t n,
                            orc_path_entry* out) {
    orc_camera_derived d;
    orc_camera_initialize(cam, &d);
    for (size_t k = 0; k < n; ++k)
        orc_camera_ray_entry(cam, &d, seed, sample_offset, px[3 * k], px[3 * k + 1], px[3 * k + 2], &out[k]);
}
void orc_ray_color_iteration_batch(orc_path_entry* e, size_t n_entries, const orc_sphere* s, size_t n, uint64_t seed,
                                   int32_t image_width, uint32_t sample_offset) {
    for (size_t k = 0; k < n_entries; ++k)
        if (!e[k].finished) orc_ray_color_iteration(&e[k], s, n, seed, image_width, sample_offset);
}

/* --------------------------------------------------------------- Colour -- */

double orc_linear_to_gamma(double x) { /* colors.rs:48-55; NaN > 0 is false, so NaN maps to 0 */
    if (x > 0.) return sqrt(x);
    return 0.;
}

static uint32_t to_byte(double x) { /* colors.rs:19: `as u32` truncates toward zero and saturates */
    double v = 255.999 * orc_clamp(0.000, 0.999, x);
    if (!(v > 0.)) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

void orc_color_to_rgb(const double* c, uint32_t* rgb) { /* colors.rs:9-24 */
    rgb[0] = to_byte(orc_linear_to_gamma(c[0]));
    rgb[1] = to_byte(orc_linear_to_gamma(c[1]));
    rgb[2] = to_byte(orc_linear_to_gamma(c[2]));
}

/* colors.rs:26-46 without the window present: accumulate, mean by division, gamma, pack ARGB. */
uint32_t orc_write_color(int32_t i, int32_t j, size_t width, const double* c, uint32_t* color, double* raw,
                         int32_t* count) {
    size_t index = (size_t)j * width + (size_t)i;
    raw[3 * index + 0] += c[0];
    raw[3 * index + 1] += c[1];
    raw[3 * index + 2] += c[2];
    count[index] += 1;
    double denom = count[index] != 0 ? (double)count[index] : 1.;
    double mean[3];
    orc_v_div(&raw[3 * index], denom, mean);
    uint32_t rgb[3];
    orc_color_to_rgb(mean, rgb);
    uint32_t packed = (255u << 24) | (rgb[0] << 16) | (rgb[1] << 8) | rgb[2];
    color[index] = packed;
    return packed;
}

/* --------------------------------------------------------------- Render -- */

static void render_one(const orc_camera* cam, const orc_camera_derived* d, const orc_sphere* s, size_t n,
                       uint32_t* color, double* raw, int32_t* count, uint64_t seed, uint32_t sample, int32_t i,
                       int32_t j, int32_t iterative, uint64_t* segments) {
    orc_rng rng;
    memset(&rng, 0, sizeof rng);
    rng.kind = 0;
    rng.key[0] = (uint32_t)seed;
    rng.key[1] = (uint32_t)(seed >> 32);
    rng.pixel = (uint32_t)j * (uint32_t)cam->image_width + (uint32_t)i;
    rng.sample = sample;
    rng.stage = 0u;
    double ro[3], rd[3], c[3];
    orc_get_ray(cam, d, i, j, &rng, ro, rd);
    double pixel_color[3] = {0., 0., 0.}; /* camera.rs:223-225: 0 + sample */
    if (iterative)
        orc_ray_color_iterative(s, n, ro, rd, cam->max_depth, &rng, c, segments);
    else
        orc_ray_color_recursive(s, n, ro, rd, cam->max_depth, &rng, c, segments);
    orc_v_add(pixel_color, c, pixel_color);
    orc_write_color(i, j, (size_t)cam->image_width, pixel_color, color, raw, count);
}

int orc_render(const orc_camera* cam, const orc_sphere* s, size_t n, uint32_t* color, double* raw, int32_t* count,
               size_t buf_len, uint64_t seed, uint32_t sample_offset, int32_t iterative, int32_t row_begin,
               int32_t row_step, int32_t threads, orc_stats* stats) {
    if (!cam || (!s && n) || !color || !raw || !count) return -1;
    if (cam->image_width < 1 || row_step < 1 || row_begin < 0) return -1;
    orc_camera_derived d;
    orc_camera_initialize(cam, &d);
    if (buf_len < (size_t)cam->image_width * (size_t)d.image_height) return -1;
    uint64_t segments = 0, samples = 0;
    const int32_t W = cam->image_width, H = d.image_height, spp = cam->samples_per_pixel;
    if (threads <= 1) {
        /* the reference's order: sample-major, then rows, then columns (camera.rs:217-219) */
        for (int32_t sample = 0; sample < spp; ++sample)
            for (int32_t j = row_begin; j < H; j += row_step)
                for (int32_t i = 0; i < W; ++i) {
                    render_one(cam, &d, s, n, color, raw, count, seed, sample_offset + (uint32_t)sample, i, j,
                               iterative, &segments);
                    samples++;
                }
    } else {
        /* Pixels are independent and each pixel still sees its samples in order, so results are identical. */
        int32_t nrows = (H - row_begin + row_step - 1) / row_step;
        if (nrows < 0) nrows = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : segments, samples)
#endif
        for (int32_t r = 0; r < nrows; ++r) {
            int32_t j = row_begin + r * row_step;
            for (int32_t i = 0; i < W; ++i)
                for (int32_t sample = 0; sample < spp; ++sample) {
                    render_one(cam, &d, s, n, color, raw, count, seed, sample_offset + (uint32_t)sample, i, j,
                               iterative, &segments);
                    samples++;
                }
        }
    }
    if (stats) { stats->segments = segments; stats->samples = samples; }
    return 0;
}

/* --------------------------------------------------------------- Scenes -- */

static void set_sphere(orc_sphere* o, double x, double y, double z, double r, uint32_t kind, double a0, double a1,
                       double a2, double f) {
    memset(o, 0, sizeof *o);
    o->center[0] = x; o->center[1] = y; o->center[2] = z;
    o->radius = fmax(r, 0.); /* sphere.rs:14 */
    o->kind = kind;
    o->albedo[0] = a0; o->albedo[1] = a1; o->albedo[2] = a2;
    o->fuzz_or_ior = (kind == ORC_MAT_METAL) ? (f < 1.0 ? f : 1.0) : f; /* material.rs:67 */
}

/* The small test scene: ground plus one Lambertian, one dielectric and one metal sphere. */
int orc_scene_three_spheres(orc_sphere* out, size_t cap) {
    if (cap < 4) return -1;
    set_sphere(&out[0], 0., -100.5, -1., 100., ORC_MAT_LAMBERTIAN, 0.8, 0.8, 0.0, 0.);
    set_sphere(&out[1], 0., 0., -1.2, 0.5, ORC_MAT_LAMBERTIAN, 0.1, 0.2, 0.5, 0.);
    set_sphere(&out[2], -1., 0., -1., 0.5, ORC_MAT_DIELECTRIC, 0., 0., 0., 1.5);
    set_sphere(&out[3], 1., 0., -1., 0.5, ORC_MAT_METAL, 0.8, 0.6, 0.2, 1.0);
    return 4;
}

/* main.rs:13-50 with the uniforms taken from one Philox stream (pixel = sample = stage = 0xFFFFFFFF), generalised:
 * cells a, b in [lo, hi), material thresholds p_diffuse and p_diffuse + p_metal, metal fuzz in [0, fuzz_max).
 * The reference's scene is lo = -11, hi = 11, 0.8, 0.15, 0.5. */
int orc_scene_recipe(uint64_t seed, int32_t lo, int32_t hi, double p_diffuse, double p_metal, double fuzz_max,
                     orc_sphere* out, size_t cap) {
    orc_rng rng;
    memset(&rng, 0, sizeof rng);
    rng.key[0] = (uint32_t)seed;
    rng.key[1] = (uint32_t)(seed >> 32);
    rng.pixel = rng.sample = rng.stage = 0xFFFFFFFFu;
    uint32_t k = 0;
    size_t n = 0;
    if (cap < 1) return -1;
    set_sphere(&out[n++], 0., -1000., 0., 1000., ORC_MAT_LAMBERTIAN, 0.5, 0.5, 0.5, 0.);
    for (int a = lo; a < hi; ++a)
        for (int b = lo; b < hi; ++b) {
            double choose_mat = orc_rng_draw(&rng, k++);
            double cx = (double)a + 0.9 * orc_rng_draw(&rng, k++);
            double cz = (double)b + 0.9 * orc_rng_draw(&rng, k++);
            double c[3] = {cx, 0.2, cz}, ref[3] = {4., 0.2, 0.}, dlt[3];
            orc_v_sub(c, ref, dlt);
            if (orc_v_length(dlt) > 0.9) {
                if (n >= cap) return -1;
                if (choose_mat < p_diffuse) {
                    double r1[3], r2[3], al[3];
                    for (int q = 0; q < 3; ++q) r1[q] = orc_rng_draw(&rng, k++);
                    for (int q = 0; q < 3; ++q) r2[q] = orc_rng_draw(&rng, k++);
                    orc_v_mul(r1, r2, al);
                    set_sphere(&out[n++], cx, 0.2, cz, 0.2, ORC_MAT_LAMBERTIAN, al[0], al[1], al[2], 0.);
                } else if (choose_mat < p_diffuse + p_metal) {
                    double al[3];
                    for (int q = 0; q < 3; ++q) al[q] = rng_range(&rng, k++, 0.5, 1.);
                    double fuzz = rng_range(&rng, k++, 0., fuzz_max);
                    set_sphere(&out[n++], cx, 0.2, cz, 0.2, ORC_MAT_METAL, al[0], al[1], al[2], fuzz);
                } else {
                    set_sphere(&out[n++], cx, 0.2, cz, 0.2, ORC_MAT_DIELECTRIC, 0., 0., 0., 1.5);
                }
            }
        }
    if (n + 3 > cap) return -1;
    set_sphere(&out[n++], 0., 1., 0., 1.0, ORC_MAT_DIELECTRIC, 0., 0., 0., 1.5);
    set_sphere(&out[n++], -4., 1., 0., 1.0, ORC_MAT_LAMBERTIAN, 0.4, 0.2, 0.1, 0.);
    set_sphere(&out[n++], 4., 1., 0., 1.0, ORC_MAT_METAL, 0.7, 0.6, 0.5, 0.0);
    return (int)n;
}

int orc_scene_final(uint64_t seed, orc_sphere* out, size_t cap) {
    return orc_scene_recipe(seed, -11, 11, 0.8, 0.15, 0.5, out, cap);
}
