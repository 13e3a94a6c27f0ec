/*
 * dray_oracle.h -- CPU restatement (plain C, f64) of the reference path tracer's
 * per-pixel render path.  TEST INFRASTRUCTURE ONLY: it is the checker for the
 * HIP path, never the thing shipped or measured.  Only tests/, the smoke check
 * and the bench's cpu_baseline leg may load it.
 *
 * PARITY STATUS: the reference is Rust and cannot be compiled in this build
 * environment (no cargo/rustc, un-vendored crates), and its own tests pin only
 * the Vec3 arithmetic (src/raytracer/vec3.rs:260-450).  Those 24 vectors are
 * checked against this file; everything after Vec3 (sphere hit, scatter,
 * camera, colour, whole renders) is "parity unpinned": it follows the cited
 * reference lines operation by operation and is checked against analytic
 * known-answer values derived from those formulas, not against reference output.
 *
 * The reference draws randomness from an OS-seeded thread-local generator
 * (src/raytracer/prelude.rs:22-25), which cannot be reproduced.  Here every
 * uniform comes from a counter-based Philox4x32-10 stream keyed on
 * (pixel, sample, stage, draw), or from a caller-supplied script of uniforms
 * for known-answer tests.
 */
#ifndef DRAY_ORACLE_H
#define DRAY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order and meaning as the public fields of the reference camera
 * (src/raytracer/camera.rs:41-54). */
typedef struct orc_camera {
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
} orc_camera;

/* Private (derived) camera fields, src/raytracer/camera.rs:55-65. */
typedef struct orc_camera_derived {
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
    double viewport_height, viewport_width, defocus_radius; /* locals of initialize, exposed for tests */
} orc_camera_derived;

enum { ORC_MAT_DEFAULT = 0, ORC_MAT_LAMBERTIAN = 1, ORC_MAT_METAL = 2, ORC_MAT_DIELECTRIC = 3, ORC_MAT_TRANSPARENT = 4 };

/* One sphere with its material flattened in (src/raytracer/sphere.rs:6-10,
 * src/raytracer/material.rs:33-36,59-63,83-86). */
typedef struct orc_sphere {
    double center[3];
    double radius;
    uint32_t kind;
    uint32_t shape; /* ORC_SHAPE_SPHERE, or ORC_SHAPE_BOX: the record is an orc_box */
    double albedo[3];
    double fuzz_or_ior;
} orc_sphere;

/* A list entry that is a BoundingBox (src/raytracer/bounding_box.rs:6-9: three Intervals) instead of a Sphere:
 * same size as orc_sphere, `shape` at the same place.  Its material is always Transparent (bounding_box.rs:65,71). */
enum { ORC_SHAPE_SPHERE = 0, ORC_SHAPE_BOX = 1 };
typedef struct orc_box {
    double min[3]; /* axes[a].min */
    double _unused0;
    uint32_t _unused_kind;
    uint32_t shape; /* ORC_SHAPE_BOX */
    double max[3]; /* axes[a].max */
    double _unused1;
} orc_box;

/* src/raytracer/hittable.rs:7-13, with the material replaced by the index of
 * the sphere that was hit. */
typedef struct orc_hit {
    double p[3];
    double normal[3];
    double t;
    int32_t front_face;
    int32_t index;
} orc_hit;

/* Random source.  kind 0: Philox stream (key, pixel, sample, stage);
 * kind 1: scripted uniforms consumed in order. */
typedef struct orc_rng {
    int32_t kind;
    uint32_t key[2];
    uint32_t pixel, sample, stage;
    const double* script;
    size_t script_len, script_pos;
    uint64_t draws; /* uniforms handed out so far */
} orc_rng;

typedef struct orc_stats {
    uint64_t segments; /* world-hit calls (src/raytracer/camera.rs:270) */
    uint64_t samples;  /* get_ray + ray_color pairs (camera.rs:224-225) */
} orc_stats;

/* Philox4x32-10 block function and the uniform derived from one word. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_u01(uint32_t word);
double orc_rng_draw(orc_rng* rng, uint32_t draw_index);

/* Vec3 (src/raytracer/vec3.rs). */
void orc_v_add(const double* a, const double* b, double* o);
void orc_v_sub(const double* a, const double* b, double* o);
void orc_v_mul(const double* a, const double* b, double* o);
void orc_v_scale(const double* a, double s, double* o);
void orc_v_div(const double* a, double s, double* o);
void orc_v_divv(const double* a, const double* b, double* o);
void orc_v_neg(const double* a, double* o);
void orc_v_min(const double* a, const double* b, double* o);
void orc_v_max(const double* a, const double* b, double* o);
double orc_v_dot(const double* a, const double* b);
void orc_v_cross(const double* a, const double* b, double* o);
double orc_v_length_squared(const double* a);
double orc_v_length(const double* a);
void orc_v_unit(const double* a, double* o);
int orc_v_near_zero(const double* a);
void orc_v_reflect(const double* v, const double* n, double* o);
void orc_v_refract(const double* uv, const double* n, double eta, double* o);
/* Draw layout inside one (pixel, sample, stage) stream: stage 0 is get_ray -- draws 0,1 are the pixel
 * offsets, disk attempt m uses draws 2+2m and 3+2m; stage k >= 1 is the scatter after the k-th hit --
 * unit-vector attempt j uses draws 4j..4j+2, the dielectric uniform is draw 0. */
void orc_random_unit_vector(orc_rng* rng, double* o);
void orc_random_in_unit_disk(orc_rng* rng, double* o);

/* Interval (src/raytracer/interval.rs:28-36). */
int orc_surrounds(double lo, double hi, double x);
double orc_clamp(double lo, double hi, double x);

/* Geometry and materials. */
int orc_sphere_hit(const orc_sphere* s, const double* ro, const double* rd, double tmin, double tmax, orc_hit* rec);
/* BoundingBox::hit, bounding_box.rs:50-76, as written: only axes 0 and 1 are looked at, and a plane's distance is
 * `min - origin / direction`.  Sets t and p of rec only (the reference leaves the other fields as they were). */
int orc_box_hit(const orc_box* b, const double* ro, const double* rd, double tmin, double tmax, orc_hit* rec);
int orc_list_hit(const orc_sphere* s, size_t n, const double* ro, const double* rd, double tmin, double tmax, orc_hit* rec);
int orc_scatter(const orc_sphere* s, const double* rd, const orc_hit* rec, orc_rng* rng, double* attenuation,
                double* so, double* sd);
double orc_reflectance(double cosine, double ri);

/* Camera. */
void orc_camera_default(orc_camera* cam);
void orc_camera_initialize(const orc_camera* cam, orc_camera_derived* d);
void orc_get_ray(const orc_camera* cam, const orc_camera_derived* d, int32_t i, int32_t j, orc_rng* rng, double* ro,
                 double* rd);
void orc_ray_color_recursive(const orc_sphere* s, size_t n, const double* ro, const double* rd, int32_t depth,
                             orc_rng* rng, double* color, uint64_t* segments);
void orc_ray_color_iterative(const orc_sphere* s, size_t n, const double* ro, const double* rd, int32_t depth,
                             orc_rng* rng, double* color, uint64_t* segments);

/* RayColorEntry + RayColorStatus (camera.rs:15-38) with the pixel index (camera.rs:8-13) and the Philox stage that
 * stand in for the reference's global generator.  Same layout as dray_path_entry. */
typedef struct orc_path_entry {
    double attenuation[3];
    double origin[3], direction[3];
    double color[3];
    int32_t depth;
    uint32_t stage;
    int32_t pixel_i, pixel_j, pixel_sample_num;
    uint32_t finished, hit_object_or_stop;
    uint32_t _pad;
} orc_path_entry;
void orc_ray_color_iteration(orc_path_entry* e, const orc_sphere* s, size_t n, uint64_t seed, int32_t image_width,
                             uint32_t sample_offset);
void orc_camera_ray_entry(const orc_camera* cam, const orc_camera_derived* d, uint64_t seed, uint32_t sample_offset,
                          int32_t i, int32_t j, int32_t sample, orc_path_entry* e);
void orc_camera_ray_entries(const orc_camera* cam, uint64_t seed, uint32_t sample_offset, const int32_t* px, size_t n,
                            orc_path_entry* out);
void orc_ray_color_iteration_batch(orc_path_entry* e, size_t n_entries, const orc_sphere* s, size_t n, uint64_t seed,
                                   int32_t image_width, uint32_t sample_offset);

/* Colour output (src/raytracer/colors.rs). */
double orc_linear_to_gamma(double x);
void orc_color_to_rgb(const double* c, uint32_t* rgb);
uint32_t orc_write_color(int32_t i, int32_t j, size_t width, const double* c, uint32_t* color, double* raw,
                         int32_t* count);

/* Whole render (src/raytracer/camera.rs:208-236).
 *  iterative   0: recursive ray_color (right-associated products), 1: iterative twin
 *  row_begin/row_step: only rows j with j >= row_begin and (j - row_begin) % row_step == 0 are rendered
 *  threads     1: the reference's sample-major single-threaded order; >1: rows in parallel (same results)
 * Returns 0, or -1 for bad arguments / too-small buffers. */
int orc_render(const orc_camera* cam, const orc_sphere* s, size_t n, uint32_t* color, double* raw, int32_t* count,
               size_t buf_len, uint64_t seed, uint32_t sample_offset, int32_t iterative, int32_t row_begin,
               int32_t row_step, int32_t threads, orc_stats* stats);

/* Scene recipes.  final: src/main.rs:13-50 with a seeded stream; returns the sphere count (<= cap) or -1. */
int orc_scene_three_spheres(orc_sphere* out, size_t cap);
int orc_scene_final(uint64_t seed, orc_sphere* out, size_t cap);
/* The same recipe over cells a, b in [lo, hi) with other material shares (final = -11, 11, 0.8, 0.15, 0.5). */
int orc_scene_recipe(uint64_t seed, int32_t lo, int32_t hi, double p_diffuse, double p_metal, double fuzz_max,
                     orc_sphere* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
