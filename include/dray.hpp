// dray.hpp -- C++ host-side mirror of the reference's scene / camera interface for the render path, over the
// C ABI of dray.h.  Header-only; link with libdray_hip.so.
//
// Names, argument meaning and failure behaviour follow src/raytracer of the reference (Rust):
//   Vec3 / Point3 / Color3            vec3.rs:5-23 (three f64, constructor order x, y, z)
//   Lambertian / Metal / Dialectric   material.rs:39, :66-70 (fuzz clamped to 1), :89 (spelling as in the reference)
//   DefaultMaterial                   material.rs:10-19 (absorbs every ray)
//   Transparent                       material.rs:22-31 (hands the incoming ray back)
//   Sphere::new                       sphere.rs:13-19 (radius clamped at 0)
//   HittableList new/add/clear/len    hittable_list.rs:11-40 (insertion order is the tie-break order, :65-70)
//   Camera: ten public fields, new()  camera.rs:41-54, :147-163
//   Camera::render                    camera.rs:208-236, returning io::Result<()> there; here it throws dray::Error
// Only render does work: it flattens the world and calls dray_render.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dray.h"

namespace dray {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what + ": " + dray_strerror(st)), status(st) {}
};

inline void check(int status, const char* what) {
    if (status < 0) throw Error(status, what);
}

struct Vec3 {
    double x = 0., y = 0., z = 0.;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};
using Point3 = Vec3;
using Color3 = Vec3;

// ---- materials ---------------------------------------------------------------------------------------------
class Material {
  public:
    virtual ~Material() = default;
    // kind, albedo and parameter as dray_sphere carries them
    virtual void flatten(dray_sphere& out) const = 0;
};

class DefaultMaterial : public Material {
  public:
    void flatten(dray_sphere& out) const override { out.kind = DRAY_MAT_DEFAULT; }
};

// material.rs:22-31: scatters the incoming ray unchanged with attenuation (1, 1, 1)
class Transparent : public Material {
  public:
    void flatten(dray_sphere& out) const override { out.kind = DRAY_MAT_TRANSPARENT; }
};

class Lambertian : public Material {
  public:
    explicit Lambertian(Color3 albedo) : albedo_(albedo) {}
    void flatten(dray_sphere& out) const override {
        out.kind = DRAY_MAT_LAMBERTIAN;
        out.albedo[0] = albedo_.x; out.albedo[1] = albedo_.y; out.albedo[2] = albedo_.z;
    }

  private:
    Color3 albedo_;
};

class Metal : public Material {
  public:
    Metal(Color3 albedo, double fuzz) : albedo_(albedo), fuzz_(fuzz < 1.0 ? fuzz : 1.0) {}
    void flatten(dray_sphere& out) const override {
        out.kind = DRAY_MAT_METAL;
        out.albedo[0] = albedo_.x; out.albedo[1] = albedo_.y; out.albedo[2] = albedo_.z;
        out.fuzz_or_ior = fuzz_;
    }

  private:
    Color3 albedo_;
    double fuzz_;
};

class Dialectric : public Material {
  public:
    explicit Dialectric(double refraction_index) : refraction_index_(refraction_index) {}
    void flatten(dray_sphere& out) const override {
        out.kind = DRAY_MAT_DIELECTRIC;
        out.fuzz_or_ior = refraction_index_;
    }

  private:
    double refraction_index_;
};

// ---- hittables ---------------------------------------------------------------------------------------------
class Hittable {
  public:
    virtual ~Hittable() = default;
    // appends this object's spheres in the order HittableList::hit would visit them
    virtual void flatten(std::vector<dray_sphere>& out) const = 0;
};

class Sphere : public Hittable {
  public:
    Sphere(Point3 center, double radius, std::shared_ptr<Material> mat)
        : center_(center), radius_(radius > 0. ? radius : 0.), mat_(std::move(mat)) {}
    Point3 center() const { return center_; }
    double radius() const { return radius_; }
    void flatten(std::vector<dray_sphere>& out) const override {
        dray_sphere s{};
        s.center[0] = center_.x; s.center[1] = center_.y; s.center[2] = center_.z;
        s.radius = radius_;
        if (!mat_) throw Error(DRAY_ERR_UNSUPPORTED, "sphere without a material");
        mat_->flatten(s);
        out.push_back(s);
    }

  private:
    Point3 center_;
    double radius_;
    std::shared_ptr<Material> mat_;
};

// bounding_box.rs:6-28: three intervals.  As a hittable it is intersected the way bounding_box.rs:50-76 is written
// (axes 0 and 1 only) and its material is always Transparent.
class BoundingBox : public Hittable {
  public:
    BoundingBox() // BoundingBox::default: three empty intervals (interval.rs: EMPTY = (+inf, -inf))
        : BoundingBox(INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY) {}
    BoundingBox(double x_min, double x_max, double y_min, double y_max, double z_min, double z_max) // new_xyz
        : min_{x_min, y_min, z_min}, max_{x_max, y_max, z_max} {}
    void flatten(std::vector<dray_sphere>& out) const override {
        dray_box b{};
        for (int a = 0; a < 3; ++a) { b.min[a] = min_[a]; b.max[a] = max_[a]; }
        b.shape = DRAY_SHAPE_BOX;
        dray_sphere s;
        static_assert(sizeof(dray_box) == sizeof(dray_sphere), "a box record takes a sphere's place in the list");
        std::memcpy(&s, &b, sizeof s);
        out.push_back(s);
    }

  private:
    double min_[3], max_[3];
};

class HittableList : public Hittable {
  public:
    HittableList() = default;
    explicit HittableList(std::shared_ptr<Hittable> object) { add(std::move(object)); }
    void clear() { objects.clear(); }
    void add(std::shared_ptr<Hittable> object) { objects.push_back(std::move(object)); }
    size_t len() const { return objects.size(); }
    void flatten(std::vector<dray_sphere>& out) const override {
        for (const auto& o : objects) {
            if (!o) throw Error(DRAY_ERR_UNSUPPORTED, "null hittable");
            o->flatten(out);
        }
    }
    std::vector<std::shared_ptr<Hittable>> objects;
};

// ---- render context ----------------------------------------------------------------------------------------
// A dray_context: the scenes, device buffers and staging memory of the one-call route, kept between calls.  Share k of
// `devices` renders scanlines k, k + G, ...; an empty list means every visible GPU.
class RenderContext {
  public:
    explicit RenderContext(const std::vector<int32_t>& devices = {}) {
        const int rc = dray_context_create(devices.empty() ? nullptr : devices.data(), (int32_t)devices.size(), &ctx_);
        if (rc != DRAY_OK) throw Error(rc, std::string("RenderContext: ") + dray_strerror(rc));
    }
    ~RenderContext() { dray_context_destroy(ctx_); }
    RenderContext(const RenderContext&) = delete;
    RenderContext& operator=(const RenderContext&) = delete;
    dray_context* handle() const { return ctx_; }
    // registers the buffer with the HIP runtime: copies then go straight to and from it.  It must outlive unpin().
    // Returns false where the driver refuses such registrations (the render then stages through the context's block).
    template <typename T>
    bool pin(std::vector<T>& buffer) {
        return dray_context_pin(ctx_, buffer.data(), buffer.size() * sizeof(T)) == DRAY_OK;
    }
    // zeroed page-locked memory of the context for `n` elements, valid until the context goes
    template <typename T>
    T* host_alloc(size_t n) {
        void* p = nullptr;
        const int rc = dray_context_host_alloc(ctx_, n * sizeof(T), &p);
        if (rc != DRAY_OK) throw Error(rc, std::string("RenderContext::host_alloc: ") + dray_strerror(rc));
        return static_cast<T*>(p);
    }
    void unpin() { dray_context_unpin(ctx_); }

  private:
    dray_context* ctx_ = nullptr;
};

// ---- camera ------------------------------------------------------------------------------------------------
class Camera {
  public:
    double aspect_ratio = 1.0;
    int32_t image_width = 100;
    int32_t samples_per_pixel = 10;
    int32_t max_depth = 10;
    double vfov = 90.0;
    Point3 lookfrom{0., 0., 0.};
    Point3 lookat{0., 0., -1.};
    Vec3 vup{0., 1., 0.};
    double defocus_angle = 0.0;
    double focus_dist = 10.0;

    // backend options, not in the reference: see dray_opts
    dray_opts opts;
    // > 0: render in passes of this many samples per pixel and call `present` after each pass; the buffers end up
    // the same as after one pass of samples_per_pixel
    int32_t samples_per_pass = 0;
    // GPUs of this node to spread a render over from this one process (dray_render_multi); empty: the one device of
    // `opts`.  A device may appear more than once.
    std::vector<int32_t> devices;
    // a context to render through (its devices then count, not `devices`): the world is uploaded and its grid built
    // when it changes, not per call, and the passes of a progressive render leave the running sums on the device
    RenderContext* context = nullptr;
    dray_stats last_stats{}; // of the whole call (counts and times summed over its passes)

    Camera() { dray_opts_default(&opts); }
    static Camera create() { return Camera(); } // Camera::new

    dray_camera pod() const {
        dray_camera c{};
        c.aspect_ratio = aspect_ratio;
        c.image_width = image_width;
        c.samples_per_pixel = samples_per_pixel;
        c.max_depth = max_depth;
        c.vfov = vfov;
        c.lookfrom[0] = lookfrom.x; c.lookfrom[1] = lookfrom.y; c.lookfrom[2] = lookfrom.z;
        c.lookat[0] = lookat.x; c.lookat[1] = lookat.y; c.lookat[2] = lookat.z;
        c.vup[0] = vup.x; c.vup[1] = vup.y; c.vup[2] = vup.z;
        c.defocus_angle = defocus_angle;
        c.focus_dist = focus_dist;
        return c;
    }

    // Camera::initialize (camera.rs:169-206); also what a caller needs to size its buffers
    dray_camera_derived initialize() const {
        const dray_camera c = pod();
        dray_camera_derived d{};
        check(dray_camera_initialize(&c, &d), "Camera::initialize");
        return d;
    }
    int32_t image_height() const { return initialize().image_height; }

    // Camera::render.  The buffers hold at least image_width * image_height pixels (the reference's caller sizes
    // them image_width^2, main.rs:69-73) and are added to, never cleared (colors.rs:38-39).  `present`, if
    // given, is called with the colour buffer after every pass (one pass unless samples_per_pass is set) where
    // the reference refreshes its window after every pixel sample (colors.rs:44).
    template <typename Present>
    void render(const Hittable& world, Present&& present, std::vector<uint32_t>& color_buffer,
                std::vector<double>& raw_buffer, std::vector<int32_t>& count_buffer) {
        std::vector<dray_sphere> spheres;
        world.flatten(spheres);
        dray_camera c = pod();
        size_t buf_len = color_buffer.size();
        if (!raw_buffer.empty() && raw_buffer.size() / 3 < buf_len) buf_len = raw_buffer.size() / 3;
        if (!count_buffer.empty() && count_buffer.size() < buf_len) buf_len = count_buffer.size();
        const int32_t total = samples_per_pixel;
        const int32_t per_pass = samples_per_pass > 0 && samples_per_pass < total ? samples_per_pass : total;
        dray_opts o = opts;
        dray_stats whole{};
        int32_t done = 0;
        const dray_sphere* sp = spheres.empty() ? nullptr : spheres.data();
        if (context) check(dray_context_set_scene(context->handle(), sp, spheres.size()), "Camera::render"); // a no-op for the list it holds
        if (context && per_pass < total) {
            // the passes of one call, two under way at a time: the sums go up before the first and come back after
            // the last, and `present` runs while the next pass renders
            struct Show {
                Present* present;
                std::vector<uint32_t>* color;
                int32_t width, height;
                std::exception_ptr failed;
                static void call(void* user, int32_t, int32_t) noexcept {
                    Show* s = (Show*)user;
                    if (s->failed) return;
                    try {
                        (*s->present)(*s->color, s->width, s->height);
                    } catch (...) { // must not unwind through the C frames: thrown again below
                        s->failed = std::current_exception();
                    }
                }
            } show{&present, &color_buffer, image_width, image_height(), nullptr};
            dray_stats st{};
            check(dray_context_render_progressive(context->handle(), &c, color_buffer.data(),
                                                  raw_buffer.empty() ? nullptr : raw_buffer.data(),
                                                  count_buffer.empty() ? nullptr : count_buffer.data(), buf_len, &o, per_pass,
                                                  &Show::call, &show, &st),
                  "Camera::render");
            if (show.failed) std::rethrow_exception(show.failed);
            last_stats = st;
            return;
        }
        do { // at least one call, so that a camera of 0 samples fails (or not) the way one call does
            c.samples_per_pixel = total > 0 && total - done < per_pass ? total - done : per_pass;
            o.sample_offset = opts.sample_offset + (uint32_t)done;
            dray_stats st{};
            double* rawp = raw_buffer.empty() ? nullptr : raw_buffer.data();
            int32_t* cntp = count_buffer.empty() ? nullptr : count_buffer.data();
            if (context) {
                // the sums go up with the first pass and come back with the last; in between only the colours move
                const bool last = total <= 0 || done + c.samples_per_pixel >= total;
                const uint32_t sync = (done == 0 ? (uint32_t)DRAY_SYNC_UPLOAD_SUMS : 0u) | (last ? (uint32_t)DRAY_SYNC_DOWNLOAD_SUMS : 0u);
                check(dray_context_render(context->handle(), &c, color_buffer.data(), rawp, cntp, buf_len, &o, sync, &st),
                      "Camera::render");
            } else {
                check(devices.empty() ? dray_render(&c, sp, spheres.size(), color_buffer.data(), rawp, cntp, buf_len, &o, &st)
                                      : dray_render_multi(&c, sp, spheres.size(), color_buffer.data(), rawp, cntp, buf_len, &o,
                                                          devices.data(), (int32_t)devices.size(), &st),
                      "Camera::render");
            }
            if (done == 0) {
                whole = st;
            } else {
                whole.segments += st.segments; whole.samples += st.samples; whole.walk_steps += st.walk_steps;
                whole.kernel_ms += st.kernel_ms; whole.resolve_ms += st.resolve_ms; whole.total_ms += st.total_ms;
            }
            present(color_buffer, image_width, st.image_height);
            done += c.samples_per_pixel;
        } while (done < total);
        last_stats = whole;
    }
    void render(const Hittable& world, std::vector<uint32_t>& color_buffer, std::vector<double>& raw_buffer,
                std::vector<int32_t>& count_buffer) {
        render(world, [](const std::vector<uint32_t>&, int32_t, int32_t) {}, color_buffer, raw_buffer, count_buffer);
    }
};

// ---- resumable path state (what the reference's distributed layer feeds on) ---------------------------------
// A HittableList uploaded to one GPU (dray_scene).
class DeviceScene {
  public:
    explicit DeviceScene(const Hittable& world, int32_t device = -1) {
        std::vector<dray_sphere> spheres;
        world.flatten(spheres);
        check(dray_scene_create(spheres.empty() ? nullptr : spheres.data(), spheres.size(), device, &scene_),
              "dray_scene_create");
    }
    ~DeviceScene() { dray_scene_destroy(scene_); }
    DeviceScene(const DeviceScene&) = delete;
    DeviceScene& operator=(const DeviceScene&) = delete;
    dray_scene* handle() const { return scene_; }

  private:
    dray_scene* scene_ = nullptr;
};

// Camera::render with the world already uploaded and the three buffers in that GPU's memory (device pointers,
// buf_len pixels each; d_raw and d_count may be null): nothing crosses the bus.  Enqueues on `stream` (a
// hipStream_t, null = the default stream); blocks only when `stats` is asked for.
inline void render_device(const Camera& cam, const DeviceScene& world, uint32_t* d_color, double* d_raw, int32_t* d_count,
                          size_t buf_len, void* stream = nullptr, dray_stats* stats = nullptr) {
    const dray_camera c = cam.pod();
    check(dray_render_device(&c, world.handle(), d_color, d_raw, d_count, buf_len, &cam.opts, stream, stats),
          "render_device");
}

// The same, for renders enqueued one behind the other without a wait (the passes of a progressive render): returns the
// call's ticket (0: nothing to render); render_collect waits for that call and returns its statistics.  A scene
// remembers its last 16 calls.
inline uint64_t render_device_async(const Camera& cam, const DeviceScene& world, uint32_t* d_color, double* d_raw,
                                    int32_t* d_count, size_t buf_len, void* stream = nullptr) {
    const dray_camera c = cam.pod();
    uint64_t ticket = 0;
    check(dray_render_device_async(&c, world.handle(), d_color, d_raw, d_count, buf_len, &cam.opts, stream, &ticket),
          "render_device_async");
    return ticket;
}
inline dray_stats render_collect(const DeviceScene& world, uint64_t ticket) {
    dray_stats st{};
    check(dray_scene_collect(world.handle(), ticket, &st), "render_collect");
    return st;
}

using PixelIndexEntry = dray_pixel_index; // camera.rs:8-13
using RayColorEntry = dray_path_entry;    // camera.rs:15-38 (entry and status in one record)

// Camera::iterate_rays (camera.rs:165) for the given pixel indices, in the given order: each yields
// RayColorEntry::new(get_ray(i, j), max_depth).
inline std::vector<RayColorEntry> camera_rays(const Camera& cam, const std::vector<PixelIndexEntry>& px) {
    std::vector<RayColorEntry> out(px.size());
    const dray_camera c = cam.pod();
    check(dray_camera_rays(&c, px.data(), px.size(), out.data(), &cam.opts), "camera_rays");
    return out;
}

// Camera::iterate_rays (camera.rs:165, CameraRayIterator :68-116): every pixel of the image once, sample number 0
// (the iterator stops after image_width * image_height items, :93-95), in a shuffled order (:96-98; the reference
// shuffles with an OS-seeded generator, here with std::mt19937_64 seeded by `shuffle_seed`).  Item k of the
// reference's iterator is (first[k], the ray of second[k]); second[k] is already RayColorEntry::new(ray, max_depth).
inline std::pair<std::vector<PixelIndexEntry>, std::vector<RayColorEntry>> iterate_rays(const Camera& cam,
                                                                                        uint64_t shuffle_seed = 0) {
    const int32_t W = cam.image_width, H = cam.image_height();
    std::vector<int32_t> seq((size_t)W * (size_t)H);
    for (size_t k = 0; k < seq.size(); ++k) seq[k] = (int32_t)k;
    std::mt19937_64 gen(shuffle_seed);
    std::shuffle(seq.begin(), seq.end(), gen);
    std::vector<PixelIndexEntry> px(seq.size());
    for (size_t k = 0; k < seq.size(); ++k) px[k] = PixelIndexEntry{seq[k] % W, seq[k] / W, 0}; // camera.rs:103-106
    std::vector<RayColorEntry> entries = camera_rays(cam, px);
    return {std::move(px), std::move(entries)};
}

// ray_color_iteration (camera.rs:118-144) on every unfinished entry, in place.  Returns the number intersected.
inline uint64_t ray_color_iteration(std::vector<RayColorEntry>& entries, const DeviceScene& world, const Camera& cam) {
    dray_stats st{};
    check(dray_path_step(world.handle(), cam.image_width, entries.data(), entries.size(), &cam.opts, &st),
          "ray_color_iteration");
    return st.segments;
}

inline void write_ppm(const std::string& path, const std::vector<uint32_t>& color, int32_t width, int32_t height,
                      bool binary = false) {
    if (color.size() < (size_t)width * (size_t)height) throw Error(DRAY_ERR_BUFFER_TOO_SMALL, "write_ppm");
    check(dray_write_ppm(path.c_str(), color.data(), width, height, binary ? 1 : 0), "write_ppm");
}

} // namespace dray
