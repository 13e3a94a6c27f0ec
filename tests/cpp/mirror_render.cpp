// Renders a small world through the C++ mirror of the reference's scene / camera interface (include/dray.hpp)
// and writes the image as PPM plus the f64 sums as a raw file, for the tests to compare with the Python
// binding's render of the same world.  The world is built the way the reference's main builds its own
// (main.rs:13-50: HittableList::new, world.add(Arc::new(Sphere::new(center, radius, material)))).
//
//   mirror_render <mode 0|1|2> <seed> <width> <spp> <depth> <out.ppm> [out.raw]
//
// Exit code: 0, or the negated dray_status of the failing call (4 = no GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "dray.hpp"

int main(int argc, char** argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s mode seed width spp depth out.ppm [out.raw]\n", argv[0]);
        return 64;
    }
    using namespace dray;
    HittableList world;
    world.add(std::make_shared<Sphere>(Point3(0.0, -100.5, -1.0), 100.0, std::make_shared<Lambertian>(Color3(0.8, 0.8, 0.0))));
    world.add(std::make_shared<Sphere>(Point3(0.0, 0.0, -1.2), 0.5, std::make_shared<Lambertian>(Color3(0.1, 0.2, 0.5))));
    world.add(std::make_shared<Sphere>(Point3(-1.0, 0.0, -1.0), 0.5, std::make_shared<Dialectric>(1.5)));
    world.add(std::make_shared<Sphere>(Point3(1.0, 0.0, -1.0), 0.5, std::make_shared<Metal>(Color3(0.8, 0.6, 0.2), 3.0))); // fuzz clamps to 1
    world.add(std::make_shared<Sphere>(Point3(0.0, 1.2, -1.5), -2.0, std::make_shared<DefaultMaterial>()));              // radius clamps to 0
    world.add(std::make_shared<Sphere>(Point3(0.6, -0.3, -0.7), 0.15, std::make_shared<Transparent>()));
    world.add(std::make_shared<BoundingBox>(0.3, 0.6, 0.2, 0.9, -1.0, 1.0)); // bounding_box.rs: new_xyz

    Camera cam = Camera::create();
    cam.aspect_ratio = 16.0 / 9.0;
    cam.opts.mode = (uint32_t)std::atoi(argv[1]);
    cam.opts.seed = std::strtoull(argv[2], nullptr, 10);
    cam.image_width = std::atoi(argv[3]);
    cam.samples_per_pixel = std::atoi(argv[4]);
    cam.max_depth = std::atoi(argv[5]);
    try {
        const int32_t W = cam.image_width, H = cam.image_height();
        // buffers sized as the reference's caller sizes them (main.rs:69-73)
        std::vector<uint32_t> color((size_t)W * W, 0u);
        std::vector<double> raw((size_t)W * W * 3, 0.0);
        std::vector<int32_t> count((size_t)W * W, 0);
        int presents = 0;
        cam.render(world, [&](const std::vector<uint32_t>&, int32_t w, int32_t h) { presents += (w == W && h == H); },
                   color, raw, count);
        if (presents != 1) return 70;
        const dray_stats first = cam.last_stats;
        write_ppm(argv[6], color, W, H, false);
        if (argc > 7) {
            std::FILE* f = std::fopen(argv[7], "wb");
            if (!f || std::fwrite(raw.data(), sizeof(double), (size_t)W * H * 3, f) != (size_t)W * H * 3) return 74;
            std::fclose(f);
        }
        // the same frame again in passes of 3 samples: one present per pass, the same buffers, the same totals
        if (cam.samples_per_pixel > 3) {
            std::vector<uint32_t> color3((size_t)W * W, 0u);
            std::vector<double> raw3((size_t)W * W * 3, 0.0);
            std::vector<int32_t> count3((size_t)W * W, 0);
            presents = 0;
            cam.samples_per_pass = 3;
            cam.render(world, [&](const std::vector<uint32_t>&, int32_t, int32_t) { ++presents; }, color3, raw3, count3);
            cam.samples_per_pass = 0;
            if (presents != (cam.samples_per_pixel + 2) / 3) return 75;
            if (color3 != color || raw3 != raw || count3 != count) return 76;
            if (cam.last_stats.segments != first.segments || cam.last_stats.samples != first.samples) return 77;
        }
        // the same frame again as three shares of interleaved scanlines on the one GPU (dray_render_multi)
        {
            std::vector<uint32_t> color4((size_t)W * W, 0u);
            std::vector<double> raw4((size_t)W * W * 3, 0.0);
            std::vector<int32_t> count4((size_t)W * W, 0);
            cam.devices = {0, 0, 0};
            cam.render(world, color4, raw4, count4);
            cam.devices.clear();
            if (color4 != color || raw4 != raw || count4 != count) return 80;
            if (cam.last_stats.segments != first.segments || cam.last_stats.rows_rendered != H) return 81;
        }
        // the same frame again through a render context on two shares, in passes of 3 samples with the caller's buffers
        // registered: scene and grid built once, sums resident on the GPU between passes, only colours moving
        {
            std::vector<uint32_t> color5((size_t)W * W, 0u);
            std::vector<double> raw5((size_t)W * W * 3, 0.0);
            std::vector<int32_t> count5((size_t)W * W, 0);
            RenderContext ctx({0, 0});
            (void)ctx.pin(color5); (void)ctx.pin(raw5); (void)ctx.pin(count5); // refused by some drivers: then staged
            cam.context = &ctx;
            cam.samples_per_pass = 3;
            presents = 0;
            cam.render(world, [&](const std::vector<uint32_t>&, int32_t, int32_t) { ++presents; }, color5, raw5, count5);
            cam.samples_per_pass = 0;
            cam.context = nullptr;
            ctx.unpin();
            if (presents != (cam.samples_per_pixel + 2) / 3 && cam.samples_per_pixel > 3) return 82;
            if (color5 != color || raw5 != raw || count5 != count) return 83;
            if (cam.last_stats.segments != first.segments || cam.last_stats.samples != first.samples) return 84;
        }
        // the same frame again through the resumable path state: rays for every (pixel, sample), stepped until all
        // are finished, colours summed per pixel in sample order (left-to-right product, camera.rs:118-144)
        unsigned long long step_segments = 0;
        double worst = 0.0;
        if (cam.opts.mode == DRAY_MODE_F64) {
            cam.opts.flags |= DRAY_FLAG_ITERATIVE_PRODUCT;
            std::vector<uint32_t> color2((size_t)W * W, 0u);
            std::vector<double> raw2((size_t)W * W * 3, 0.0);
            std::vector<int32_t> count2((size_t)W * W, 0);
            cam.render(world, color2, raw2, count2);
            std::vector<PixelIndexEntry> px;
            for (int32_t s = 0; s < cam.samples_per_pixel; ++s)
                for (int32_t j = 0; j < H; ++j)
                    for (int32_t i = 0; i < W; ++i) px.push_back(PixelIndexEntry{i, j, s});
            DeviceScene scene(world);
            std::vector<RayColorEntry> entries = camera_rays(cam, px);
            // Camera::iterate_rays: every pixel once, sample 0, shuffled; each item equals the entry asked for in order
            {
                const auto it = iterate_rays(cam, 7);
                if (it.first.size() != (size_t)W * H || it.second.size() != it.first.size()) return 78;
                std::vector<char> seen((size_t)W * H, 0);
                bool identity = true;
                for (size_t k = 0; k < it.first.size(); ++k) {
                    const PixelIndexEntry& q = it.first[k];
                    if (q.pixel_sample_num != 0 || q.pixel_i < 0 || q.pixel_i >= W || q.pixel_j < 0 || q.pixel_j >= H) return 78;
                    const size_t at = (size_t)q.pixel_j * W + q.pixel_i; // = its place in the ordered first pass of px
                    if (seen[at]++) return 78;
                    identity = identity && at == k;
                    if (std::memcmp(&it.second[k], &entries[at], sizeof(RayColorEntry)) != 0) return 79;
                }
                if (identity) return 78;
            }
            for (int guard = 0; guard <= cam.max_depth + 1; ++guard) {
                const uint64_t moved = ray_color_iteration(entries, scene, cam);
                step_segments += moved;
                if (moved == 0) break;
            }
            std::vector<double> sum((size_t)W * H * 3, 0.0);
            for (const RayColorEntry& e : entries) {
                if (!e.finished) return 71;
                for (int q = 0; q < 3; ++q) sum[3 * ((size_t)e.px.pixel_j * W + e.px.pixel_i) + q] += 0.0 + e.color[q];
            }
            for (size_t k = 0; k < sum.size(); ++k) {
                const double d = sum[k] > raw2[k] ? sum[k] - raw2[k] : raw2[k] - sum[k];
                if (d > worst) worst = d;
            }
            if (worst != 0.0 || step_segments != cam.last_stats.segments) return 72;
            cam.opts.flags &= ~(uint32_t)DRAY_FLAG_ITERATIVE_PRODUCT;
        }
        std::printf("{\"width\": %d, \"height\": %d, \"segments\": %llu, \"samples\": %llu, \"spheres\": %d, "
                    "\"step_segments\": %llu}\n", W, H, (unsigned long long)first.segments,
                    (unsigned long long)first.samples, first.n_spheres, step_segments);
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return -e.status;
    }
    return 0;
}
