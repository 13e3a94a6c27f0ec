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
        write_ppm(argv[6], color, W, H, false);
        if (argc > 7) {
            std::FILE* f = std::fopen(argv[7], "wb");
            if (!f || std::fwrite(raw.data(), sizeof(double), (size_t)W * H * 3, f) != (size_t)W * H * 3) return 74;
            std::fclose(f);
        }
        std::printf("{\"width\": %d, \"height\": %d, \"segments\": %llu, \"samples\": %llu, \"spheres\": %d}\n", W, H,
                    (unsigned long long)cam.last_stats.segments, (unsigned long long)cam.last_stats.samples,
                    cam.last_stats.n_spheres);
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return -e.status;
    }
    return 0;
}
