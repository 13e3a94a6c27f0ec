// Reference-order render kernels for gfx950: every arithmetic step is written in
// the operation order of the Rust reference so that the f64 instantiation can
// be compared with the CPU oracle bit for bit.  Build with -ffp-contract=off.
//
// One thread owns one pixel and walks its samples in order (the reference's
// sample-major loop, src/raytracer/camera.rs:217-219, visits a pixel's samples
// in the same order; pixels do not interact).
#pragma clang fp contract(off)

#include "ref_device.h"

namespace dray {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void render_ref_kernel(CamDev<T> cam, const T* __restrict__ geom,
                                                         const T* __restrict__ mat, const uint32_t* __restrict__ kind,
                                                         int32_t n, RefGrid rg, Partition part, RngKey key, uint32_t flags,
                                                         uint32_t* __restrict__ color, double* __restrict__ raw,
                                                         int32_t* __restrict__ count, DeviceCounters* counters) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)part.n_rows * cam.W;
    unsigned long long my_segments = 0, my_samples = 0;
    if (tid < total) {
        const int32_t r = (int32_t)(tid / cam.W), i = (int32_t)(tid % cam.W);
        const int32_t j = part.row_begin + r * part.row_step;
        const size_t out = (size_t)(part.compact ? r : j) * (size_t)cam.W + (size_t)i;
        const bool iterative = (flags & DRAY_FLAG_ITERATIVE_PRODUCT) != 0;

        double acc[3] = {0., 0., 0.};
        if (raw) { acc[0] = raw[3 * out + 0]; acc[1] = raw[3 * out + 1]; acc[2] = raw[3 * out + 2]; }

        uint32_t stack[kExactMaxDepth]; // spheres whose albedo multiplies the sample, outermost first

        // One loop over the path segments of all the pixel's samples, in sample order: a lane whose sample has ended
        // starts its next one at once.  (Written as a loop over samples around a loop over segments, the lanes of a
        // wave would meet again at every sample and wait there for the longest path among them.)
        Stream st = {key.k0, key.k1, (uint32_t)j * (uint32_t)cam.W + (uint32_t)i, key.sample_offset, 0u};
        V3<T> ro = {(T)0., (T)0., (T)0.}, rd = {(T)0., (T)0., (T)1.};
        V3<T> att = {(T)1., (T)1., (T)1.}; // camera.rs:26 (iterative form)
        int32_t s = 0, depth = cam.max_depth, sp = 0;
        bool live = cam.spp > 0;
        if (live) get_ray(cam, i, j, st, ro, rd);
        while (live) {
            V3<T> c = {(T)0., (T)0., (T)0.};
            bool ended = true;
            if (depth > 0) { // else camera.rs:265-267 / :120-123: the sample is black
                Hit<T> rec;
                my_segments++;
                if (list_hit<T>(geom, n, rg, ro, rd, rec)) { // camera.rs:270
                    st.stage = (uint32_t)(cam.max_depth - depth + 1);
                    V3<T> so = ro, sd;
                    if (scatter<T>(kind[rec.index], mat + 4 * rec.index, rd, rec, st, so, sd)) {
                        const uint32_t k = kind[rec.index];
                        const bool white = k == DRAY_MAT_DIELECTRIC || k == DRAY_MAT_TRANSPARENT; // attenuation (1, 1, 1)
                        if (iterative) {
                            V3<T> al = white ? V3<T>{(T)1., (T)1., (T)1.}
                                             : V3<T>{mat[4 * rec.index], mat[4 * rec.index + 1], mat[4 * rec.index + 2]};
                            att = vmul(att, al); // camera.rs:130
                        } else if (!white) {
                            stack[sp++] = (uint32_t)rec.index; // a product with (1,1,1) is exact, so those are skipped
                        }
                        ro = so;
                        rd = sd;
                        depth -= 1;
                        ended = false;
                    } // else camera.rs:276: absorbed, black
                } else {
                    V3<T> ud = vunit(rd); // camera.rs:279-281
                    T a = (T)0.5 * (ud.y + (T)1.0);
                    c = vadd(vscale(V3<T>{(T)1.0, (T)1.0, (T)1.0}, (T)1.0 - a), vscale(V3<T>{(T)0.5, (T)0.7, (T)1.0}, a));
                }
            }
            if (ended) {
                if (iterative) {
                    c = vmul(att, c); // camera.rs:121,135,142
                } else {
                    while (sp > 0) { // camera.rs:274, innermost product first
                        const uint32_t q = stack[--sp];
                        c = vmul(V3<T>{mat[4 * q], mat[4 * q + 1], mat[4 * q + 2]}, c);
                    }
                }
                // camera.rs:223-225 (0 + sample) then colors.rs:38
                acc[0] += (double)((T)0. + c.x);
                acc[1] += (double)((T)0. + c.y);
                acc[2] += (double)((T)0. + c.z);
                my_samples++;
                if (++s < cam.spp) { // camera.rs:217: the pixel's next sample
                    st.sample = key.sample_offset + (uint32_t)s;
                    st.stage = 0u;
                    get_ray(cam, i, j, st, ro, rd);
                    att = {(T)1., (T)1., (T)1.};
                    depth = cam.max_depth;
                    sp = 0;
                } else {
                    live = false;
                }
            }
        }

        int32_t cnt = (count ? count[out] : 0) + cam.spp; // colors.rs:39
        const double denom = cnt != 0 ? (double)cnt : 1.; // colors.rs:40
        color[out] = pack_argb(acc[0] / denom, acc[1] / denom, acc[2] / denom);
        if (raw) { raw[3 * out + 0] = acc[0]; raw[3 * out + 1] = acc[1]; raw[3 * out + 2] = acc[2]; }
        if (count) count[out] = cnt;
    }
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        my_segments += __shfl_down(my_segments, off, 64);
        my_samples += __shfl_down(my_samples, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && counters) {
        atomicAdd(&counters->segments, my_segments);
        atomicAdd(&counters->samples, my_samples);
    }
}

// CameraRayIterator::next (camera.rs:103-114) + RayColorEntry::new (camera.rs:23-31) for a list of pixel indices.
__global__ __launch_bounds__(256) void camera_rays_kernel(CamDev<double> cam, RngKey key,
                                                          const dray_pixel_index* __restrict__ px, size_t n,
                                                          dray_path_entry* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const dray_pixel_index id = px[k];
    const Stream st = {key.k0, key.k1, (uint32_t)id.pixel_j * (uint32_t)cam.W + (uint32_t)id.pixel_i,
                       key.sample_offset + (uint32_t)id.pixel_sample_num, 0u};
    V3<double> ro, rd;
    get_ray(cam, id.pixel_i, id.pixel_j, st, ro, rd);
    dray_path_entry e = {};
    e.attenuation[0] = e.attenuation[1] = e.attenuation[2] = 1.;
    e.origin[0] = ro.x; e.origin[1] = ro.y; e.origin[2] = ro.z;
    e.direction[0] = rd.x; e.direction[1] = rd.y; e.direction[2] = rd.z;
    e.depth = cam.max_depth;
    e.stage = 1u;
    e.px = id;
    out[k] = e;
}

// ray_color_iteration, camera.rs:118-144: one segment of every unfinished path of the batch.
__global__ __launch_bounds__(256) void path_step_kernel(const double* __restrict__ geom, const double* __restrict__ mat,
                                                        const uint32_t* __restrict__ kind, int32_t n_spheres, RefGrid rg,
                                                        RngKey key, int32_t W, dray_path_entry* __restrict__ entries, size_t n,
                                                        DeviceCounters* counters) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long my_segments = 0;
    if (k < n && !entries[k].finished) {
        dray_path_entry e = entries[k];
        V3<double> att = {e.attenuation[0], e.attenuation[1], e.attenuation[2]};
        const V3<double> ro = {e.origin[0], e.origin[1], e.origin[2]}, rd = {e.direction[0], e.direction[1], e.direction[2]};
        const V3<double> zero = {0., 0., 0.};
        V3<double> c = zero;
        bool finished = true, hit_or_stop = true;
        if (e.depth <= 0) { // camera.rs:120-123
            c = vmul(att, zero);
        } else {
            Hit<double> rec;
            my_segments = 1;
            if (list_hit<double>(geom, n_spheres, rg, ro, rd, rec)) { // camera.rs:126
                const Stream st = {key.k0, key.k1, (uint32_t)e.px.pixel_j * (uint32_t)W + (uint32_t)e.px.pixel_i,
                                   key.sample_offset + (uint32_t)e.px.pixel_sample_num, e.stage};
                V3<double> so = ro, sd;
                const uint32_t mk = kind[rec.index];
                if (scatter<double>(mk, mat + 4 * rec.index, rd, rec, st, so, sd)) { // camera.rs:129-133
                    const V3<double> al = (mk == DRAY_MAT_DIELECTRIC || mk == DRAY_MAT_TRANSPARENT)
                                              ? V3<double>{1., 1., 1.}
                                              : V3<double>{mat[4 * rec.index], mat[4 * rec.index + 1], mat[4 * rec.index + 2]};
                    att = vmul(att, al);
                    e.origin[0] = so.x; e.origin[1] = so.y; e.origin[2] = so.z;
                    e.direction[0] = sd.x; e.direction[1] = sd.y; e.direction[2] = sd.z;
                    e.depth -= 1;
                    e.stage += 1u;
                    finished = false;
                } else {
                    c = vmul(att, zero); // camera.rs:135
                }
            } else { // camera.rs:140-143
                const V3<double> ud = vunit(rd);
                const double t = 0.5 * (ud.y + 1.0);
                c = vmul(att, vadd(vscale(V3<double>{1.0, 1.0, 1.0}, 1.0 - t), vscale(V3<double>{0.5, 0.7, 1.0}, t)));
                hit_or_stop = false;
            }
        }
        e.attenuation[0] = att.x; e.attenuation[1] = att.y; e.attenuation[2] = att.z;
        if (finished) { e.color[0] = c.x; e.color[1] = c.y; e.color[2] = c.z; }
        e.finished = finished ? 1u : 0u;
        e.hit_object_or_stop = hit_or_stop ? 1u : 0u;
        entries[k] = e;
    }
    for (int off = 32; off > 0; off >>= 1) my_segments += __shfl_down(my_segments, off, 64);
    if ((threadIdx.x & 63) == 0 && counters && my_segments) atomicAdd(&counters->segments, my_segments);
}

template <typename T>
hipError_t launch_ref(const CamDev<T>& cam, const T* geom, const T* mat, const SceneDev& scene, const Partition& part,
                      const RngKey& key, uint32_t flags, int closest_hit, uint32_t* d_color, double* d_raw, int32_t* d_count,
                      DeviceCounters* d_counters, hipStream_t stream) {
    const long long total = (long long)part.n_rows * cam.W;
    if (total <= 0) return hipSuccess;
    const int block = 256;
    const long long grid = (total + block - 1) / block;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    const RefGrid rg = ref_grid_of(scene, closest_hit != kRefWholeList);
    hipLaunchKernelGGL(render_ref_kernel<T>, dim3((unsigned)grid), dim3(block), 0, stream, cam, geom, mat, scene.kind,
                       scene.n, rg, part, key, flags, d_color, d_raw, d_count, d_counters);
    return hipGetLastError();
}

} // namespace

hipError_t launch_render_ref_f64(const CamDev<double>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, int closest_hit, uint32_t* d_color, double* d_raw,
                                 int32_t* d_count, DeviceCounters* d_counters, hipStream_t stream) {
    return launch_ref<double>(cam, scene.geom64, scene.mat64, scene, part, key, flags, closest_hit, d_color, d_raw, d_count,
                              d_counters, stream);
}

hipError_t launch_render_ref_f32(const CamDev<float>& cam, const SceneDev& scene, const Partition& part,
                                 const RngKey& key, uint32_t flags, uint32_t* d_color, double* d_raw, int32_t* d_count,
                                 DeviceCounters* d_counters, hipStream_t stream) {
    return launch_ref<float>(cam, scene.geom32, scene.mat32, scene, part, key, flags, kRefWholeList, d_color, d_raw, d_count,
                             d_counters, stream);
}

hipError_t launch_camera_rays(const CamDev<double>& cam, const RngKey& key, const dray_pixel_index* d_px, size_t n,
                              dray_path_entry* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t grid = (n + 255) / 256;
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(camera_rays_kernel, dim3((unsigned)grid), dim3(256), 0, stream, cam, key, d_px, n, d_out);
    return hipGetLastError();
}

hipError_t launch_path_step(const SceneDev& scene, const RngKey& key, int32_t image_width, bool use_grid,
                            dray_path_entry* d_entries, size_t n, DeviceCounters* d_counters, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t grid = (n + 255) / 256;
    if (grid > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(path_step_kernel, dim3((unsigned)grid), dim3(256), 0, stream, scene.geom64, scene.mat64,
                       scene.kind, scene.n, ref_grid_of(scene, use_grid), key, image_width, d_entries, n, d_counters);
    return hipGetLastError();
}

} // namespace dray