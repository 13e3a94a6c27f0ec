"""Regenerates the golden fixtures in this directory from the CPU oracle.

The Rust reference cannot be built or run in the build environment, so these vectors are outputs
of oracle/dray_oracle.c (parity unpinned beyond Vec3, see its header), committed so that GPU runs
and later rounds can detect any drift of either side.  Run: python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402


def main():
    L = O.lib()
    # (i) 3-sphere scene, 64x36, 4 spp, depth 10
    sp = O.scene_three_spheres()
    cam = O.camera_three_spheres(width=64, spp=4, depth=10)
    col, raw, cnt, st = O.render(cam, sp, seed=42)
    np.savez_compressed(os.path.join(HERE, "render_three_spheres_64x36.npz"), spheres=sp, color=col, raw=raw,
                        count=cnt, seed=42, segments=st.segments, samples=st.samples)
    # (ii) final scene for seed 1 (sphere list as text + render 96x54, 8 spp, depth 50)
    sp = O.scene_final(1)
    with open(os.path.join(HERE, "scene_final_seed1.txt"), "w") as f:
        f.write("# cx cy cz radius kind albedo_r albedo_g albedo_b fuzz_or_ior (kind: 1 lambertian, 2 metal, 3 dielectric)\n")
        for s in sp:
            f.write("%r %r %r %r %d %r %r %r %r\n" % (float(s["center"][0]), float(s["center"][1]), float(s["center"][2]),
                                                    float(s["radius"]), int(s["kind"]), float(s["albedo"][0]),
                                                    float(s["albedo"][1]), float(s["albedo"][2]), float(s["fuzz_or_ior"])))
    cam = O.camera_final(width=96, spp=8, depth=50)
    col, raw, cnt, st = O.render(cam, sp, seed=7, threads=8)
    np.savez_compressed(os.path.join(HERE, "render_final_96x54.npz"), spheres=sp, color=col, raw=raw, count=cnt,
                        seed=7, segments=st.segments, samples=st.samples)
    # (iii) per-function vectors: sphere hit and the three scatters with scripted uniforms
    rows = []
    rng = np.random.default_rng(123)
    spheres = [O.make_sphere((0, 0, -1), 0.5, O.MAT_LAMBERTIAN, (0.1, 0.2, 0.5)),
               O.make_sphere((0.3, -0.2, -2), 0.7, O.MAT_METAL, (0.8, 0.6, 0.2), 0.3),
               O.make_sphere((-0.4, 0.1, -1.5), 0.6, O.MAT_DIELECTRIC, (0, 0, 0), 1.5)]
    for s in spheres:
        for _ in range(40):
            ro = rng.uniform(-0.3, 0.3, 3)
            rd = np.array(s.center) - ro + rng.uniform(-0.6, 0.6, 3)
            rec = O.Hit()
            hit = L.orc_sphere_hit(C.byref(s), O.v3(*ro), O.v3(*rd), 0.001, float("inf"), C.byref(rec))
            u = rng.uniform(0, 1, 12)
            r = O.script_rng(list(u))
            att, so, sd = O.D3(), O.D3(), O.D3()
            sc = L.orc_scatter(C.byref(s), O.v3(*rd), C.byref(rec), C.byref(r), att, so, sd) if hit else 0
            rows.append([s.kind, s.radius, *s.center, *s.albedo, s.fuzz_or_ior, *ro, *rd, hit, rec.t, *rec.p, *rec.normal,
                         rec.front_face, *u, sc, *att, *sd, r.script_pos])
    np.save(os.path.join(HERE, "hit_scatter_vectors.npy"), np.array(rows, dtype=np.float64))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
