// Links libdray_hip.so (built by `make -C distributed-raytracer_amd/csrc`, hipcc --offload-arch=gfx950).
// DRAY_HIP_LIB_DIR names the directory that holds it; the HIP runtime it needs is found through ROCM_PATH.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=DRAY_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    let dir = env::var("DRAY_HIP_LIB_DIR")
        .expect("set DRAY_HIP_LIB_DIR to the directory holding libdray_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dray_hip");
    // so that the binary finds the library where it was linked from
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={rocm}/lib");
}
