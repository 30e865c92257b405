// Links the prebuilt HIP engine instead of compiling CUDA: the reference's build.rs drives nvcc over cuda/*.cu
// (P1A 6block/build.rs:66-95); here the kernels are already in libmi355msm.so (built by `python -c "import
// __graft_entry__ as g; g.build()"` with hipcc --offload-arch=gfx950) and this script only tells rustc where it is.
//
//   MI355_MSM_LIB_DIR   directory holding libmi355msm.so and the harness-named shims (default: ../2022-entries_amd)
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MI355_MSM_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../2022-entries_amd"));
    let dir = dir.canonicalize().expect("MI355_MSM_LIB_DIR does not exist: build the HIP library first");
    assert!(dir.join("libmi355msm.so").exists(), "libmi355msm.so not found in {}", dir.display());

    // the harness-named entry points (mult_pippenger_init / mult_pippenger_inf) live in a per-curve shim object,
    // exactly one of which is linked -- the reference selects the curve the same way, with a cargo feature
    let shim = if cfg!(feature = "bls12_381") { "mi355msm_zprize_381" } else { "mi355msm_zprize_377" };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib={}", shim);
    println!("cargo:rustc-link-lib=dylib=mi355msm");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MI355_MSM_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
