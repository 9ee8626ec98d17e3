// Where libligero_hip.so lives: LIGERO_HIP_LIB_DIR, else <this repository>/ligero_amd/lib (built by `make -C ligero_amd/csrc`).
// At run time the library needs ROCm's libamdhip64.so on the loader path; an rpath to the directory found here is added so that
// `cargo test` of the reference finds libligero_hip.so itself.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("LIGERO_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ligero_amd/lib")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ligero_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=LIGERO_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=src/lib.rs");
}
