// build.rs — locate libzkhal_mi355x.so (built by `python -m zeth_amd.build` with hipcc --offload-arch=gfx950) and link it.
// ZKHAL_LIB_DIR names the directory that holds the library; the default is this repository's zeth_amd/ next to rust/.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("ZKHAL_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../zeth_amd")
    });
    let lib = dir.join("libzkhal_mi355x.so");
    if !lib.exists() {
        panic!("{} not found: build it with `python -m zeth_amd.build` or set ZKHAL_LIB_DIR", lib.display());
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=zkhal_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ZKHAL_LIB_DIR");
    println!("cargo:rerun-if-changed={}", lib.display());
}
