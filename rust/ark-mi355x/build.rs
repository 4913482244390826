// Link against libark355.so.  ARK355_LIB_DIR = directory that holds the library (default: the in-tree build,
// <repo>/snark_amd).  The library itself is built by `python -m snark_amd.build` (hipcc, gfx950); it links librccl.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("ARK355_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../snark_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ark355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ARK355_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/ark355.h");
}
