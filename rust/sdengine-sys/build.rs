// UNCOMPILED (no Rust toolchain in the build image).  Links the in-tree shared library; SDENGINE_LIB_DIR overrides its directory.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("SDENGINE_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../diffusers_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=sdengine");
    println!("cargo:rerun-if-env-changed=SDENGINE_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/sdengine.h");
}
