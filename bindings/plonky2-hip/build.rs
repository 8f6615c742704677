// Points rustc at libp2aes.so: P2AES_LIB_DIR, or the in-tree build directory of this repository.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("P2AES_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../plonky2-aes_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=p2aes");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=P2AES_LIB_DIR");
}
