// Locates libzkstark_hip.so: ZKSTARK_LIB_DIR (the directory holding the .so built by `python -m zk_evm_amd.build`,
// i.e. <repo>/zk_evm_amd) or, failing that, the in-tree location relative to this crate.  The library itself links the
// HIP runtime (/opt/rocm/lib); nothing else is needed at link time.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("ZKSTARK_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../zk_evm_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=zkstark_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ZKSTARK_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/zkstark.h");
}
