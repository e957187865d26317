// build.rs for the reference crate (antoinedesbois/Ray-Tracer-Rust) once src/rtx_ffi.rs is added: links librtx.so.
// RTX_LIB_DIR = the directory that holds librtx.so (ray-tracer-rust_amd/ of this repository after
// `make -C ray-tracer-rust_amd/csrc`).  Not compiled in this repository's build image (no Rust toolchain there);
// tests/test_ffi_layout.py keeps the binding next to it in step with include/rtx.h without one.
use std::env;

fn main() {
    let dir = env::var("RTX_LIB_DIR").expect("set RTX_LIB_DIR to the directory holding librtx.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
}
