// Link against libthreecrate_hip.so.  THREECRATE_HIP_LIB_DIR points at the directory that holds it
// (built with `make -C threecrate_amd/csrc`); ROCm's libamdhip64 is found through the .so's own NEEDED entry.
fn main() {
    if let Ok(dir) = std::env::var("THREECRATE_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=threecrate_hip");
    println!("cargo:rerun-if-env-changed=THREECRATE_HIP_LIB_DIR");
}
