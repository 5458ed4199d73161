// Link against libibu_hip.so built by `make -C ibu_amd/csrc` (path via IBU_HIP_LIB_DIR).
fn main() {
    if let Ok(dir) = std::env::var("IBU_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=ibu_hip");
}
