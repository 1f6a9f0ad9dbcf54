// Link against libwrs_mi355x.so.  WRS_LIB_DIR = directory that holds it (the repo's whisper_rs_amd/ after `make`).
fn main() {
    let dir = std::env::var("WRS_LIB_DIR").unwrap_or_else(|_| "../../whisper_rs_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=wrs_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=WRS_LIB_DIR");
}
