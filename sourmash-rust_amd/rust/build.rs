// Link against the in-tree shared object (make -C ../csrc builds it).
fn main() {
    let dir = std::env::var("SOURMASH_AMD_LIB_DIR").unwrap_or_else(|_| "../lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=sourmash_amd");
}
