// Tells rustc where libark_hip.so lives: ARK_HIP_LIB_DIR=<this repository>/algebra_amd
fn main() {
    let dir = std::env::var("ARK_HIP_LIB_DIR").unwrap_or_else(|_| "../../algebra_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ark_hip");
    println!("cargo:rerun-if-env-changed=ARK_HIP_LIB_DIR");
}
