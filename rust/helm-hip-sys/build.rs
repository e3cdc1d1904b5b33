// Links the in-tree libraries built by `make -C helm_amd/csrc` (gfx950 only).
// HELM_AMD_ROOT = checkout of this repository.
fn main() {
    let root = std::env::var("HELM_AMD_ROOT").expect("set HELM_AMD_ROOT to the helm_amd checkout");
    println!("cargo:rustc-link-search=native={}/helm_amd/csrc", root);
    println!("cargo:rustc-link-lib=dylib=helm_hip");
    println!("cargo:rustc-link-lib=dylib=helm_host");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/helm_amd/csrc", root);
    println!("cargo:rerun-if-env-changed=HELM_AMD_ROOT");
}
