// Where libnova_mi355x.so lives: NOVA_MI355X_LIB_DIR (the repo's nova_amd/ directory after `python -c "import __graft_entry__ as g; g.build()"`),
// and the HIP runtime it depends on under ROCM_PATH (default /opt/rocm).  Same role as blitzar-sys' build script behind
// src/provider/blitzar.rs:7-40: locate a prebuilt shared library, emit the link lines, nothing is compiled here.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=NOVA_MI355X_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    if let Ok(dir) = env::var("NOVA_MI355X_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=nova_mi355x");
}
