// Links libmi355sat.so.  MI355SAT_LIB_DIR points at the directory holding it (default: the in-tree build
// output, timberborn_support_solver_amd/, produced by `make -C timberborn_support_solver_amd/csrc`).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("MI355SAT_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../timberborn_support_solver_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mi355sat");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MI355SAT_LIB_DIR");
}
