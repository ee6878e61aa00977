// build.rs -- compiles the HIP sources of ../whisper_amd/csrc for gfx950 into libwhisper_gfx950.so and links it.
// The source list is the Makefile's SRC (whisper_amd/csrc/Makefile); tests/test_shim_matches_header.py compares the two.
use std::path::PathBuf;
use std::process::Command;

const SRC: &[&str] = &[
    "api.hip", "comm.hip", "count64.hip", "grouptest.hip", "expand.hip", "meas.hip", "chain.hip", "vdpf.hip", "prove.hip",
    "ntt.hip", "shard.hip", "sort.hip", "client.cpp", "hh_client.cpp",
];

fn main() {
    let root = PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
    let csrc = root.join("..").join("whisper_amd").join("csrc");
    let out = PathBuf::from(std::env::var("OUT_DIR").unwrap());
    let hipcc = std::env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let arch = std::env::var("WH_OFFLOAD_ARCH").unwrap_or_else(|_| "gfx950".into());
    // a prebuilt library (make -C whisper_amd/csrc) is taken as it is
    if let Ok(dir) = std::env::var("WHISPER_GFX950_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    } else {
        let lib = out.join("libwhisper_gfx950.so");
        let status = Command::new(&hipcc)
            .arg(format!("--offload-arch={arch}"))
            .args(["-O3", "-std=c++17", "-fPIC", "-shared", "-o"])
            .arg(&lib)
            .args(SRC.iter().map(|f| csrc.join(f)))
            .args(["-lpthread", "-ldl"]) // RCCL is bound at run time (dlopen), not linked
            .status()
            .expect("hipcc not found: set HIPCC or WHISPER_GFX950_LIB_DIR");
        assert!(status.success(), "hipcc failed");
        println!("cargo:rustc-link-search=native={}", out.display());
    }
    println!("cargo:rustc-link-lib=dylib=whisper_gfx950");
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", root.join("..").join("include").join("whisper_gfx950.h").display());
    println!("cargo:rerun-if-env-changed=WHISPER_GFX950_LIB_DIR");
}
