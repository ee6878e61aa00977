"""The Rust shim (shim/) cannot be compiled in this image (no cargo / rustc), so it is checked as text: every entry point
include/whisper_gfx950.h declares must have an `extern "C"` item in shim/src/ffi.rs with the same name, arity, integer widths and
pointer constness; the structs and constants must agree; build.rs must compile the Makefile's sources; and everything the safe
wrappers call must be declared.  A crate built from these files then links against exactly what the library exports."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "whisper_gfx950.h")
FFI = os.path.join(ROOT, "shim", "src", "ffi.rs")
LIBRS = os.path.join(ROOT, "shim", "src", "lib.rs")

C_BASE = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "void": "c_void", "char": "c_char",
          "wh_ctx": "WhCtx", "wh_prio3_params": "WhPrio3Params", "wh_prio3_sizes": "WhPrio3Sizes"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_type_to_rust(c):
    """'const uint8_t *const *' -> '*const *const u8' (array parameters decay to pointers)."""
    toks = re.findall(r"\w+|\*", c)
    stars = toks.count("*")
    base = [t for t in toks if t not in ("const", "*")]
    assert len(base) == 1, c
    rust = C_BASE[base[0]]
    # constness per pointer level, innermost first: the qualifier left of the first '*' applies to the pointee of level 1, a
    # 'const' between stars to the level before it
    quals, const_seen = [], False
    for t in toks:
        if t == "const":
            const_seen = True
        elif t == "*":
            quals.append(const_seen)
            const_seen = False
    assert len(quals) == stars
    for q in quals:
        rust = ("*const " if q else "*mut ") + rust
    return rust


def header_prototypes():
    src = strip_c_comments(open(HEADER).read())
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+(?:\s*\*+)?)\s*(wh_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                m = re.match(r"^(.*?)(\w+)\s*(\[\s*\w*\s*\])?$", a.strip())
                ctype = m.group(1).strip() + (" *" if m.group(3) else "")
                params.append(c_type_to_rust(ctype))
        ret = " ".join(ret.split())
        protos[name] = (params, None if ret == "void" else c_type_to_rust(ret))
    return protos


def ffi_items():
    src = re.sub(r"//.*", "", open(FFI).read())
    block = re.search(r'extern\s+"C"\s*\{(.*)\}', src, flags=re.S).group(1)
    items = {}
    for name, args, ret in re.findall(r"pub\s+fn\s+(wh_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            ty = a.split(":", 1)[1].strip()
            params.append(" ".join(ty.replace("std::os::raw::", "").split()))
        items[name] = (params, None if not ret else " ".join(ret.replace("std::os::raw::", "").split()))
    return items


def test_every_header_entry_point_is_declared_with_the_same_types():
    protos, items = header_prototypes(), ffi_items()
    assert len(protos) >= 70                                   # the parser found the header's entry points
    assert sorted(protos) == sorted(items), (sorted(set(protos) - set(items)), sorted(set(items) - set(protos)))
    for name, (params, ret) in protos.items():
        assert items[name][0] == params, (name, items[name][0], params)
        assert items[name][1] == ret, (name, items[name][1], ret)


def test_header_matches_what_the_library_exports():
    lib = os.path.join(ROOT, "whisper_amd", "libwhisper_gfx950.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T wh_" in l)
    assert exported == sorted(header_prototypes())


def test_structs_and_constants_agree():
    h = strip_c_comments(open(HEADER).read())
    rs = open(FFI).read()
    for cname, rname in (("wh_prio3_params", "WhPrio3Params"), ("wh_prio3_sizes", "WhPrio3Sizes")):
        body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + cname + r"\s*;", h).group(1)
        c_fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            ty, names = decl.split(None, 1)
            c_fields += [(n.strip(), C_BASE[ty]) for n in names.split(",")]
        rbody = re.search(r"pub\s+struct\s+" + rname + r"\s*\{(.*?)\}", rs, flags=re.S).group(1)
        r_fields = re.findall(r"pub\s+(\w+)\s*:\s*(\w+)", rbody)
        assert r_fields == c_fields, (cname, r_fields, c_fields)
        assert re.search(r"#\[repr\(C\)\]\s*(?:#\[derive[^\]]*\]\s*)?pub\s+struct\s+" + rname, rs)
    # enumerators and #defines with explicit values
    consts = dict(re.findall(r"\b(WH_\w+)\s*=\s*(-?\d+)", h))
    consts["WH_COMM_ID_BYTES"] = re.search(r"#define\s+WH_COMM_ID_BYTES\s+(\d+)", h).group(1)
    r_consts = {k: v.replace(" ", "") for k, v in re.findall(r"pub\s+const\s+(WH_\w+)\s*:\s*\w+\s*=\s*([^;]+);", rs)}
    for name in ("WH_SUMVEC", "WH_HISTOGRAM", "WH_AVERAGE", "WH_OK", "WH_ERR_PARAM", "WH_ERR_SHORT", "WH_ERR_MODULUS", "WH_ERR_AGG_ID",
                 "WH_ERR_NUM_QUERIES", "WH_ERR_QUERY_RAND", "WH_ERR_DECIDE", "WH_ERR_JOINT_RAND", "WH_ERR_KEY_FORMAT", "WH_ERR_HIP",
                 "WH_ERR_NO_DEVICE", "WH_ERR_STATE", "WH_ERR_UNSUPPORTED", "WH_ERR_COMM", "WH_COMM_ID_BYTES", "WH_NUM_KERNELS", "WH_NUM_OPTS"):
        assert r_consts[name] == consts[name], (name, r_consts.get(name), consts.get(name))
    assert r_consts["WH_SORT_NOALLOC_MAX_N"] == "1<<20" and re.search(r"#define\s+WH_SORT_NOALLOC_MAX_N\s+\(1u\s*<<\s*20\)", h)


def test_build_rs_compiles_the_makefile_sources():
    mk = open(os.path.join(ROOT, "whisper_amd", "csrc", "Makefile")).read()
    src = re.search(r"^SRC\s*=\s*(.*)$", mk, flags=re.M).group(1).split()
    rs = open(os.path.join(ROOT, "shim", "build.rs")).read()
    listed = re.findall(r'"([\w.]+\.(?:hip|cpp))"', re.search(r"const\s+SRC\s*:[^=]*=\s*&\[(.*?)\];", rs, flags=re.S).group(1))
    assert listed == src
    for f in src:
        assert os.path.exists(os.path.join(ROOT, "whisper_amd", "csrc", f)), f
    assert "--offload-arch" in rs and "rustc-link-lib=dylib=whisper_gfx950" in rs


def test_safe_wrappers_only_call_declared_entry_points():
    declared = set(ffi_items())
    called = set(re.findall(r"ffi::(wh_\w+)\s*\(", open(LIBRS).read()))
    assert called and called <= declared, sorted(called - declared)
    # the calls INTEGRATION.md names for the three servers are wrapped
    for name in ("wh_prio3_prepare_batch", "wh_prio3_out_shares", "wh_prio3_aggregate", "wh_sort_by_testing_id", "wh_prio3_allgather_reduce",
                 "wh_comm_init", "wh_prio3_prepare_init_batch", "wh_prio3_prepare_finish_batch", "wh_hh_eval_and_check_batch", "wh_hh_aggregate",
                 "wh_ctx_lock", "wh_ctx_unlock"):
        assert name in called, name


def test_integration_md_points_at_the_shim_files():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for path in ("shim/build.rs", "shim/src/ffi.rs", "shim/src/lib.rs"):
        assert path in text and os.path.exists(os.path.join(ROOT, path)), path
