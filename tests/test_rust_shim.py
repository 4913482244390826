"""The Rust shim (rust/ark-mi355x, uncompiled here: the image has no Rust toolchain) must declare exactly the entry
points of include/ark355.h: same names, same order, same argument types.  Both files are parsed and compared; the
constants and #[repr(C)] structs are compared field by field as well."""
import os
import re

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "ark355.h")
FFI = os.path.join(ROOT, "rust", "ark-mi355x", "src", "ffi.rs")

C2RUST = {
    "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "float": "f32", "void": "()",
    "uint8_t": "u8", "int8_t": "i8", "char": "c_char",
}


def _strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def _c_type(t, array):
    """C parameter type -> Rust spelling."""
    t = t.strip()
    const = False
    ptr = t.count("*") + (1 if array else 0)
    words = [w for w in re.split(r"[\s\*]+", t) if w]
    # `const T* const x[3]` (array of const pointers) / `const T*`
    if words and words[0] == "const":
        const = True
        words = words[1:]
    words = [w for w in words if w != "const"]
    base = words[0]
    rust = C2RUST.get(base, base)
    if ptr == 0:
        return rust
    if base == "void":
        rust = "c_void"
    out = rust
    for level in range(ptr):
        # innermost pointer carries the constness of the pointee; outer levels of `T* const x[]` are const as well
        out = ("*const " if const else "*mut ") + out
    return out


def c_prototypes():
    src = _strip_comments(open(HDR).read())
    protos = []
    for m in re.finditer(r"\b((?:const\s+)?[a-z_0-9]+\s*\**)\s*(ark355_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args != "void" and args:
            for a in args.split(","):
                a = a.strip()
                array = bool(re.search(r"\[[^\]]*\]$", a))
                a = re.sub(r"\[[^\]]*\]$", "", a).strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)
                params.append(_c_type(mm.group(1), array))
        protos.append((name, params, _c_type(ret, False)))
    return protos


def rust_prototypes():
    src = _strip_comments(open(FFI).read())
    block = src[src.index('extern "C" {'):]
    protos = []
    for m in re.finditer(r"pub fn (ark355_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "()").strip()
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if ":" in a]
        protos.append((name, params, ret))
    return protos


def test_every_abi_entry_point_is_bound_with_matching_arguments():
    c, r = c_prototypes(), rust_prototypes()
    assert len(c) >= 40, "header parse failed"
    assert [n for n, _, _ in c] == [n for n, _, _ in r], "entry points / order differ"
    for (name, cp, cr), (_, rp, rr) in zip(c, r):
        assert cp == rp, "%s: C %s vs Rust %s" % (name, cp, rp)
        assert cr == rr, "%s: return C %s vs Rust %s" % (name, cr, rr)


def test_constants_and_structs_agree():
    h = _strip_comments(open(HDR).read())
    f = open(FFI).read()
    consts = dict((k, int(v)) for k, v in re.findall(r"\b(ARK355_[A-Z0-9_]+)\s*=\s*(-?\d+)", h))
    consts.update((k, int(v)) for k, v in re.findall(r"#define\s+(ARK355_[A-Z0-9_]+)\s+(\d+)", h))
    assert len(consts) >= 14
    for k, v in consts.items():
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % k, f)
        assert m and int(m.group(1)) == v, k
    for struct in ("ark355_pk_desc", "ark355_proof_raw", "ark355_timings", "ark355_sched_report"):
        body = re.search(r"typedef struct \{([^{}]*?)\}\s*%s;" % struct, h, flags=re.S).group(1)
        c_fields = [re.sub(r"\[\d+\]", "", x.strip().split()[-1].lstrip("*")) for x in body.split(";") if x.strip()]
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % struct, f, flags=re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", rbody)
        assert c_fields == r_fields, struct


def test_shim_maps_device_errors_to_distinct_variants():
    """a HIP / RCCL / out-of-memory / no-device failure must not read as `Unsatisfiable`."""
    src = open(os.path.join(ROOT, "rust", "ark-mi355x", "src", "error.rs")).read()
    arms = dict(re.findall(r"ffi::(ARK355_[A-Z_]+) => Self::(\w+)", src))
    assert arms["ARK355_EHIP"] == "Hip" and arms["ARK355_ERCCL"] == "Rccl" and arms["ARK355_ENOMEM"] == "OutOfMemory"
    assert arms["ARK355_ENODEV"] == "NoDevice" and arms["ARK355_EINVAL"] == "InvalidArgument"
    assert arms["ARK355_E_UNSATISFIABLE"] == "Synthesis"
    lib = open(os.path.join(ROOT, "rust", "ark-mi355x", "src", "lib.rs")).read()
    for needle in ("impl<E, P1, P2> SNARK<E::ScalarField> for Mi355xGroth16<E>", "CircuitSpecificSetupSNARK<E::ScalarField>",
                   "ark355_prove_batch", "ark355_prove_sharded", "ark355_pk_load_shard", "ark355_comm_init"):
        assert needle in lib, needle
