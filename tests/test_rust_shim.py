"""CPU: the Rust shim that cannot be compiled here (no cargo / rustc in the image) is at least kept honest --
rust/zkstark-sys/src/lib.rs declares exactly the functions of include/zkstark.h (it is generated from the header and
must be regenerated when the header changes), the POD structs have the C layout's field order, and the reference-side
patch touches the binding points SURVEY 8(f)3 names."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "rust", "zkstark-sys", "src", "lib.rs")


def _header_symbols():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zkstark.h")).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", src)))


def test_sys_crate_declares_exactly_the_header():
    rs = open(SYS).read()
    fns = re.findall(r"pub fn (zk_\w+)\(", rs)
    assert sorted(fns) == _header_symbols() and len(fns) == len(set(fns))
    assert '#[link(name = "zkstark_hip")]' in rs


def test_sys_crate_is_regenerated_from_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "rust/zkstark-sys/src/lib.rs is stale: run tools/gen_rust_sys.py"


def test_struct_field_order_matches_c():
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zkstark.h")).read(), flags=re.S)
    rs = open(SYS).read()
    for name in ("zk_cfg", "zk_fri_batch", "zk_table_proof_view", "zk_table_in"):
        body = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*%s\s*;" % name, hdr).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                c_fields += [re.sub(r"\[\d+\]", "", x).strip("* ").split()[-1].strip("*") for x in decl.split(",")]
        r_fields = re.findall(r"pub (\w+):", re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, re.S).group(1))
        assert r_fields == c_fields, name


def test_safe_wrapper_only_uses_declared_functions():
    rs = open(SYS).read()
    declared = set(re.findall(r"pub fn (zk_\w+)\(", rs))
    used = set(re.findall(r"\b(zk_[a-z0-9_]+)\(", open(os.path.join(ROOT, "rust", "zkstark", "src", "lib.rs")).read()))
    assert used <= declared, used - declared


def test_reference_patch_touches_the_binding_points():
    p = open(os.path.join(ROOT, "rust", "evm_arithmetization_hip.patch")).read()
    for path in ("evm_arithmetization/src/prover.rs", "evm_arithmetization/src/hip.rs", "evm_arithmetization/Cargo.toml",
                 "zero/src/ops.rs"):
        assert ("+++ b/" + path) in p, path
    assert 'feature = "hip"' in p and "zk" "stark::prove_segment" in p
    # a patch, not a copy: only a few context lines of the reference's text travel with it
    context = [ln for ln in p.splitlines() if ln.startswith(" ")]
    assert len(context) < 60


def test_shim_abort_and_public_values_use_what_exists():
    """r02 verdict, weak 10: the abort signal is the reference's own `AtomicBool` byte (no twin that nobody stores to),
    and the public-value elements are restated with the crate's limb helpers instead of a `Challenger` accessor plonky2
    1.0.0 does not have."""
    p = open(os.path.join(ROOT, "rust", "evm_arithmetization_hip.patch")).read()
    lib = open(os.path.join(ROOT, "rust", "zkstark", "src", "lib.rs")).read()
    assert "input_buffer_history" not in p and "AtomicI32" not in p and "AtomicI32" not in lib
    assert "arm_abort_flag(abort_signal" in p
    assert "zk_ctx_set_abort_flag_u8(self.raw, p)" in lib and "as_ptr() as *const u8" in lib
    for helper in ("h256_limbs", "u256_limbs", "u256_to_u32", "u256_to_u64"):
        assert helper in p
    # hunk header of the new file matches its body
    m = re.search(r"\+\+\+ b/evm_arithmetization/src/hip\.rs.*?\n@@ -0,0 \+1,(\d+) @@\n((?:\+.*\n)+)", p)
    assert int(m.group(1)) == len(m.group(2).splitlines())


def test_patch_uses_only_upstream_items_of_the_table():
    """rust/upstream_api.json (tools/rust_api_table.py) lists every plonky2 / starky item the
    reference-side patch relies on with the in-tree use that corroborates it.  An import or a TimingTree method the table does
    not know is how `pop_with_duration_ms` and a wrong module path got into r03: fail here instead."""
    import json
    rows = json.load(open(os.path.join(ROOT, "rust", "upstream_api.json")))
    known = {r["item"] for r in rows}
    p = open(os.path.join(ROOT, "rust", "evm_arithmetization_hip.patch")).read()
    for ln in p.splitlines():
        m = re.match(r"\+use ((?:plonky2|starky)(?:::\w+)*)::(\{[^}]*\}|\w+);", ln.strip())
        if m:
            items = m.group(2).strip("{}").split(",") if m.group(2).startswith("{") else [m.group(2)]
            for it in items:
                assert "%s::%s" % (m.group(1), it.strip()) in known, ln
    methods = set(re.findall(r"\btiming\.(\w+)\(", "\n".join(ln for ln in p.splitlines() if ln.startswith("+"))))
    assert methods <= {"push", "pop"}, methods
    assert {"TimingTree::push", "TimingTree::pop"} <= known
    # every row says where it was corroborated, or admits that it was not
    assert all(r["status"] in ("corroborated", "recalled") and (r["in_tree"] or r["status"] == "recalled") for r in rows)
    assert "plonky2::hash::hashing::PlonkyPermutation" in known and not any("plonk_common::PlonkyPermutation" in k for k in known)
