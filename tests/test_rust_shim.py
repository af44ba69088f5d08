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


def test_recalled_upstream_items_live_in_one_module_only():
    """r04 verdict, item 8: the items of rust/upstream_api.json that no in-tree code corroborates ("recalled") are each spelled in
    exactly one file, rust/zkstark/src/upstream_compat.rs, so that the first `cargo build` can only fail there for a mis-remembered
    upstream name.  Fails when the patch or the safe wrappers name one of them, when upstream_compat.rs does not, or when the
    README's list of constructors is out of step with the file."""
    import json
    rows = json.load(open(os.path.join(ROOT, "rust", "upstream_api.json")))
    member_patterns = {                           # recalled rows that are not imports: how a use of them looks in source text
        "TimingTree::push": r"\.push\(.*log::Level", "TimingTree::pop": r"timing\.pop\(",
        "FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }": r"\bFriProof\s*\{",
        "FriQueryRound { initial_trees_proof, steps } / FriInitialTreeProof { evals_proofs } / FriQueryStep { evals, merkle_proof }":
            r"\b(FriQueryRound|FriInitialTreeProof|FriQueryStep)\s*\{",
        "MerkleProof { siblings }": r"\bMerkleProof\s*\{", "FieldExtension::from_basefield_array": r"from_basefield_array\(",
    }
    patterns = {}
    for r in rows:
        if r["status"] != "recalled":
            continue
        if r["kind"] == "import":
            patterns[r["item"]] = r"\b%s\b" % r["item"].rsplit("::", 1)[1]
        else:
            assert r["item"] in member_patterns, "new recalled member %r: add its pattern here" % r["item"]
            patterns[r["item"]] = member_patterns[r["item"]]
    assert len(patterns) >= 10
    compat = open(os.path.join(ROOT, "rust", "zkstark", "src", "upstream_compat.rs")).read()
    code = "\n".join(ln for ln in compat.splitlines() if not ln.lstrip().startswith("//"))
    elsewhere = {
        "evm_arithmetization_hip.patch": "\n".join(ln[1:] for ln in open(os.path.join(ROOT, "rust", "evm_arithmetization_hip.patch"))
                                                   if ln.startswith("+") and not ln.startswith("+++")),
        "zkstark/src/lib.rs": open(os.path.join(ROOT, "rust", "zkstark", "src", "lib.rs")).read(),
    }
    for item, pat in patterns.items():
        assert re.search(pat, code), "upstream_compat.rs does not use the recalled item %s" % item
        for name, text in elsewhere.items():
            body = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("//"))
            assert not re.search(pat, body), "%s names the recalled upstream item %s outside upstream_compat.rs" % (name, item)
    # the wrappers reach the module only behind its feature, and the patch asks for that feature
    assert '#[cfg(feature = "upstream")]\npub mod upstream_compat;' in elsewhere["zkstark/src/lib.rs"]
    assert 'features = ["upstream"]' in elsewhere["evm_arithmetization_hip.patch"] and "zkstark::upstream_compat as up" in elsewhere["evm_arithmetization_hip.patch"]
    toml = open(os.path.join(ROOT, "rust", "zkstark", "Cargo.toml")).read()
    assert re.search(r'upstream = \[[^\]]*"dep:plonky2"[^\]]*"dep:starky"', toml)
    # README lists every constructor of the module, and nothing that is not there
    ctors = set(re.findall(r"^pub (?:fn|type) (\w+)", compat, re.M))
    readme = open(os.path.join(ROOT, "rust", "README.md")).read()
    listed = set(re.findall(r"^\| `(\w+)`(?: / `(\w+)`)? \|", readme, re.M))
    listed = {x for pair in listed for x in pair if x}
    assert ctors == listed, (ctors ^ listed)
    used = set(re.findall(r"\bup::(\w+)", elsewhere["evm_arithmetization_hip.patch"]))
    assert used <= ctors, used - ctors


def test_safe_wrapper_covers_the_multi_gpu_entry_points():
    """r04 verdict, item 2: levels 2 and 3 are one call from Rust, not a porting guide."""
    lib = open(os.path.join(ROOT, "rust", "zkstark", "src", "lib.rs")).read()
    for name in ("zk_comm_unique_id", "zk_comm_create", "zk_comm_create_host", "zk_comm_free", "zk_assign_tables",
                 "zk_prove_segment_table_parallel"):
        assert name + "(" in lib, name
    assert "pub struct Comm" in lib and "impl Drop for Comm" in lib and "ZK_ERR_COMM" in lib
