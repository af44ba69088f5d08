#!/usr/bin/env python3
"""Every plonky2 / starky 1.0.0 item the reference-side patch (rust/evm_arithmetization_hip.patch) and rust/zkstark rely on,
with where the SAME item is used inside the reference tree (an in-tree caller corroborates the path / name / field; the
crates themselves are not vendored and rustc is absent, so anything without one is marked `recalled`).
  tools/rust_api_table.py            (in the build container, /root/reference present) rewrites rust/upstream_api.json
  tools/rust_api_table.py --markdown prints the table of INTEGRATION.md section 6 from that file
tests/test_rust_shim.py fails when the patch imports or calls an upstream item that is not in the file."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "rust", "evm_arithmetization_hip.patch")
COMPAT = os.path.join(ROOT, "rust", "zkstark", "src", "upstream_compat.rs")      # the one file that names the recalled items
OUT = os.path.join(ROOT, "rust", "upstream_api.json")
REF = "/root/reference"

# struct fields, methods and associated items the patch touches (name, owner, kind, regex that finds an in-tree use)
MEMBERS = [
    ("TimingTree::push", "plonky2::util::timing::TimingTree", "method (&mut self, ctx: &str, level: log::Level)", r"timing\.push\("),
    ("TimingTree::pop", "plonky2::util::timing::TimingTree", "method (&mut self)", r"timing\.pop\("),
    ("StarkConfig.fri_config / .num_challenges", "starky::config::StarkConfig", "pub fields", r"config\.fri_config\.|config\.num_challenges"),
    ("FriConfig.rate_bits / .cap_height / .proof_of_work_bits / .num_query_rounds / .reduction_strategy", "plonky2::fri::FriConfig", "pub fields", r"fri_config\.(rate_bits|cap_height|num_query_rounds|proof_of_work_bits|reduction_strategy)"),
    ("FriReductionStrategy::ConstantArityBits(usize, usize)", "plonky2::fri::reduction_strategies::FriReductionStrategy", "enum variant", r"ConstantArityBits\("),
    ("StarkProofWithMetadata { proof, init_challenger_state }", "starky::proof::StarkProofWithMetadata", "pub fields", r"StarkProofWithMetadata \{"),
    ("StarkProof { trace_cap, auxiliary_polys_cap, quotient_polys_cap, openings, opening_proof }", "starky::proof::StarkProof", "pub fields", r"\.proof\.(trace_cap|openings|opening_proof|auxiliary_polys_cap|quotient_polys_cap)"),
    ("StarkOpeningSet { local_values, next_values, auxiliary_polys, auxiliary_polys_next, ctl_zs_first, quotient_polys }", "starky::proof::StarkOpeningSet", "pub fields", r"openings\.(local_values|next_values|auxiliary_polys|ctl_zs_first|quotient_polys)"),
    ("FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }", "plonky2::fri::proof::FriProof", "pub fields", r"(commit_phase_merkle_caps|query_round_proofs|pow_witness)"),
    ("FriQueryRound { initial_trees_proof, steps } / FriInitialTreeProof { evals_proofs } / FriQueryStep { evals, merkle_proof }", "plonky2::fri::proof", "pub fields", r"(initial_trees_proof|evals_proofs|FriQueryStep)"),
    ("MerkleProof { siblings }", "plonky2::hash::merkle_proofs::MerkleProof", "pub field", r"MerkleProof \{|\.siblings"),
    ("MerkleCap(pub Vec<H::Hash>)", "plonky2::hash::merkle_tree::MerkleCap", "tuple struct, pub field", r"MerkleCap\(|\.0\.len\(\)|cap\.0"),
    ("GenericHashOut::from_bytes / Hasher::HASH_SIZE", "plonky2::plonk::config", "trait items", r"from_bytes\(|HASH_SIZE"),
    ("PlonkyPermutation::new(iter)", "plonky2::hash::hashing::PlonkyPermutation", "trait method", r"Permutation::new\("),
    ("GrandProductChallengeSet { challenges } / GrandProductChallenge { beta, gamma }", "starky::lookup", "pub fields", r"GrandProductChallengeSet \{|GrandProductChallenge \{"),
    ("PolynomialValues.values / PolynomialCoeffs::new", "plonky2::field::polynomial", "pub field / constructor", r"\.values\b|PolynomialCoeffs::new|PolynomialValues::new"),
    ("Field::from_canonical_u64 / PrimeField64::to_canonical_u64", "plonky2::field::types", "trait methods", r"from_canonical_u64\(|to_canonical_u64\("),
    ("FieldExtension::from_basefield_array", "plonky2::field::extension::FieldExtension", "trait method", r"from_basefield_array\("),
]


def patch_uses():
    """[(path, item)] of the `use plonky2:: / starky::` lines the patch ADDS"""
    out = []
    for ln in list(open(PATCH)) + ["+" + x for x in open(COMPAT)]:
        m = re.match(r"\+use ((?:plonky2|starky)(?:::\w+)*)::(\{[^}]*\}|\w+);", ln.strip())
        if not m:
            continue
        items = m.group(2).strip("{}").split(",") if m.group(2).startswith("{") else [m.group(2)]
        out += [(m.group(1), it.strip()) for it in items if it.strip() and (m.group(1), it.strip()) not in out]
    return out


def grep_ref(pattern, fixed=False):
    r = subprocess.run(["grep", "-rnE" if not fixed else "-rnF", "--include=*.rs", pattern, "evm_arithmetization/src", "zero/src"],
                       cwd=REF, capture_output=True, text=True)
    hits = [h for h in r.stdout.splitlines() if "/hip.rs" not in h]
    return hits[0].rsplit(":", 1)[0] if False else (":".join(hits[0].split(":")[:2]) if hits else None)


def corroborate_import(path, item):
    # the same item imported from the same module somewhere in the tree (possibly inside a brace list)
    for pat in (r"use %s::%s\b" % (path, item), r"use %s::\{[^}]*\b%s\b" % (path, item), r"%s::%s\b" % (path, item)):
        hit = grep_ref(pat)
        if hit:
            return hit
    return None


def build():
    rows = []
    for path, item in patch_uses():
        hit = corroborate_import(path, item)
        rows.append({"item": "%s::%s" % (path, item), "kind": "import", "in_tree": hit, "status": "corroborated" if hit else "recalled"})
    for name, owner, kind, pat in MEMBERS:
        hit = grep_ref(pat)
        rows.append({"item": name, "owner": owner, "kind": kind, "in_tree": hit, "status": "corroborated" if hit else "recalled"})
    json.dump(rows, open(OUT, "w"), indent=1)
    return rows


def markdown(rows):
    print("| Upstream item (plonky2 / starky 1.0.0) | Kind | In-tree use that corroborates it | Status |")
    print("|---|---|---|---|")
    for r in rows:
        print("| `%s` | %s | %s | %s |" % (r["item"], r["kind"], ("`%s`" % r["in_tree"]) if r["in_tree"] else "—", r["status"]))


if __name__ == "__main__":
    if "--markdown" in sys.argv:
        markdown(json.load(open(OUT)))
    else:
        if not os.path.isdir(REF):
            sys.exit("the reference tree is not mounted: rust/upstream_api.json is regenerated in the build container only")
        rows = build()
        print("%d items, %d recalled" % (len(rows), sum(r["status"] == "recalled" for r in rows)))
        for r in rows:
            if r["status"] == "recalled":
                print("  recalled:", r["item"])
