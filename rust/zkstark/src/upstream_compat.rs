//! The ONE place where this crate (and the reference-side patch, `evm_arithmetization/src/hip.rs`) spells a plonky2 / starky
//! 1.0.0 item that no code inside the reference tree spells: the reference only ever RECEIVES a `FriProof`, a
//! `StarkOpeningSet`, a `MerkleProof` from `starky::prover` -- it never builds one by hand -- so their field names, the
//! `PolynomialCoeffs::new` / `FieldExtension::from_basefield_array` constructors and the `TimingTree::push` / `pop` signatures
//! rest on recollection of the 1.0.0 sources (rust/upstream_api.json, status "recalled"; the crates are not vendored and this
//! image has no `cargo`).  Everything else the shim uses is corroborated by an in-tree caller (same file).
//!
//! So: the first `cargo build --features hip` can only fail HERE for a mis-remembered upstream spelling -- one constructor per
//! recalled type, nothing else in the function bodies.  tests/test_rust_shim.py fails when a recalled item is named anywhere
//! outside this file.  rust/README.md lists the constructors as the compile errors to expect, with what to check for each.
use plonky2::field::extension::{Extendable, FieldExtension};
use plonky2::field::polynomial::PolynomialCoeffs;
use plonky2::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use plonky2::hash::hash_types::RichField;
use plonky2::hash::merkle_proofs::MerkleProof;
use plonky2::hash::merkle_tree::MerkleCap;
use plonky2::plonk::config::{GenericConfig, Hasher};
use plonky2::util::timing::TimingTree;
use starky::proof::{StarkOpeningSet, StarkProof};

/// [recalled path] `plonky2::fri::proof::FriProof`, for signatures outside this file.
pub type UpFriProof<F, H, const D: usize> = FriProof<F, H, D>;

/// [recalled] `FieldExtension::from_basefield_array([F; D])`.  If absent: `<F::Extension as OEF<D>>`-style constructors or
/// `F::Extension::from_basefield` + powers of the generator give the same element.
pub fn extension_from_base<F: RichField + Extendable<D>, const D: usize>(limbs: [F; D]) -> F::Extension {
    F::Extension::from_basefield_array(limbs)
}

/// [recalled] `PolynomialCoeffs::new(Vec<T>)` (pub field `coeffs`).
pub fn polynomial_coeffs<T: plonky2::field::types::Field>(coeffs: Vec<T>) -> PolynomialCoeffs<T> {
    PolynomialCoeffs::new(coeffs)
}

/// [recalled] `MerkleProof { siblings: Vec<H::Hash> }`.
pub fn merkle_proof<F: RichField, H: Hasher<F>>(siblings: Vec<H::Hash>) -> MerkleProof<F, H> {
    MerkleProof { siblings }
}

/// [recalled] `FriQueryStep { evals: Vec<F::Extension>, merkle_proof }`.
pub fn fri_query_step<F: RichField + Extendable<D>, H: Hasher<F>, const D: usize>(
    evals: Vec<F::Extension>,
    merkle_proof: MerkleProof<F, H>,
) -> FriQueryStep<F, H, D> {
    FriQueryStep { evals, merkle_proof }
}

/// [recalled] `FriInitialTreeProof { evals_proofs: Vec<(Vec<F>, MerkleProof<F, H>)> }` -- one (leaf, path) per oracle.
pub fn fri_initial_tree_proof<F: RichField, H: Hasher<F>>(evals_proofs: Vec<(Vec<F>, MerkleProof<F, H>)>) -> FriInitialTreeProof<F, H> {
    FriInitialTreeProof { evals_proofs }
}

/// [recalled] `FriQueryRound { initial_trees_proof, steps }`.
pub fn fri_query_round<F: RichField + Extendable<D>, H: Hasher<F>, const D: usize>(
    initial_trees_proof: FriInitialTreeProof<F, H>,
    steps: Vec<FriQueryStep<F, H, D>>,
) -> FriQueryRound<F, H, D> {
    FriQueryRound { initial_trees_proof, steps }
}

/// [recalled] `FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }`.
pub fn fri_proof<F: RichField + Extendable<D>, H: Hasher<F>, const D: usize>(
    commit_phase_merkle_caps: Vec<MerkleCap<F, H>>,
    query_round_proofs: Vec<FriQueryRound<F, H, D>>,
    final_poly: PolynomialCoeffs<F::Extension>,
    pow_witness: F,
) -> FriProof<F, H, D> {
    FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }
}

/// [recalled as a literal; the FIELD NAMES are corroborated by readers, `recursive_verifier.rs:304`, `verifier.rs:286-292`]
/// `StarkOpeningSet { local_values, next_values, auxiliary_polys, auxiliary_polys_next, ctl_zs_first, quotient_polys }` with the
/// four optional members as `Option<Vec<_>>`.
#[allow(clippy::too_many_arguments)]
pub fn stark_opening_set<F: RichField + Extendable<D>, const D: usize>(
    local_values: Vec<F::Extension>,
    next_values: Vec<F::Extension>,
    auxiliary_polys: Option<Vec<F::Extension>>,
    auxiliary_polys_next: Option<Vec<F::Extension>>,
    ctl_zs_first: Option<Vec<F>>,
    quotient_polys: Option<Vec<F::Extension>>,
) -> StarkOpeningSet<F, D> {
    StarkOpeningSet { local_values, next_values, auxiliary_polys, auxiliary_polys_next, ctl_zs_first, quotient_polys }
}

/// [recalled as a literal; field names corroborated by `get_challenges.rs:283-296`]
/// `StarkProof { trace_cap, auxiliary_polys_cap: Option<_>, quotient_polys_cap: Option<_>, openings, opening_proof }`.
pub fn stark_proof<F: RichField + Extendable<D>, C: GenericConfig<D, F = F>, const D: usize>(
    trace_cap: MerkleCap<F, C::Hasher>,
    auxiliary_polys_cap: Option<MerkleCap<F, C::Hasher>>,
    quotient_polys_cap: Option<MerkleCap<F, C::Hasher>>,
    openings: StarkOpeningSet<F, D>,
    opening_proof: FriProof<F, C::Hasher, D>,
) -> StarkProof<F, C, D> {
    StarkProof { trace_cap, auxiliary_polys_cap, quotient_polys_cap, openings, opening_proof }
}

/// [recalled] `TimingTree::push(&mut self, ctx: &str, level: log::Level)` / `pop(&mut self)`: an empty scope whose label carries
/// a stage time measured on the device (the reference itself only uses the `timed!` macro, `prover.rs:95-111`).
pub fn timing_scope(timing: &mut TimingTree, label: &str) {
    timing.push(label, log::Level::Debug);
    timing.pop();
}
