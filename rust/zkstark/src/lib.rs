//! Safe layer over `zkstark-sys` (include/zkstark.h).  Mirrors the Python host mirror of this repository
//! (`zk_evm_amd/context.py`, `polynomial_batch.py`, `segment.py`): RAII handles, `anyhow` errors carrying
//! `zk_last_error`, owned copies of proofs.  No arithmetic happens here and there is no CPU fallback.
use std::ffi::CStr;
use std::ptr::{null, null_mut};
use std::sync::atomic::AtomicBool;
use std::sync::Arc;

use anyhow::{anyhow, Result};
pub use zkstark_sys as sys;
use zkstark_sys::*;

/// Every construction of a plonky2 / starky type whose spelling could not be checked against an in-tree use (feature `upstream`).
#[cfg(feature = "upstream")]
pub mod upstream_compat;

/// `StarkConfig` / `FriConfig` fields the path consumes (`StarkConfig::standard_fast_config()` by default).
#[derive(Clone, Copy)]
pub struct Config(pub zk_cfg);

impl Config {
    pub fn standard_fast(hasher: zk_hasher) -> Self {
        Config(zk_cfg { rate_bits: 1, cap_height: 4, hasher: hasher as u32, num_challenges: 2, proof_of_work_bits: 16,
                        num_query_rounds: 84, arity_bits: 4, final_poly_bits: 5 })
    }
}

/// One `zk_ctx`: a GPU, a stream, an HBM arena.  Not `Sync`: a context must not be used by two threads at once
/// (include/zkstark.h); create one per worker thread, several per GPU if segments should overlap.
pub struct Context {
    raw: *mut zk_ctx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = null_mut();
        let rc = unsafe { zk_ctx_create(device, &mut raw) };
        if rc != ZK_OK || raw.is_null() {
            return Err(anyhow!("zk_ctx_create(device {device}) failed ({rc}): no usable HIP device (no CPU fallback)"));
        }
        Ok(Context { raw })
    }
    pub fn raw(&self) -> *mut zk_ctx { self.raw }
    pub fn check(&self, rc: i32) -> Result<()> {
        if rc == ZK_OK { return Ok(()); }
        if rc == ZK_ERR_ABORTED { return Err(anyhow!("Stopping job from abort signal.")); }   // prover.rs:346-354
        let msg = unsafe { CStr::from_ptr(zk_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(anyhow!("zkstark error {rc}: {msg}"))
    }
    /// `abort_signal` of `prove` (prover.rs:56) handed over as it is: the library polls the `AtomicBool`'s own byte
    /// between kernels (`zk_ctx_set_abort_flag_u8`), so a store from another thread while a proof runs ends it with
    /// "Stopping job from abort signal." (prover.rs:346-354).  The returned guard keeps the `Arc` alive and clears the
    /// pointer when dropped; `None` arms nothing.
    pub fn arm_abort_flag(&self, flag: Option<Arc<AtomicBool>>) -> Result<AbortArmed<'_>> {
        let p = flag.as_ref().map(|f| f.as_ptr() as *const u8).unwrap_or(null());
        self.check(unsafe { zk_ctx_set_abort_flag_u8(self.raw, p) })?;
        Ok(AbortArmed { ctx: self, _keep: flag })
    }
    pub fn mem_reserve(&self, bytes: usize) -> Result<()> { self.check(unsafe { zk_ctx_mem_reserve(self.raw, bytes) }) }
    pub fn synchronize(&self) -> Result<()> { self.check(unsafe { zk_ctx_synchronize(self.raw) }) }
    /// The ctx's plan table (`zk_ctx_set_plans`): which of two equivalent kernels serves a shape, as data -- e.g.
    /// `"v20f0=2;b20r1=96x1;T=1;"`.  `None` = the initial table (`ZK_NTT_SWAP_PLANS` or the one compiled in), `Some("")` = the first
    /// implementation everywhere.  Proofs do not depend on it; the library never measures or spawns anything to fill it.
    pub fn set_plans(&self, plans: Option<&str>) -> Result<()> {
        match plans {
            None => self.check(unsafe { zk_ctx_set_plans(self.raw, null()) }),
            Some(p) => {
                let c = std::ffi::CString::new(p).map_err(|_| anyhow!("plan string with an interior NUL"))?;
                self.check(unsafe { zk_ctx_set_plans(self.raw, c.as_ptr()) })
            }
        }
    }
    pub fn plans(&self) -> String {
        let mut buf = vec![0u8; 4100];
        let n = unsafe { zk_ctx_get_plans(self.raw, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
        buf.truncate(n.min(4099));
        String::from_utf8_lossy(&buf).into_owned()
    }
    /// [ifft, lde, leaf hash, tree] ms of the last commit: children of the reference's "compute trace commitment" scope.
    pub fn last_timings(&self) -> Result<[f32; 4]> {
        let mut t = [0f32; 4];
        self.check(unsafe { zk_ctx_last_timings(self.raw, t.as_mut_ptr()) })?;
        Ok(t)
    }
}
impl Drop for Context {
    fn drop(&mut self) { unsafe { zk_ctx_destroy(self.raw) } }
}

/// Scope of an armed abort flag (`Context::arm_abort_flag`).
pub struct AbortArmed<'c> { ctx: &'c Context, _keep: Option<Arc<AtomicBool>> }
impl Drop for AbortArmed<'_> {
    fn drop(&mut self) { unsafe { zk_ctx_set_abort_flag_u8(self.ctx.raw, null()); } }
}

/// `PolynomialBatch::from_values` result, resident in HBM.
pub struct Batch<'c> { ctx: &'c Context, raw: *mut zk_batch, cap_digests: usize }

impl<'c> Batch<'c> {
    /// `cols[c]` = the `values` of `PolynomialValues<GoldilocksField>` number c (`#[repr(transparent)]` over u64;
    /// non-canonical representatives are fine).  Replaces prover.rs:100-107 without the `trace.clone()`.
    pub fn from_values(ctx: &'c Context, cfg: &Config, cols: &[&[u64]]) -> Result<Self> {
        let n = cols.first().map(|c| c.len()).unwrap_or(0);
        if n == 0 || !n.is_power_of_two() || cols.iter().any(|c| c.len() != n) {
            return Err(anyhow!("columns must have one power-of-two length"));
        }
        let ptrs: Vec<*const u64> = cols.iter().map(|c| c.as_ptr()).collect();
        let mut raw = null_mut();
        ctx.check(unsafe { zk_commit_columns(ctx.raw, &cfg.0, ptrs.as_ptr(), ptrs.len(), n.trailing_zeros(), &mut raw) })?;
        Ok(Batch { ctx, raw, cap_digests: 1usize << cfg.0.cap_height })
    }
    /// `merkle_tree.cap` as 32-byte slots (Poseidon: 4 x u64; Keccak-25: 25 bytes + padding).
    pub fn cap(&self) -> Result<Vec<[u64; 4]>> {
        let mut out = vec![[0u64; 4]; self.cap_digests];
        self.ctx.check(unsafe { zk_batch_cap(self.raw, out.as_mut_ptr() as *mut u64) })?;
        Ok(out)
    }
    pub fn raw(&self) -> *const zk_batch { self.raw }
}
impl Drop for Batch<'_> {
    fn drop(&mut self) { unsafe { zk_batch_free(self.raw) } }
}

/// A column-major `[cols][rows]` device matrix from the context's arena (`zk_dev_alloc` / `zk_dev_upload_columns`).
pub struct DeviceMatrix<'c> { ctx: &'c Context, ptr: *mut u64, cols: usize, rows: usize }

impl<'c> DeviceMatrix<'c> {
    pub fn upload(ctx: &'c Context, cols: &[&[u64]]) -> Result<Self> {
        let rows = cols.first().map(|c| c.len()).unwrap_or(0);
        if rows == 0 || cols.iter().any(|c| c.len() != rows) {
            return Err(anyhow!("DeviceMatrix::upload: columns must be non-empty and of one length"));
        }
        let mut p: *mut core::ffi::c_void = null_mut();
        ctx.check(unsafe { zk_dev_alloc(ctx.raw, 8 * rows * cols.len(), &mut p) })?;
        let m = DeviceMatrix { ctx, ptr: p as *mut u64, cols: cols.len(), rows };
        let ptrs: Vec<*const u64> = cols.iter().map(|c| c.as_ptr()).collect();
        ctx.check(unsafe { zk_dev_upload_columns(ctx.raw, ptrs.as_ptr(), ptrs.len(), rows, m.ptr, rows) })?;
        Ok(m)
    }
    pub fn ptr(&self) -> *const u64 { self.ptr }
    pub fn cols(&self) -> usize { self.cols }
    pub fn rows(&self) -> usize { self.rows }
}
impl Drop for DeviceMatrix<'_> {
    fn drop(&mut self) { unsafe { zk_dev_free(self.ctx.raw, self.ptr as *mut core::ffi::c_void); } }
}

/// One table handed to `prove_segment` (`zk_table_in`): device trace + the table's static description.
pub struct TableIn<'a> {
    /// the table's trace, column-major, 2^k rows: the borrow ties the device pointer and its shape to a live allocation
    pub trace: &'a DeviceMatrix<'a>,
    pub air_id: zk_air,
    pub air_consts: &'a [u64],
    pub lookup_program: &'a [u64],
    pub in_use: bool,
    pub optional: bool,
}

/// `StarkProofWithMetadata` in the flat layout of `zk_table_proof_view` (owned copy).
#[derive(Clone, Debug)]
pub struct TableProof {
    pub degree_bits: u32,
    pub n_trace_cols: usize,
    pub n_aux_cols: usize,
    pub n_quotient_cols: usize,
    pub n_ctl_zs: usize,
    pub trace_cap: Vec<[u64; 4]>,
    pub aux_cap: Option<Vec<[u64; 4]>>,
    pub quotient_cap: Vec<[u64; 4]>,
    /// (c0, c1) per opened value: at zeta (trace, aux, quotient), at g*zeta (trace, aux), at 1 (ctl_zs_first)
    pub openings: Vec<[u64; 2]>,
    /// flat `FriProof`, layout documented at `zk_fri_prove_openings`
    pub opening_proof: Vec<u64>,
    pub init_challenger_state: [u64; 12],
}

/// `AllProof` minus the `PublicValues` the caller already owns.
#[derive(Clone, Debug)]
pub struct SegmentProof {
    pub tables: Vec<Option<TableProof>>,
    pub ctl_challenges: Vec<(u64, u64)>,
    /// `MemCap::from_merkle_cap` of the MemBefore / MemAfter trace caps (prover.rs:261-271), as `to_vec` elements
    pub mem_before: Vec<[u64; 4]>,
    pub mem_after: Vec<[u64; 4]>,
    /// ms: [0] "compute all trace commitments", [1] "compute CTL data", [2 + t] "prove <stark field> STARK"
    pub stage_ms: Vec<f64>,
}

unsafe fn caps(p: *const u64, n: usize) -> Vec<[u64; 4]> {
    (0..n).map(|i| { let s = std::slice::from_raw_parts(p.add(4 * i), 4); [s[0], s[1], s[2], s[3]] }).collect()
}

/// `prove_with_traces` (prover.rs:72-194) as one call.  `ctl_wiring` / the lookup programs are the encodings of
/// `all_stark.cross_table_lookups` and each table's `lookups()` (shipped ready-made in include/zk_all_stark.h).
pub fn prove_segment(ctx: &Context, cfg: &Config, tables: &[TableIn], ctl_wiring: &[u64], public_value_elements: &[u64],
                     constraint_degree: u32, mem_before_table: i32, mem_after_table: i32) -> Result<SegmentProof> {
    if tables.iter().any(|t| !t.trace.rows().is_power_of_two() || t.trace.ctx.raw != ctx.raw) {
        return Err(anyhow!("prove_segment: every trace needs 2^k rows and must live in this context"));
    }
    let tin: Vec<zk_table_in> = tables.iter().map(|t| zk_table_in {
        d_trace: t.trace.ptr(), col_stride: t.trace.rows(), n_cols: t.trace.cols(), log_n: t.trace.rows().trailing_zeros(),
        air_id: t.air_id as u32,
        air_consts: if t.air_consts.is_empty() { null() } else { t.air_consts.as_ptr() }, n_air_consts: t.air_consts.len(),
        lookup_program: if t.lookup_program.is_empty() { null() } else { t.lookup_program.as_ptr() },
        lookup_words: t.lookup_program.len(), in_use: t.in_use as i32, optional: t.optional as i32,
    }).collect();
    let mut h: *mut zk_segment_proof = null_mut();
    ctx.check(unsafe {
        zk_prove_segment(ctx.raw, &cfg.0, tin.as_ptr(), tin.len(), ctl_wiring.as_ptr(), ctl_wiring.len(),
                         public_value_elements.as_ptr(), public_value_elements.len(), constraint_degree,
                         mem_before_table, mem_after_table, &mut h)
    })?;
    read_segment_proof(ctx, cfg, h, tables.len())
}

/// Owned copy of a `zk_segment_proof` (freed here).
fn read_segment_proof(ctx: &Context, cfg: &Config, h: *mut zk_segment_proof, n_tables: usize) -> Result<SegmentProof> {
    struct Guard(*mut zk_segment_proof);
    impl Drop for Guard { fn drop(&mut self) { unsafe { zk_segment_proof_free(self.0) } } }
    let _g = Guard(h);
    let nd = 1usize << cfg.0.cap_height;
    let mut out = SegmentProof { tables: Vec::new(), ctl_challenges: Vec::new(), mem_before: vec![[0; 4]; nd],
                                 mem_after: vec![[0; 4]; nd], stage_ms: vec![0.0; 2 + n_tables] };
    unsafe {
        let mut cc = vec![0u64; 2 * cfg.0.num_challenges as usize];
        zk_segment_proof_ctl_challenges(h, cc.as_mut_ptr(), cc.len());
        out.ctl_challenges = cc.chunks(2).map(|c| (c[0], c[1])).collect();
        zk_segment_proof_mem_caps(h, out.mem_before.as_mut_ptr() as *mut u64, out.mem_after.as_mut_ptr() as *mut u64, 4 * nd);
        zk_segment_proof_stage_ms(h, out.stage_ms.as_mut_ptr(), out.stage_ms.len());
        for t in 0..n_tables {
            let tp = zk_segment_proof_table(h, t);
            out.tables.push(if tp.is_null() { None } else { Some(read_table_proof(ctx, tp)?) });
        }
    }
    Ok(out)
}

unsafe fn read_table_proof(ctx: &Context, tp: *const zk_table_proof) -> Result<TableProof> {
    let mut v: zk_table_proof_view = std::mem::zeroed();
    ctx.check(zk_table_proof_get(tp, &mut v))?;
    let op = std::slice::from_raw_parts(v.openings, 2 * v.n_openings);
    Ok(TableProof {
        degree_bits: v.degree_bits, n_trace_cols: v.n_trace_cols, n_aux_cols: v.n_aux_cols,
        n_quotient_cols: v.n_quotient_cols, n_ctl_zs: v.n_ctl_zs,
        trace_cap: caps(v.trace_cap, v.cap_digests),
        aux_cap: if v.aux_cap.is_null() { None } else { Some(caps(v.aux_cap, v.cap_digests)) },
        quotient_cap: caps(v.quotient_cap, v.cap_digests),
        openings: op.chunks(2).map(|c| [c[0], c[1]]).collect(),
        opening_proof: std::slice::from_raw_parts(v.opening_proof, v.proof_words).to_vec(),
        init_challenger_state: v.init_challenger_state,
    })
}

// ---- several GPUs behind one call (include/zkstark.h "the multi-GPU provers behind this ABI") ------------------------------
/// The ranks of one job: one process (or thread) per GPU, one `Context` each.  `Comm::rccl` is collective like
/// `ncclCommInitRank`: rank 0 makes the id (`Comm::unique_id`), hands it to the others by whatever the caller has (paladin's
/// own channel, a pipe, a file), and every rank calls `Comm::rccl(ctx, &id, rank, world)`.  `Comm::host` is the host-staged
/// transport over POSIX shared memory (ranks may share a GPU; tests, and the fallback where RCCL cannot come up).
pub struct Comm<'c> { ctx: &'c Context, raw: *mut zk_comm }

impl<'c> Comm<'c> {
    pub fn unique_id() -> Result<[u8; ZK_COMM_ID_BYTES]> {
        let mut id = [0u8; ZK_COMM_ID_BYTES];
        let rc = unsafe { zk_comm_unique_id(id.as_mut_ptr()) };
        if rc != ZK_OK { return Err(anyhow!("zk_comm_unique_id failed ({rc}): librccl.so could not be loaded")); }
        Ok(id)
    }
    pub fn rccl(ctx: &'c Context, id: &[u8; ZK_COMM_ID_BYTES], rank: u32, world: u32) -> Result<Self> {
        let mut raw = null_mut();
        ctx.check(unsafe { zk_comm_create(ctx.raw, id.as_ptr(), rank, world, &mut raw) })?;
        Ok(Comm { ctx, raw })
    }
    pub fn host(ctx: &'c Context, name: &str, rank: u32, world: u32) -> Result<Self> {
        let cname = std::ffi::CString::new(name)?;
        let mut raw = null_mut();
        ctx.check(unsafe { zk_comm_create_host(ctx.raw, cname.as_ptr(), rank, world, 0, &mut raw) })?;
        Ok(Comm { ctx, raw })
    }
    pub fn rank(&self) -> u32 { unsafe { zk_comm_rank(self.raw) } }
    pub fn world(&self) -> u32 { unsafe { zk_comm_world(self.raw) } }
    fn check(&self, rc: i32) -> Result<()> {
        if rc == ZK_ERR_COMM {
            let msg = unsafe { CStr::from_ptr(zk_comm_last_error(self.raw)) }.to_string_lossy().into_owned();
            return Err(anyhow!("zkstark communicator failed (dead on every rank, free it): {msg}"));
        }
        self.ctx.check(rc)
    }
}
impl Drop for Comm<'_> {
    fn drop(&mut self) { unsafe { zk_comm_free(self.raw) } }
}

/// `zk_assign_tables`: which rank holds table t's trace (largest cost first onto the least loaded rank; deterministic, the same
/// on every rank).  `row_sharded[t]`: table t is spread over all ranks instead.
pub fn assign_tables(n_cols: &[usize], log_n: &[u32], world: u32, row_sharded: Option<&[u8]>) -> Result<Vec<u32>> {
    let mut owner = vec![0u32; n_cols.len()];
    let rc = unsafe { zk_assign_tables(n_cols.as_ptr(), log_n.as_ptr(), n_cols.len(), world,
                                       row_sharded.map(|r| r.as_ptr()).unwrap_or(null()), owner.as_mut_ptr()) };
    if rc != ZK_OK { return Err(anyhow!("zk_assign_tables failed ({rc})")); }
    Ok(owner)
}

/// One table of a table-parallel segment on THIS rank: the whole trace where this rank owns the table, this rank's row block
/// where the table is row-sharded, `None` elsewhere.  The description (`n_cols`, `log_n` of the WHOLE table, AIR, programs) is
/// the same on every rank.
pub struct ShardedTableIn<'a> {
    pub trace: Option<&'a DeviceMatrix<'a>>,
    pub n_cols: usize,
    pub log_n: u32,
    pub air_id: zk_air,
    pub air_consts: &'a [u64],
    pub lookup_program: &'a [u64],
    pub in_use: bool,
    pub optional: bool,
}

/// `prove_with_traces` (prover.rs:72-194) as a COLLECTIVE call over the ranks of `comm` (SURVEY 8(e) levels 2 and 3): every rank
/// passes the same descriptions and gets the whole `AllProof`, bit-identical to `prove_segment`'s on one GPU.
/// `fri_mode` 0: the two-column FRI layers replicated after one all-gather; 1: every layer on the rank that owns its leaves.
#[allow(clippy::too_many_arguments)]
pub fn prove_segment_table_parallel(ctx: &Context, comm: &Comm, cfg: &Config, tables: &[ShardedTableIn], row_sharded: Option<&[u8]>,
                                    ctl_wiring: &[u64], public_value_elements: &[u64], constraint_degree: u32,
                                    mem_before_table: i32, mem_after_table: i32, fri_mode: u32) -> Result<SegmentProof> {
    if comm.ctx.raw != ctx.raw || tables.iter().any(|t| t.trace.map_or(false, |m| m.ctx.raw != ctx.raw || m.cols() != t.n_cols)) {
        return Err(anyhow!("prove_segment_table_parallel: communicator and traces must live in this context"));
    }
    let tin: Vec<zk_table_in> = tables.iter().map(|t| zk_table_in {
        d_trace: t.trace.map_or(null(), |m| m.ptr()), col_stride: t.trace.map_or(0, |m| m.rows()), n_cols: t.n_cols, log_n: t.log_n,
        air_id: t.air_id as u32,
        air_consts: if t.air_consts.is_empty() { null() } else { t.air_consts.as_ptr() }, n_air_consts: t.air_consts.len(),
        lookup_program: if t.lookup_program.is_empty() { null() } else { t.lookup_program.as_ptr() },
        lookup_words: t.lookup_program.len(), in_use: t.in_use as i32, optional: t.optional as i32,
    }).collect();
    let mut h: *mut zk_segment_proof = null_mut();
    comm.check(unsafe {
        zk_prove_segment_table_parallel(ctx.raw, comm.raw, &cfg.0, tin.as_ptr(), tin.len(), row_sharded.map(|r| r.as_ptr()).unwrap_or(null()),
                                        ctl_wiring.as_ptr(), ctl_wiring.len(), public_value_elements.as_ptr(), public_value_elements.len(),
                                        constraint_degree, mem_before_table, mem_after_table, fri_mode, &mut h)
    })?;
    read_segment_proof(ctx, cfg, h, tables.len())
}
