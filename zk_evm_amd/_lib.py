"""ctypes binding of libzkstark_hip.so (C ABI in include/zkstark.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``zk_evm_amd.build``; if it is
missing this module raises -- there is deliberately no fallback implementation.
"""
import ctypes as C
import os

from .config import ZkCfg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libzkstark_hip.so"


class ZkStarkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zkstark error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # ZK_STARK_LIB: a diagnostic build of the same sources (tools/asan_build.sh: AddressSanitizer + UBSan on the host side)
    return os.environ.get("ZK_STARK_LIB") or os.path.join(_HERE, _LIB_NAME)


_lib = None

# name -> (restype, argtypes); must list every symbol include/zkstark.h declares
vp, u64p, sz, u32, ui = C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint
SIGNATURES = {
    "zk_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "zk_ctx_destroy": (None, [vp]),
    "zk_ctx_set_stream": (C.c_int, [vp, vp]),
    "zk_ctx_synchronize": (C.c_int, [vp]),
    "zk_ctx_mem_reserve": (C.c_int, [vp, sz]),
    "zk_ctx_mem_trim": (C.c_int, [vp, C.POINTER(sz)]),
    "zk_ctx_mem_stats": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "zk_dev_alloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
    "zk_dev_free": (C.c_int, [vp, vp]),
    "zk_dev_upload_columns": (C.c_int, [vp, C.POINTER(vp), sz, sz, u64p, sz]),
    "zk_last_error": (C.c_char_p, [vp]),
    "zk_ctx_set_abort_flag": (C.c_int, [vp, vp]),
    "zk_ctx_set_abort_flag_u8": (C.c_int, [vp, vp]),
    "zk_ctx_set_plans": (C.c_int, [vp, C.c_char_p]),
    "zk_ctx_get_plans": (C.c_size_t, [vp, C.c_char_p, C.c_size_t]),
    "zk_ctx_last_timings": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "zk_ctx_commit_totals": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "zk_ctx_side_commit_totals": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.c_int]),
    "zk_commit_columns": (C.c_int, [vp, C.POINTER(ZkCfg), C.POINTER(vp), sz, ui, C.POINTER(vp)]),
    "zk_commit_columns_device": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, sz, sz, ui, C.POINTER(vp)]),
    "zk_commit_coeffs_device": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, sz, sz, ui, C.POINTER(vp)]),
    "zk_batch_free": (None, [vp]),
    "zk_batch_num_cols": (sz, [vp]),
    "zk_batch_log_n": (ui, [vp]),
    "zk_batch_log_lde": (ui, [vp]),
    "zk_batch_cap": (C.c_int, [vp, u64p]),
    "zk_batch_coeffs": (C.c_int, [vp, sz, u64p]),
    "zk_batch_leaf": (C.c_int, [vp, sz, u64p]),
    "zk_batch_merkle_path": (C.c_int, [vp, sz, u64p]),
    "zk_batch_lde_values": (C.c_int, [vp, sz, sz, u64p]),
    "zk_batch_lde_device": (vp, [vp]),
    "zk_batch_digests_device": (vp, [vp]),
    "zk_ifft": (C.c_int, [vp, u64p, sz, sz, ui]),
    "zk_fft": (C.c_int, [vp, u64p, sz, sz, ui]),
    "zk_coset_fft": (C.c_int, [vp, u64p, sz, sz, ui, C.c_uint64]),
    "zk_coset_ifft": (C.c_int, [vp, u64p, sz, sz, ui, C.c_uint64]),
    "zk_lde": (C.c_int, [vp, u64p, sz, u64p, sz, sz, ui, ui]),
    "zk_gl_vec_op": (C.c_int, [vp, u32, u64p, u64p, u64p, sz]),
    "zk_poseidon_permute": (C.c_int, [vp, u64p, sz]),
    "zk_keccak_f1600": (C.c_int, [vp, u64p, sz]),
    "zk_hash_rows": (C.c_int, [vp, u32, u64p, sz, sz, sz, u64p]),
    "zk_merkle_num_digests": (sz, [ui, ui]),
    "zk_merkle_build": (C.c_int, [vp, u32, u64p, ui, ui]),
    "zk_challenger_create": (C.c_int, [u32, C.POINTER(vp)]),
    "zk_challenger_free": (None, [vp]),
    "zk_challenger_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "zk_challenger_observe_elements": (C.c_int, [vp, u64p, sz]),
    "zk_challenger_observe_cap": (C.c_int, [vp, u64p, sz]),
    "zk_challenger_get_challenge": (C.c_uint64, [vp]),
    "zk_challenger_get_extension_challenge": (C.c_int, [vp, u64p]),
    "zk_challenger_compact": (C.c_int, [vp, u64p]),
    "zk_challenger_export": (C.c_int, [vp, u64p]),
    "zk_challenger_import": (C.c_int, [vp, u64p]),
    "zk_fri_reduction_arity_bits": (sz, [C.POINTER(ZkCfg), ui, vp, sz]),
    "zk_fri_openings": (C.c_int, [vp, vp, sz, vp, sz, u64p]),
    "zk_fri_proof_words": (sz, [C.POINTER(ZkCfg), ui, vp, sz]),
    "zk_fri_prove_openings": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, vp, sz, u64p, vp, u64p]),
    "zk_lookup_helper_columns": (C.c_int, [vp, u64p, sz, sz, ui, u64p, sz, C.c_uint64, ui, u64p, sz,
                                           C.POINTER(sz)]),
    "zk_ctl_partial_sums": (C.c_int, [vp, u64p, sz, sz, ui, u64p, sz, C.c_uint64, C.c_uint64, ui, u64p, sz,
                                      C.POINTER(sz)]),
    "zk_quotient_polys": (C.c_int, [vp, C.POINTER(ZkCfg), u32, u64p, sz, vp, vp, u64p, u64p, sz, u64p, sz,
                                    u64p, sz, ui, C.POINTER(vp)]),
    "zk_prove_table": (C.c_int, [vp, C.POINTER(ZkCfg), u32, u64p, sz, u64p, sz, vp, u64p, sz, u64p, sz, u64p, sz, u64p,
                                 ui, C.c_int, vp, C.POINTER(vp)]),
    "zk_prove_table_with_aux": (C.c_int, [vp, C.POINTER(ZkCfg), u32, u64p, sz, u64p, sz, vp, u64p, sz, u64p, sz, u64p, sz, u64p,
                                          ui, C.c_int, vp, vp, C.POINTER(vp)]),
    "zk_table_aux_commit": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, sz, sz, ui, u64p, sz, u64p, sz, u64p, sz, u64p, ui,
                                      C.POINTER(vp)]),
    "zk_shard_values_to_lde": (C.c_int, [vp, u64p, sz, ui, ui, u64p, u64p]),
    "zk_shard_pack_leaf_rows": (C.c_int, [vp, u64p, sz, sz, ui, ui, u64p]),
    "zk_batch_from_parts": (C.c_int, [vp, C.POINTER(ZkCfg), sz, ui, u64p, u64p, u64p, u64p, ui, ui, C.POINTER(vp)]),
    "zk_gl_add_scalar_columns": (C.c_int, [vp, u64p, sz, sz, sz, u64p]),
    "zk_quotient_values_sharded": (C.c_int, [vp, C.POINTER(ZkCfg), u32, u64p, sz, u64p, u64p, sz, u64p, u64p, sz, ui, ui, ui,
                                             u64p, u64p, sz, u64p, sz, u64p, sz, ui, u64p]),
    "zk_quotient_commit_values": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, ui, ui, C.POINTER(vp)]),
    "zk_fri_combine_sharded": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, vp, sz, u64p, u64p, u64p]),
    "zk_fri_prove_from_values": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, vp, sz, u64p, vp, u64p, u64p]),
    "zk_fri_commit_round_sharded": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, ui, ui, u64p]),
    "zk_fri_fold_values_sharded": (C.c_int, [vp, C.POINTER(ZkCfg), u64p, ui, ui, ui, C.c_uint64, u64p, u64p]),
    "zk_fri_proof_of_work": (C.c_int, [vp, C.POINTER(ZkCfg), vp, u64p]),
    "zk_fri_initial_openings": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, u64p, sz, u64p]),
    "zk_table_proof_get": (C.c_int, [vp, vp]),
    "zk_table_proof_free": (None, [vp]),
    "zk_ctx_set_check_ctls": (C.c_int, [vp, C.c_int]),
    "zk_ctx_set_ctl_extra_looking": (C.c_int, [vp, sz, u64p, sz, sz]),
    "zk_prove_segment": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, u64p, sz, u64p, sz, ui, C.c_int, C.c_int,
                                   C.POINTER(vp)]),
    "zk_segment_proof_num_tables": (sz, [vp]),
    "zk_segment_proof_table": (vp, [vp, sz]),
    "zk_segment_proof_ctl_challenges": (sz, [vp, u64p, sz]),
    "zk_segment_proof_mem_caps": (sz, [vp, u64p, u64p, sz]),
    "zk_segment_proof_stage_ms": (sz, [vp, C.POINTER(C.c_double), sz]),
    "zk_segment_proof_free": (None, [vp]),
    "zk_plonk_circuit_create": (C.c_int, [vp, vp, vp, sz, u64p, sz, u64p, u64p, C.POINTER(vp)]),
    "zk_plonk_circuit_free": (None, [vp]),
    "zk_plonk_circuit_cap": (C.c_int, [vp, u64p]),
    "zk_plonk_prove": (C.c_int, [vp, u64p, sz, u64p, sz, C.POINTER(vp)]),
    "zk_plonk_prove_batch": (C.c_int, [vp, vp, sz, vp, sz, sz, ui, vp]),
    "zk_plonk_proof_get": (C.c_int, [vp, vp]),
    "zk_plonk_proof_free": (None, [vp]),
    "zk_keccak_generate_trace": (C.c_int, [vp, u64p, u64p, sz, ui, u64p, sz]),
    "zk_range_check_columns": (C.c_int, [vp, u64p, sz, sz, ui, sz, sz, sz, sz, C.c_uint64]),
    "zk_logic_generate_trace": (C.c_int, [vp, u64p, sz, ui, u64p, sz]),
    "zk_poseidon_generate_trace": (C.c_int, [vp, u64p, sz, vp, sz, ui, u64p, sz]),
    "zk_arithmetic_generate_trace": (C.c_int, [vp, u64p, sz, ui, u64p, sz, C.POINTER(sz)]),
    "zk_memory_trace_begin": (C.c_int, [vp, u64p, sz, u64p, sz, C.POINTER(vp)]),
    "zk_memory_gen_unpadded_length": (sz, [vp]),
    "zk_memory_gen_log_n": (ui, [vp]),
    "zk_memory_trace_finish": (C.c_int, [vp, vp, u64p, sz, u64p, sz, C.POINTER(sz)]),
    "zk_memory_gen_num_mem_after": (sz, [vp]),
    "zk_memory_gen_final_values": (C.c_int, [vp, vp, u64p]),
    "zk_memory_gen_mem_after_trace": (C.c_int, [vp, vp, ui, u64p, sz]),
    "zk_memory_gen_free": (None, [vp]),
    "zk_memory_continuation_generate_trace": (C.c_int, [vp, u64p, sz, ui, u64p, sz]),
    "zk_initial_memory_merkle_cap": (C.c_int, [vp, C.POINTER(ZkCfg), vp, sz, u64p]),
    "zk_byte_packing_generate_trace": (C.c_int, [vp, u64p, sz, ui, u64p, sz]),
    "zk_keccak_sponge_generate_trace": (C.c_int, [vp, u64p, sz, vp, sz, ui, u64p, sz]),
    "zk_comm_unique_id": (C.c_int, [vp]),
    "zk_comm_create": (C.c_int, [vp, vp, ui, ui, C.POINTER(vp)]),
    "zk_comm_create_host": (C.c_int, [vp, C.c_char_p, ui, ui, sz, C.POINTER(vp)]),
    "zk_comm_free": (None, [vp]),
    "zk_comm_rank": (ui, [vp]),
    "zk_comm_world": (ui, [vp]),
    "zk_comm_transport": (C.c_char_p, [vp]),
    "zk_comm_last_error": (C.c_char_p, [vp]),
    "zk_comm_stats": (C.c_int, [vp, u64p]),
    "zk_comm_barrier": (C.c_int, [vp]),
    "zk_comm_all_gather_host": (C.c_int, [vp, vp, sz, vp]),
    "zk_comm_broadcast_host": (C.c_int, [vp, vp, sz, ui]),
    "zk_comm_all_to_all_host": (C.c_int, [vp, vp, vp, vp, vp]),
    "zk_comm_all_to_all_device": (C.c_int, [vp, vp, vp, vp, vp]),
    "zk_comm_all_gather_device": (C.c_int, [vp, vp, sz, vp]),
    "zk_comm_last_timing": (sz, [vp, C.POINTER(C.c_double), sz, C.c_int]),
    "zk_commit_rows_sharded": (C.c_int, [vp, vp, C.POINTER(ZkCfg), u64p, sz, sz, ui, C.POINTER(vp)]),
    "zk_sharded_batch_cap": (C.c_int, [vp, u64p]),
    "zk_sharded_batch_rows": (vp, [vp]),
    "zk_sharded_batch_columns": (vp, [vp]),
    "zk_sharded_batch_free": (None, [vp]),
    "zk_prove_table_sharded": (C.c_int, [vp, vp, C.POINTER(ZkCfg), u32, u64p, sz, u64p, sz, sz, ui, vp, u64p, sz, u64p, sz, u64p,
                                         ui, C.c_int, ui, vp, C.POINTER(vp)]),
    "zk_assign_tables": (C.c_int, [vp, vp, sz, ui, vp, vp]),
    "zk_prove_segment_table_parallel": (C.c_int, [vp, vp, C.POINTER(ZkCfg), vp, sz, vp, u64p, sz, u64p, sz, ui, C.c_int, C.c_int, ui,
                                                  C.POINTER(vp)]),
    "zk_version": (C.c_char_p, []),
    "zk_device_info": (C.c_int, [C.c_int, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(sz)]),
}


def load_library():
    """dlopen the HIP library and type every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ZkStarkError(-3, f"{path} not found: build it with `python -m zk_evm_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64, and this binding
    # uses torch for device tensors and streams.  If libzkstark_hip.so (linked against /opt/rocm) were loaded
    # first, a later `import torch` would bind to that copy and fail with "No HIP GPUs are available".  Loading
    # torch first makes both share torch's runtime.  (A C / Rust caller without torch just links /opt/rocm.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
