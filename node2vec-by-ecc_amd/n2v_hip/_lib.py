"""ctypes binding of libn2v_hip.so (the C-ABI declared in include/n2v_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

# torch first: it ships its own libamdhip64; loading ours before torch's would put a second,
# device-less HIP runtime into the process (kernel launches then fail with "no ROCm-capable
# device").  With torch loaded first the library binds to the runtime torch initialised.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("N2V_HIP_LIB") or os.path.join(_HERE, "libn2v_hip.so")  # override: A/B builds

N2V_OK = 0
N2V_STATUS_ZERO_NORM = 1
RNG_UNIFORMS = 0
RNG_PHILOX = 1
RNG_UNIFORMS_TILED = 2

# name -> (restype, argtypes); mirrors include/n2v_hip.h, n2v_bine.h and n2v_sim.h one to one
_i64, _i32, _u64, _f64, _ptr = C.c_int64, C.c_int32, C.c_uint64, C.c_double, C.c_void_p
BINE_SEQUENTIAL = 0
BINE_PARALLEL = 1
BINE_PARALLEL_STORE = 2
SIGNATURES = {
    "n2v_abi_version": (C.c_int, []),
    "n2v_last_error": (C.c_char_p, []),
    "n2v_alias_setup_tables": (C.c_int, [_i64, _ptr, _ptr, _ptr]),
    "n2v_build_node_tables": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "n2v_build_edge_tables": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _f64, _f64, _i32, _ptr, _ptr, _i64, _i64,
                                        _ptr, _ptr, _ptr]),
    "n2v_build_edge_tables_wave": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _f64, _f64, _i32, _ptr, _ptr, _i64, _i64,
                                             _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr]),
    "n2v_edge_tables_wave_scratch_bytes": (C.c_int64, [_i64]),
    "n2v_build_edge_recs": (C.c_int, [_i64, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr]),
    "n2v_walk": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _ptr, _ptr,
                           _u64, _ptr, _ptr, _ptr]),
    "n2v_mt19937_jump_host": (C.c_int, [_ptr, _i64, _i32, _ptr]),
    "n2v_mt19937_jump_polys_host": (C.c_int, [_i64, _i32, _ptr]),
    "n2v_mt19937_jump_device": (C.c_int, [_ptr, _i32, _ptr, _i32, _ptr]),
    "n2v_mt19937_fill": (C.c_int, [_ptr, _i32, _i32, _i64, _i64, _ptr, _ptr, _ptr]),
    "n2v_mt19937_fill_tiled": (C.c_int, [_ptr, _i32, _i32, _i64, _i64, _i32, _ptr, _ptr, _ptr]),
    "n2v_build_fat_slots": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "n2v_walk_fat": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _ptr, _ptr, _i64, _u64,
                               _ptr, _ptr, _ptr]),
    "n2v_walk_on_the_fly": (C.c_int, [_ptr, _ptr, _ptr, _f64, _f64, _i32, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _i32,
                                      _i32, _ptr, _ptr, _u64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "n2v_walk_otf_lds_slots": (C.c_int32, []),
    "n2v_walk_otf_max_waves": (C.c_int32, []),
    "n2v_walk_hybrid": (C.c_int, [_ptr, _ptr, _ptr, _f64, _f64, _i32, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64,
                                  _i64, _i32, _i32, _ptr, _ptr, _u64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "n2v_sgns_init": (C.c_int, [_ptr, _ptr, _i64, _i32, _i32, _u64, _ptr]),
    "n2v_build_neg_lut": (C.c_int, [_ptr, _i64, _i32, _ptr, _ptr]),
    "n2v_sgns_train": (C.c_int, [_ptr, _ptr, _i64, _i32, _ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _ptr, _ptr,
                                 _ptr, _i32, C.c_float, C.c_float, _i64, _i64, _i64, _i64, _u64, _u64, _ptr, _i32,
                                 _i32, _i32, _ptr, _ptr]),
    "n2v_sgns_train_span": (C.c_int, [_ptr, _ptr, _i64, _i32, _ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _ptr, _ptr,
                                      _ptr, _i32, C.c_float, C.c_float, _i64, _i64, _i64, _u64, _ptr, _i32, _i32, _i32,
                                      _ptr, _i32, _i32, _i64, _i64, _ptr, _ptr]),
    "n2v_sgns_default_blocks": (C.c_int32, [_i64, _i32]),
    "n2v_merge_snapshot": (C.c_int, [_ptr, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "n2v_merge_hot_apply": (C.c_int, [_ptr, _ptr, _ptr, _i32, _ptr, _ptr, _i64, _ptr, _i32, _ptr]),
    "n2v_merge_flush": (C.c_int, [_ptr, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _ptr, _i32, _ptr]),
    "n2v_merge_pack_rows": (C.c_int, [_ptr, _ptr, _i32, _ptr, _i64, _ptr, _i32, _ptr]),
    "n2v_tsum_pack": (C.c_int, [_ptr, _i32, _i32, _ptr, _i32, _ptr]),
    "n2v_tsum_apply": (C.c_int, [_ptr, _i32, _i32, _ptr, _i32, _ptr]),
    # include/n2v_bine.h
    "n2v_bine_spmv": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "n2v_bine_hits_normalise": (C.c_int, [_i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "n2v_bine_walk_counts": (C.c_int, [_ptr, _i64, _i64, _i32, _i32, _ptr, _ptr, _ptr]),
    "n2v_bine_walk_lengths": (C.c_int, [_ptr, _ptr, _ptr, _i64, _i64, _f64, _i32, _u64, _ptr, _ptr]),
    "n2v_bine_walk": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _u64, _ptr, _ptr]),
    "n2v_bine_neg_pools": (C.c_int, [_ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _f64, _u64, _ptr, _ptr]),
    "n2v_lsh_sha1_labels": (C.c_int, [_ptr, _i32, _ptr, _i64, _ptr, _ptr]),
    "n2v_lsh_minhash": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr]),
    "n2v_lsh_forest_query": (C.c_int, [_ptr, _ptr, _ptr, _i64, _i32, _ptr, _ptr, _ptr]),
    "n2v_lsh_leader_round": (C.c_int, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr]),
    "n2v_lsh_pools": (C.c_int, [_ptr, _ptr, _i32, _ptr, _i64, _i32, _i32, _u64, _i32, _ptr, _i64, _i32, _ptr, _ptr]),
    "n2v_bine_init": (C.c_int, [_ptr, _ptr, _i64, _i32, _i32, _u64, _ptr]),
    "n2v_bine_train_pass": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _ptr,
                                      _ptr, _ptr, _ptr, _i32, _i32, _i32, _f64, _f64, _f64, _ptr, _i32, _u64, _u64,
                                      _i32, _i32, _ptr]),
    "n2v_bine_lambda_step": (C.c_int, [_ptr, _f64, _ptr]),
    # include/n2v_sim.h
    "n2v_sim_prepare": (C.c_int, [_ptr, _i32, _i32, _ptr, _i64, _i32, _ptr, _i32, _ptr]),
    "n2v_sim_block": (C.c_int, [_ptr, _i64, _i64, _ptr, _i64, _i32, _i32, _i64, _ptr, _i64, _ptr]),
    "n2v_sim_topk_scan": (C.c_int, [_ptr, _i64, _i64, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr,
                                    _i64, _ptr, _ptr]),
    "n2v_sim_rows_count": (C.c_int, [_ptr, _i64, _i64, _i64, C.c_float, _ptr, _ptr]),
    "n2v_sim_rows_fill": (C.c_int, [_ptr, _i64, _i64, _i64, C.c_float, _ptr, _ptr, _ptr, _ptr]),
    "n2v_sim_rows_topk": (C.c_int, [_ptr, _i64, _i64, _i64, _i32, _ptr, _ptr, _ptr]),
}

_lib = None


class N2VError(RuntimeError):
    pass


def load():
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "libn2v_hip.so not found at %s — build it with `python __graft_entry__.py` "
            "(or `make -C node2vec-by-ecc_amd/csrc`); there is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != N2V_OK:
        msg = load().n2v_last_error()
        raise N2VError("libn2v_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device (or host) pointer of a torch tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "non-contiguous tensor passed to the C-ABI"
    return t.data_ptr()


def stream_ptr(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
