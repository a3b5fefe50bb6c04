"""ctypes binding of include/mcgra.h (libmcgra_hip.so, built in-tree by
``__graft_entry__.build()``).

There is no CPU fallback: importing this module without the built library, or
calling into it without a HIP device, raises.
"""
import ctypes as C
import os

# torch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  It must be the
# first HIP runtime mapped into the process, otherwise torch binds to the system one and
# loses the device; libmcgra_hip.so then resolves its libamdhip64.so.7 to the loaded copy,
# so device pointers and streams are shared with torch.
import torch  # noqa: F401  isort:skip

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCGRA_LIB_PATH") or os.path.join(_HERE, "libmcgra_hip.so")   # override: A/B builds

MAX_LAYERS = 8
MEASURES = {"HSIC": 0, "MSELoss": 1, "KL": 2, "CKA": 3, "DP": 4, "KDE": 5}

# every symbol include/mcgra.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "mcgra_version", "mcgra_last_error", "mcgra_device_count", "mcgra_sgemm",
    "mcgra_ssyrk_lower", "mcgra_ssymm_lower",
    "mcgra_get_modified_adj", "mcgra_pack_tril", "mcgra_normalize_adj", "mcgra_info_entropy",
    "mcgra_dot_product_decode", "mcgra_dot_product_decode2", "mcgra_linear_hsic", "mcgra_hsic_regular", "mcgra_hsic_normalized", "mcgra_hsic_regular2", "mcgra_hsic_normalized_cca", "mcgra_distmat", "mcgra_mmd",
    "mcgra_mmd_pxpy_pxy", "mcgra_mse", "mcgra_mutual_information",
    "mcgra_gcn_forward",
    "mcgra_attack_create", "mcgra_attack_destroy", "mcgra_attack_set_model", "mcgra_attack_set_graph",
    "mcgra_attack_set_adj_changes", "mcgra_attack_get_adj_changes", "mcgra_attack_step",
    "mcgra_attack_exchange_bytes", "mcgra_attack_bind_exchange", "mcgra_attack_shard_begin", "mcgra_attack_shard_next",
    "mcgra_attack_shard_scalars", "mcgra_attack_get_rows", "mcgra_attack_path_stats", "mcgra_attack_fused_steps", "mcgra_attack_gram_split_steps", "mcgra_attack_product_mode", "mcgra_ssymm_split_bf16", "mcgra_ssymm_split_f16",
    "mcgra_attack_monitor", "mcgra_attack_finalize", "mcgra_attack_buffer", "mcgra_attack_copy_buffer",
    "mcgra_attack_profile",
    "mcgra_attack_gemm_stats",
    "mcgra_attack_product_replay",
    "mcgra_attack_test_mutate",
    "mcgra_attack_masked_fused_steps",
    "mcgra_attack_cut_product_steps",
]


class AttackConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("nfeat", C.c_int32), ("nclass", C.c_int32), ("nlayer", C.c_int32),
        ("emb_nlayer", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("measure", C.c_int32),
        ("n_attack", C.c_int32), ("weight_sup", C.c_float), ("w", C.c_float * 10), ("lr", C.c_float),
        ("eps", C.c_float), ("num_edges", C.c_double), ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("act", C.c_int32), ("head_act", C.c_int32), ("has_self", C.c_int32), ("fin_layers", C.c_int32 * 2),
        ("shard_world", C.c_int32), ("shard_rows", C.c_int32),
    ]


class Exchange(C.Structure):
    """mcgra_exchange_t"""
    _fields_ = [("kind", C.c_int32), ("count", C.c_int32), ("offset", C.c_int64), ("offset2", C.c_int64),
                ("chunk_bytes", C.c_int64)]


class McgraError(RuntimeError):
    pass


class McgraNotSupported(NotImplementedError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MC-GRA hot path has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
    lib.mcgra_version.restype = C.c_char_p
    lib.mcgra_last_error.restype = C.c_char_p
    lib.mcgra_device_count.restype = C.c_int
    sig = {
        "mcgra_sgemm": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fp, C.c_int, fp, C.c_int,
                        C.c_float, fp, C.c_int],
        "mcgra_ssyrk_lower": [vp, C.c_int, C.c_int, C.c_float, fp, C.c_int, C.c_float, fp, C.c_int],
        "mcgra_ssymm_lower": [vp, C.c_int, C.c_int, C.c_float, fp, C.c_int, fp, C.c_int, C.c_float, fp, C.c_int],
        "mcgra_get_modified_adj": [vp, C.c_int, fp, fp, fp],
        "mcgra_pack_tril": [vp, C.c_int, fp, C.c_int, fp],
        "mcgra_normalize_adj": [vp, C.c_int, fp, fp],
        "mcgra_info_entropy": [vp, C.c_int, fp, fp],
        "mcgra_dot_product_decode": [vp, C.c_int, C.c_int, fp, fp],
        "mcgra_dot_product_decode2": [vp, C.c_int, C.c_int, fp, C.c_int, fp],
        "mcgra_linear_hsic": [vp, C.c_int, C.c_int, C.c_int, fp, fp, fp],
        "mcgra_hsic_regular": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, fp],
        "mcgra_hsic_normalized": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, fp],
        "mcgra_hsic_regular2": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_int, fp],
        "mcgra_hsic_normalized_cca": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, fp],
        "mcgra_distmat": [vp, C.c_int, C.c_int, fp, fp],
        "mcgra_mmd": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, fp],
        "mcgra_mmd_pxpy_pxy": [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, fp],
        "mcgra_mse": [vp, C.c_int64, fp, fp, fp],
        "mcgra_mutual_information": [vp, C.c_int, C.c_int, fp, fp, fp, fp, fp],
        "mcgra_gcn_forward": [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), fp, fp, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_void_p), fp, fp, C.c_int, C.c_int, fp, fp],
        "mcgra_attack_create": [C.POINTER(C.c_void_p), C.POINTER(AttackConfig)],
        "mcgra_attack_destroy": [vp],
        "mcgra_attack_set_model": [vp, vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), fp, fp, C.POINTER(C.c_void_p)],
        "mcgra_attack_set_graph": [vp, vp, fp, fp, fp, fp, ip, ip],
        "mcgra_attack_set_adj_changes": [vp, vp, fp],
        "mcgra_attack_get_adj_changes": [vp, vp, fp],
        "mcgra_attack_step": [vp, vp, fp, C.POINTER(C.c_double)],
        "mcgra_attack_bind_exchange": [vp, vp, C.c_int64],
        "mcgra_attack_shard_begin": [vp, vp, C.c_int, C.c_int],
        "mcgra_attack_shard_next": [vp, vp, C.POINTER(Exchange)],
        "mcgra_attack_shard_scalars": [vp, vp, C.POINTER(C.c_double)],
        "mcgra_attack_get_rows": [vp, vp, fp],
        "mcgra_attack_product_mode": [vp],
        "mcgra_ssymm_split_bf16": [vp, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp, C.c_int],
        "mcgra_ssymm_split_f16": [vp, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp, C.c_int],
        "mcgra_attack_path_stats": [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
        "mcgra_attack_monitor": [vp, vp, fp, C.POINTER(C.c_double)],
        "mcgra_attack_finalize": [vp, vp, C.c_int, fp, fp, fp, fp],
        "mcgra_attack_buffer": [vp, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int)],
        "mcgra_attack_copy_buffer": [vp, vp, C.c_char_p, fp, C.c_int],
        "mcgra_attack_profile": [vp, C.c_int],
        "mcgra_attack_gemm_stats": [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "mcgra_attack_product_replay": [vp, vp, C.c_int, C.POINTER(C.c_double)],
        "mcgra_attack_test_mutate": [vp, C.c_int],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mcgra_attack_fused_steps.argtypes = [vp]
    lib.mcgra_attack_fused_steps.restype = C.c_longlong
    lib.mcgra_attack_masked_fused_steps.argtypes = [vp]
    lib.mcgra_attack_masked_fused_steps.restype = C.c_longlong
    lib.mcgra_attack_cut_product_steps.argtypes = [vp]
    lib.mcgra_attack_cut_product_steps.restype = C.c_longlong
    lib.mcgra_attack_gram_split_steps.argtypes = [vp]
    lib.mcgra_attack_gram_split_steps.restype = C.c_longlong
    lib.mcgra_attack_exchange_bytes.argtypes = [vp]
    lib.mcgra_attack_exchange_bytes.restype = C.c_int64
    return lib


lib = _load()


def check(rc: int):
    if rc == 0:
        return
    msg = lib.mcgra_last_error().decode("utf-8", "replace")
    if rc == -3:
        raise McgraNotSupported(msg)
    raise McgraError(f"mcgra error {rc}: {msg}")


def require_device():
    n = lib.mcgra_device_count()
    if n <= 0:
        raise McgraError("no HIP device visible: the MC-GRA hot path runs on MI355X only (no CPU fallback); "
                         + lib.mcgra_last_error().decode("utf-8", "replace"))
    return n
