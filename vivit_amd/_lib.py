"""ctypes binding of libvivit_hip.so (the C ABI declared in include/vivit_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  Loading fails loudly when the
shared object is missing, and every wrapper in :mod:`vivit_amd.kernels` raises if it is handed a
tensor that does not live on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIVIT_HIP_LIB: another build of the same sources (tests/test_abi_errors_sanitized.py points it at the host-only
# AddressSanitizer + UBSan build of _build.build_host_sanitized(), which has no device code and cannot compute)
LIB_PATH = os.environ.get("VIVIT_HIP_LIB") or os.path.join(_HERE, "libvivit_hip.so")

_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_ptr = ctypes.c_void_p
_sz = ctypes.c_size_t
_int = ctypes.c_int

# name -> (restype, argtypes); mirrors include/vivit_hip.h one-to-one.
SIGNATURES = {
    "vivit_hip_abi_version": (_int, []),
    "vivit_hip_target": (ctypes.c_char_p, []),
    "vivit_hip_source_hash": (ctypes.c_char_p, []),
    "vivit_hip_status_string": (ctypes.c_char_p, [_int]),
    "vivit_persistent_kernels": (_int, [_int]),
    "vivit_take_persist_timeout": (_int, [_ptr, _ptr]),
    "vivit_gemm_split_mode": (_int, []),
    "vivit_gram_syrk_f32_workspace_bytes": (_sz, [_i64, _i64]),
    "vivit_gram_syrk_f32": (_int, [_ptr, _i64, _i64, _i64, _ptr, _i64, _f32, _f32, _ptr, _sz, _ptr]),
    "vivit_gemm_f32_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vivit_gemm_nt_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _ptr, _sz, _ptr]),
    "vivit_gemm_nn_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _ptr, _sz, _ptr]),
    "vivit_gemm_tn_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _ptr, _sz, _ptr]),
    "vivit_gram_hadamard_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _f32, _f32, _ptr]),
    "vivit_gram_hadamard_block_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _ptr]),
    "vivit_class_contract_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr]),
    "vivit_class_expand_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr]),
    "vivit_linear_weight_mjp_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr]),
    "vivit_conv2d_weight_mjp_f32": (_int, [_ptr, _ptr, _ptr] + [_i64] * 16 + [_ptr]),
    "vivit_act_jac_t_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _int, _f32, _ptr]),
    "vivit_channel_scale_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr]),
    "vivit_maxpool2d_jac_t_f32": (_int, [_ptr, _ptr, _ptr, _ptr] + [_i64] * 12 + [_ptr]),
    "vivit_avgpool2d_jac_t_f32": (_int, [_ptr, _ptr] + [_i64] * 11 + [_ptr]),
    "vivit_conv2d_jac_t_f32": (_int, [_ptr, _ptr, _ptr] + [_i64] * 15 + [_ptr]),
    "vivit_row_dot_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr]),
    "vivit_bn_eval_rules_f32": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _ptr]),
    "vivit_ce_sqrt_hessian_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _ptr]),
    "vivit_symeig_f32_workspace_bytes": (_sz, [_i64, _int]),
    "vivit_symeig_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _i64, _ptr, _sz, _ptr, _ptr]),
    "vivit_symeig_rows_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _ptr, _sz, _ptr, _ptr]),
    "vivit_symeig_reduce_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_symeig_select_f32_workspace_bytes": (_sz, [_i64, _i64]),
    "vivit_symeig_reduce_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _sz, _ptr, _ptr]),
    "vivit_symeig_select_f32": (_int, [_ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _sz, _ptr, _sz, _ptr, _ptr]),
    "vivit_sytrd_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_sytrd_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "vivit_sy2sb_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_sy2sb_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _sz, _ptr]),
    "vivit_symeig_prepare_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _sz, _ptr]),
    "vivit_sy2sb_panel_qr_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_sy2sb_panel_qr_f32": (_int, [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "vivit_symeig_banded_rows_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _sz, _ptr, _ptr]),
    "vivit_sb2st_half_bandwidth": (_int, []),
    "vivit_sb2st_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_sb2st_f32": (_int, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    "vivit_q2_apply_f32_workspace_bytes": (_sz, [_i64]),
    "vivit_q2_apply_f32": (_int, [_ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _sz, _int, _ptr]),
    "vivit_stedc_f32_workspace_bytes": (_sz, [_i64, _int]),
    "vivit_stedc_f32": (_int, [_ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _sz, _ptr, _ptr]),
    "vivit_dir_curvature_f32": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _ptr]),
    "vivit_scale_cols_rsqrt_f32": (_int, [_ptr, _ptr, _i64, _i64, _i64, _f32, _ptr]),
    "vivit_row_sqnorm_workspace_bytes": (_sz, [_i64, _i64]),
    "vivit_row_sqnorm_acc_f32": (_int, [_ptr, _ptr, _i64, _i64, _ptr, _sz, _ptr]),
    "vivit_scale_rows_rsqrt_f32": (_int, [_ptr, _ptr, _i64, _i64, _ptr]),
    "vivit_profile_begin": (_int, [_int]),
    "vivit_profile_end": (_int, [ctypes.POINTER(ctypes.c_double)]),
    "vivit_profile_stages": (_int, [ctypes.POINTER(ctypes.c_double), _int]),
    "vivit_symmetrize_lower_f32": (_int, [_ptr, _i64, _i64, _ptr]),
    "vivit_pack_lower_f32": (_int, [_ptr, _i64, _i64, _ptr, _ptr]),
    "vivit_unpack_lower_f32": (_int, [_ptr, _i64, _ptr, _i64, _ptr]),
}

ABI_VERSION = 1007  # include/vivit_hip.h of this checkout (vivit_hip_abi_version)

_lib = None


# status codes of include/vivit_hip.h
VIVIT_OK, VIVIT_E_BADARG, VIVIT_E_WORKSPACE, VIVIT_E_LAUNCH, VIVIT_E_UNSUPPORTED = 0, -1, -2, -3, -4
VIVIT_INFO_PERSIST_TIMEOUT = -1000  # device-side info word: a persistent kernel gave up (include/vivit_hip.h)


class VivitHipError(RuntimeError):
    """A libvivit_hip.so entry point returned a non-zero status (``.status``: the VIVIT_E_* code)."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


def load():
    """Load the shared library (once) and attach the prototypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). vivit_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.vivit_hip_abi_version.restype = _int
    have = lib.vivit_hip_abi_version()
    if have != ABI_VERSION:
        raise ImportError(
            f"{LIB_PATH} has ABI version {have}, this package needs {ABI_VERSION}: rebuild it "
            "(`python -c 'import __graft_entry__ as g; g.build()'`)"
        )
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _check_provenance(lib)
    _lib = lib
    return lib


def _check_provenance(lib):
    """The library must have been compiled from the sources beside this package (content hash, _build.source_hash) with
    the flags of a known build (_build.flags_digest of the product's or the host-sanitizer build's flags): a stale shipped
    binary, or a timing-only variant relinked with other defines, is refused instead of being tested by accident.
    VIVIT_HIP_ALLOW_STALE=1 turns the error into a warning (bisecting with an old library, timing a variant); a
    binary-only install (no csrc/) has nothing to compare with."""
    from . import _build

    want = _build.source_hash()
    have = lib.vivit_hip_source_hash().decode()
    known = {f"{want}-{_build.flags_digest(fl)}" for fl in (_build.FLAGS, _build.ASAN_FLAGS)}
    if want is None or have in known:
        return
    msg = (f"{LIB_PATH} was built from other sources or with other flags (library hash {have}, this tree + product flags "
           f"{want}-{_build.flags_digest(_build.FLAGS)}): rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    if os.environ.get("VIVIT_HIP_ALLOW_STALE") == "1":
        import warnings

        warnings.warn(msg)
        return
    raise ImportError(msg)


def check(status, what):
    if status != 0:
        msg = load().vivit_hip_status_string(int(status)).decode()
        raise VivitHipError(f"{what}: status {status} ({msg})", int(status))
