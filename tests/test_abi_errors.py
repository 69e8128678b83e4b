"""The error half of the C ABI (include/vivit_hip.h: "return value: 0 = ok; <0 = bad argument"), provoked through ctypes.

No GPU is needed and none is used: every call below must be REFUSED by the host-side argument checks before anything is
enqueued (a missing check shows up here as VIVIT_E_LAUNCH on a box without a device, or as a launch on one with).  The
pointers handed over are fake non-null addresses, never dereferenced on the host.  Mirrors the reference's error
behaviour for bad input (vivit/utils/eig.py:37-40, vivit/utils/checks.py) one level down, where the Python guards of
vivit_amd.kernels normally hide it.
"""
import ctypes

import pytest

from vivit_amd import _lib

OK, BADARG, WORKSPACE, LAUNCH, UNSUPPORTED = 0, -1, -2, -3, -4
P = 0x7F0000001000  # fake device pointer (16-byte aligned, never touched by the host code)
Q = P + 0x100000
R = P + 0x200000
W = P + 0x400000    # fake workspace
BIG = 1 << 40       # "plenty" of workspace bytes


def lib():
    return _lib.load()


def call(name, *args):
    return getattr(lib(), name)(*args)


def test_status_strings_and_codes():
    L = lib()
    assert (_lib.VIVIT_OK, _lib.VIVIT_E_BADARG, _lib.VIVIT_E_WORKSPACE, _lib.VIVIT_E_LAUNCH, _lib.VIVIT_E_UNSUPPORTED) == (0, -1, -2, -3, -4)
    assert L.vivit_hip_status_string(OK) == b"ok"
    assert b"bad argument" in L.vivit_hip_status_string(BADARG)
    assert b"workspace" in L.vivit_hip_status_string(WORKSPACE)
    assert b"launch" in L.vivit_hip_status_string(LAUNCH)
    assert b"unsupported" in L.vivit_hip_status_string(UNSUPPORTED)
    assert b"unknown" in L.vivit_hip_status_string(-77)
    with pytest.raises(_lib.VivitHipError) as exc:
        _lib.check(UNSUPPORTED, "probe")
    assert exc.value.status == _lib.VIVIT_E_UNSUPPORTED and "status -4" in str(exc.value)


# (entry point, arguments, expected status): one line per refused call
n, p = 512, 2048
SYM_WS = None  # filled lazily (needs the library)

BADARG_CASES = [
    # K1 Gram SYRK: null A, null G, lda < p, ldg < n, negative sizes
    ("vivit_gram_syrk_f32", (None, n, p, p, Q, n, 1.0, 0.0, W, BIG, None)),
    ("vivit_gram_syrk_f32", (P, n, p, p, None, n, 1.0, 0.0, W, BIG, None)),
    ("vivit_gram_syrk_f32", (P, n, p, p - 1, Q, n, 1.0, 0.0, W, BIG, None)),
    ("vivit_gram_syrk_f32", (P, n, p, p, Q, n - 1, 1.0, 0.0, W, BIG, None)),
    ("vivit_gram_syrk_f32", (P, -1, p, p, Q, n, 1.0, 0.0, W, BIG, None)),
    ("vivit_gram_syrk_f32", (P, n, -5, p, Q, n, 1.0, 0.0, W, BIG, None)),
    # K2 / K5-K9 GEMMs
    ("vivit_gemm_nt_f32", (None, Q, R, 64, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, None, R, 64, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, Q, None, 64, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, Q, R, 64, 64, 64, 63, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, Q, R, 64, 64, 64, 64, 63, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, Q, R, 64, 64, 64, 64, 64, 63, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nt_f32", (P, Q, R, 64, 64, -1, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nn_f32", (None, Q, R, 64, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_nn_f32", (P, Q, R, 64, 96, 64, 64, 95, 96, 1.0, 0.0, W, BIG, None)),    # ldb < n (B is [k, n])
    ("vivit_gemm_nn_f32", (P, Q, R, -3, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_tn_f32", (P, None, R, 64, 64, 64, 64, 64, 64, 1.0, 0.0, W, BIG, None)),
    ("vivit_gemm_tn_f32", (P, Q, R, 96, 64, 32, 95, 64, 64, 1.0, 0.0, W, BIG, None)),    # lda < m (A is [k, m])
    # K1' factorised Gram, class contractions
    ("vivit_gram_hadamard_f32", (None, Q, R, 2, 8, 1.0, 0.0, None)),
    ("vivit_gram_hadamard_f32", (P, Q, R, -2, 8, 1.0, 0.0, None)),
    ("vivit_gram_hadamard_block_f32", (P, Q, None, 2, 8, 2, 8, 16, 1.0, 0.0, None)),
    ("vivit_gram_hadamard_block_f32", (P, Q, R, 2, 8, 2, 8, 15, 1.0, 0.0, None)),        # ldg < Cc * Nc
    ("vivit_class_contract_f32", (P, Q, None, 2, 3, 4, 5, None)),
    ("vivit_class_contract_f32", (P, Q, R, 2, -3, 4, 5, None)),
    ("vivit_class_expand_f32", (None, Q, R, 2, 3, 4, 5, None)),
    # f1 factor rules
    ("vivit_linear_weight_mjp_f32", (None, Q, R, 2, 3, 4, 5, None)),
    ("vivit_linear_weight_mjp_f32", (P, Q, R, 2, 3, -4, 5, None)),
    ("vivit_conv2d_weight_mjp_f32", (None, Q, R, 6, 3, 2, 8, 8, 4, 3, 3, 6, 6, 1, 1, 0, 0, 1, 1, None)),
    ("vivit_conv2d_weight_mjp_f32", (P, Q, R, 6, 3, 2, 8, 8, 4, 3, 3, 6, 6, 0, 1, 0, 0, 1, 1, None)),   # stride 0
    ("vivit_act_jac_t_f32", (None, Q, R, 2, 64, 0, 0.0, None)),
    ("vivit_act_jac_t_f32", (P, Q, R, 2, 64, 99, 0.0, None)),                              # unknown activation kind
    ("vivit_channel_scale_f32", (P, None, R, 4, 3, 5, None)),
    ("vivit_maxpool2d_jac_t_f32", (P, Q, R, None, 2, 6, 8, 8, 4, 4, 2, 2, 2, 2, 0, 0, None)),  # no index scratch
    ("vivit_avgpool2d_jac_t_f32", (P, None, 12, 8, 8, 4, 4, 2, 2, 2, 2, 0, 0, None)),
    ("vivit_conv2d_jac_t_f32", (P, None, R, 6, 2, 8, 8, 4, 3, 3, 6, 6, 1, 1, 0, 0, 1, 1, None)),
    ("vivit_row_dot_f32", (None, None, R, 8, 8, 16, None)),
    ("vivit_bn_eval_rules_f32", (None, Q, R, R, R, R, 12, 6, 3, 16, None, None, None)),            # no factor
    ("vivit_bn_eval_rules_f32", (P, Q, None, R, R, R, 12, 6, 3, 16, None, None, None)),            # scaled output without a scale
    ("vivit_bn_eval_rules_f32", (P, Q, R, None, None, None, 12, 6, 3, 16, None, None, None)),      # nothing to compute
    ("vivit_bn_eval_rules_f32", (P, Q, R, R, R, R, 12, 6, 3, 16, R, None, None)),                  # mean without rstd
    ("vivit_bn_eval_rules_f32", (P, Q, R, R, R, R, 13, 6, 3, 16, None, None, None)),               # rows not a multiple of C
    ("vivit_take_persist_timeout", (None, None)),                                                  # no info word
    ("vivit_ce_sqrt_hessian_f32", (None, None, R, 4, 3, 3, 1.0, None)),
    ("vivit_ce_sqrt_hessian_f32", (P, None, R, 4, 3, 2, 1.0, None)),                       # exact factor needs V == C
    # K3 / K4 eigensolver
    ("vivit_symeig_f32", (None, n, n, Q, R, n, W, BIG, P, None)),
    ("vivit_symeig_f32", (P, n, n - 1, Q, R, n, W, BIG, P, None)),
    ("vivit_symeig_f32", (P, n, n, None, R, n, W, BIG, P, None)),
    ("vivit_symeig_f32", (P, n, n, Q, R, n - 1, W, BIG, P, None)),
    ("vivit_symeig_f32", (P, n, n, Q, R, n, W, BIG, None, None)),                          # no info word
    ("vivit_symeig_f32", (P, -1, n, Q, R, n, W, BIG, P, None)),
    ("vivit_symeig_rows_f32", (P, n, n, Q, None, n, 0, 8, W, BIG, P, None)),
    ("vivit_symeig_rows_f32", (P, n, n, Q, R, n, 9, 8, W, BIG, P, None)),                  # row_end < row_begin
    ("vivit_symeig_rows_f32", (P, n, n, Q, R, n, 0, n + 1, W, BIG, P, None)),
    ("vivit_symeig_reduce_f32", (None, n, n, Q, W, BIG, P, None)),
    ("vivit_symeig_reduce_f32", (P, n, n - 1, Q, W, BIG, P, None)),
    ("vivit_symeig_select_f32", (None, n, n, Q, 4, R, n, W, BIG, W, BIG, P, None)),
    ("vivit_symeig_select_f32", (P, n, n, None, 4, R, n, W, BIG, W, BIG, P, None)),        # K > 0 without an index list
    ("vivit_symeig_select_f32", (P, n, n, Q, n + 1, R, n, W, BIG, W, BIG, P, None)),
    ("vivit_sytrd_f32", (P, 2, 2, Q, R, W, W, BIG, None)),                                 # n < 3
    ("vivit_sytrd_f32", (P, n, n, None, R, W, W, BIG, None)),
    ("vivit_sy2sb_f32", (None, n, n, Q, R, W, BIG, None)),
    ("vivit_sy2sb_f32", (P, n, n - 1, Q, R, W, BIG, None)),
    ("vivit_symeig_prepare_f32", (P, n, n, None, W, BIG, None)),
    ("vivit_sy2sb_panel_qr_f32", (None, 256, Q, 256, R, R, R, W, BIG, None)),
    ("vivit_sy2sb_panel_qr_f32", (P, 256, Q, 255, R, R, R, W, BIG, None)),                 # ldv < mp
    ("vivit_symeig_banded_rows_f32", (P, n, n, None, Q, R, R, n, 0, 8, W, BIG, P, None)),
    ("vivit_symeig_banded_rows_f32", (P, 100, 100, Q, Q, R, R, 100, 0, 8, W, BIG, P, None)),  # n <= 2 NB
    ("vivit_sb2st_f32", (None, n, Q, R, R, W, BIG, None)),
    ("vivit_sb2st_f32", (P, 0, Q, R, R, W, BIG, None)),
    ("vivit_q2_apply_f32", (None, n, 16, n, Q, n, R, W, BIG, 0, None)),
    ("vivit_q2_apply_f32", (P, n - 1, 16, n, Q, n, R, W, BIG, 0, None)),                    # ldz < n
    ("vivit_q2_apply_f32", (P, n, 16, n, Q, n, None, W, BIG, 1, None)),
    ("vivit_q2_apply_f32", (P, n, 16, n, Q, n, R, W, BIG, 2, None)),                        # unknown mode
    ("vivit_stedc_f32", (None, Q, n, R, None, n, W, BIG, P, None)),
    ("vivit_stedc_f32", (P, Q, n, R, R, n - 1, W, BIG, P, None)),
    # K5 / K6 / K10 epilogues, packed triangles
    ("vivit_dir_curvature_f32", (P, None, R, 2, 8, 4, 1.0, None)),
    ("vivit_scale_cols_rsqrt_f32", (P, Q, 8, 4, 3, 1.0, None)),                            # ldx < K
    ("vivit_row_sqnorm_acc_f32", (None, Q, 4, 100, W, BIG, None)),
    ("vivit_scale_rows_rsqrt_f32", (P, None, 4, 100, None)),
    ("vivit_symmetrize_lower_f32", (None, n, n, None)),
    ("vivit_symmetrize_lower_f32", (P, n, n - 1, None)),
    ("vivit_pack_lower_f32", (P, n, n, None, None)),
    ("vivit_unpack_lower_f32", (None, n, P, n, None)),
]


@pytest.mark.parametrize("name,args", BADARG_CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(BADARG_CASES)])
def test_bad_arguments_are_refused(name, args):
    assert call(name, *args) == BADARG


def _symeig_ws(nn, vec):
    return lib().vivit_symeig_f32_workspace_bytes(nn, vec)


def test_workspace_one_byte_short_is_refused():
    L = lib()
    nn = 512
    for vec, Z in ((0, None), (1, R)):
        need = L.vivit_symeig_f32_workspace_bytes(nn, vec)
        assert need > 0
        assert L.vivit_symeig_f32(P, nn, nn, Q, Z, nn, W, need - 1, P, None) == WORKSPACE
        assert L.vivit_symeig_f32(P, nn, nn, Q, Z, nn, None, need, P, None) == WORKSPACE
    need = L.vivit_symeig_f32_workspace_bytes(nn, 1)
    assert L.vivit_symeig_rows_f32(P, nn, nn, Q, R, nn, 0, 8, W, need - 1, P, None) == WORKSPACE
    need = L.vivit_symeig_reduce_f32_workspace_bytes(nn)
    assert L.vivit_symeig_reduce_f32(P, nn, nn, Q, W, need - 1, P, None) == WORKSPACE
    sel = L.vivit_symeig_select_f32_workspace_bytes(nn, 4)
    assert L.vivit_symeig_select_f32(P, nn, nn, Q, 4, R, nn, W, need - 1, W, sel, P, None) == WORKSPACE   # state short
    assert L.vivit_symeig_select_f32(P, nn, nn, Q, 4, R, nn, W, need, W, sel - 1, P, None) == WORKSPACE   # scratch short
    need = L.vivit_sytrd_f32_workspace_bytes(nn)
    assert L.vivit_sytrd_f32(P, nn, nn, Q, R, R, W, need - 1, None) == WORKSPACE
    need = L.vivit_sy2sb_f32_workspace_bytes(nn)
    assert L.vivit_sy2sb_f32(P, nn, nn, Q, R, W, need - 1, None) == WORKSPACE
    need = L.vivit_sb2st_f32_workspace_bytes(nn)
    assert L.vivit_sb2st_f32(P, nn, Q, R, R, W, need - 1, None) == WORKSPACE
    need = L.vivit_stedc_f32_workspace_bytes(nn, 1)
    assert L.vivit_stedc_f32(P, Q, nn, R, R, nn, W, need - 1, P, None) == WORKSPACE
    need = L.vivit_sy2sb_panel_qr_f32_workspace_bytes(256)
    assert L.vivit_sy2sb_panel_qr_f32(P, 256, Q, 256, R, R, R, W, need - 1, None) == WORKSPACE
    assert L.vivit_symeig_prepare_f32(P, nn, nn, Q, W, 8 * nn + 255, None) == WORKSPACE
    need = L.vivit_row_sqnorm_workspace_bytes(4, 100000)
    assert need > 0
    assert L.vivit_row_sqnorm_acc_f32(P, Q, 4, 100000, W, need - 1, None) == WORKSPACE
    need = L.vivit_q2_apply_f32_workspace_bytes(nn)
    assert L.vivit_q2_apply_f32(P, nn, 16, nn, Q, nn, R, W, need - 1, 0, None) == WORKSPACE
    need = L.vivit_symeig_f32_workspace_bytes(nn, 1)
    assert L.vivit_symeig_banded_rows_f32(P, nn, nn, Q, Q, R, R, nn, 0, 8, W, need - 1, P, None) == WORKSPACE


def test_unsupported_sizes_are_refused():
    L = lib()
    small = 192   # single-workgroup sizes: the row-range and two-phase entries send the caller to vivit_symeig_f32
    assert L.vivit_symeig_rows_f32(P, small, small, Q, R, small, 0, 8, W, BIG, P, None) == UNSUPPORTED
    assert L.vivit_symeig_reduce_f32(P, small, small, Q, W, BIG, P, None) == UNSUPPORTED
    assert L.vivit_symeig_select_f32(P, small, small, Q, 4, R, small, W, BIG, W, BIG, P, None) == UNSUPPORTED
    huge = (1 << 31) // 8 + 8   # beyond the 32-bit index range of the multi-kernel solver
    assert L.vivit_symeig_f32(P, huge, huge, Q, None, huge, W, BIG << 8, P, None) == UNSUPPORTED
    # sliding-window Q2: n % 4 != 0
    assert L.vivit_q2_apply_f32(P, 512, 16, 510, Q, 510, R, W, BIG, 1, None) == UNSUPPORTED
    # conv weight rule: more rows x tiles than one launch can index
    assert L.vivit_conv2d_weight_mjp_f32(P, Q, R, 1 << 40, 1 << 20, 2, 8, 8, 4, 3, 3, 6, 6, 1, 1, 0, 0, 1, 1, None) == UNSUPPORTED


def test_empty_problems_are_ok_without_a_device():
    """Zero-sized problems return VIVIT_OK before anything is launched (the reference's empty-tensor einsums)."""
    L = lib()
    assert L.vivit_gram_syrk_f32(P, 0, 128, 128, Q, 0, 1.0, 0.0, None, 0, None) == OK
    assert L.vivit_gemm_nt_f32(P, Q, R, 0, 64, 64, 64, 64, 64, 1.0, 0.0, None, 0, None) == OK
    assert L.vivit_gram_hadamard_f32(P, Q, R, 0, 8, 1.0, 0.0, None) == OK
    assert L.vivit_scale_rows_rsqrt_f32(P, Q, 0, 10, None) == OK
    assert L.vivit_symmetrize_lower_f32(P, 0, 0, None) == OK
