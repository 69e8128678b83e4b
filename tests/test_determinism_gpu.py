"""Bit-reproducibility of the hot path (include/vivit_hip.h: "results are deterministic").  The multi-GPU design rests on
it -- every rank runs the replicated eigensolver stages on the same matrix and nothing is broadcast (DESIGN.md section 6;
SURVEY.md 8e "deterministic kernel => identical results, no broadcast") -- and the bf16-pipe tile product ends its
accumulation chains with float atomics (single writer per element; gemm_f32.hip:bx_flush_tiles), so it is tested where
that runs: the 256-tile SYRK at K = 65 552 and at the headline K = 401 408, `gram += gram_p`, the split-K products, the
band reduction's streaming panel product on the bf16 pipe, the convolution rules of the factor provider, the
two-stage eigensolver with vectors (persistent band reduction panels, persistent bulge chase, Q2, Q1 on the bf16 pipe) at
n = 4100 and 8192, and the reduce + select pair.  Each case runs twice in this process and once in a fresh child process
(other addresses, other workgroup placement)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def child_hashes(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("det") / "child.json")
    proc = subprocess.run([sys.executable, os.path.join(HERE, "determinism_child.py"), out], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stdout[-3000:]
    with open(out) as f:
        return json.load(f)


def _cases():
    sys.path.insert(0, HERE)
    import determinism_child as dc

    return dc


@pytest.mark.parametrize("name", ["syrk256_k65552", "syrk256_k401408", "syrk_accumulate", "gemm_nt_splitk",
                                  "symeig_two_stage_4100", "symeig_two_stage_8192", "symeig_values_8192",
                                  "symeig_reduce_select", "panel_product_bx", "conv_rules"])
def test_bit_identical_across_calls_and_processes(name, child_hashes):
    dc = _cases()
    first = dc.CASES[name]()
    second = dc.CASES[name]()
    assert first == second, f"{name}: two calls in one process differ"
    assert first == child_hashes[name], f"{name}: a fresh process computed different bits"
