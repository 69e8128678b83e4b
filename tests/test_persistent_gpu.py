"""The persistent kernels against the launch chains they replace.

* ``sb2st_persist_kernel`` (csrc/sb2st.hip) runs the same tasks with the same arithmetic as one launch per wavefront step:
  d, e, every reflector and every tau must be BIT-identical -- any stale band row (a missed dependency, a cache line
  served from the wrong L2) changes them.
* ``qr_persist_kernel`` (csrc/sy2sb.hip) sums the column partials in another order than the launch chain: the band agrees
  entry by entry to rounding (full-rank input: reflector signs are fixed by the data) and has the spectrum of the input.
* ``trd_persist_kernel`` (csrc/sytrd_persist.hip) is an unblocked right-looking reduction, ``sytrd.hip`` a blocked one:
  different rounding, same tridiagonal matrix up to signs -- the spectra of (d, e) must agree to fp32 accuracy and with
  the fp64 spectrum of the input.

The knobs are read once per process, hence the child processes (as tests/test_gram_precision_gpu.py).
Reference semantics: the eigenvalues ``Tensor.symeig`` returns at vivit/linalg/eigh.py:248-250."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.linalg
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
HERE = os.path.dirname(os.path.abspath(__file__))


def _child(tmp, tag, **env):
    out = tmp / f"persist_{tag}.json"
    subprocess.run([sys.executable, os.path.join(HERE, "persist_child.py"), str(out)], env=dict(os.environ, **env), check=True,
                   timeout=600)
    return json.loads(out.read_text())


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("persist")
    return {"on": _child(tmp, "on", VIVIT_SB2ST_PERSIST="1", VIVIT_SYTRD_PERSIST="1", VIVIT_QR_PERSIST="1"),
            "off": _child(tmp, "off", VIVIT_SB2ST_PERSIST="0", VIVIT_SYTRD_PERSIST="0", VIVIT_QR_PERSIST="0"),
            # the first attempt of every arrival gate gives up at once: the retry queued behind it does the work
            "retry": _child(tmp, "retry", VIVIT_SB2ST_PERSIST="1", VIVIT_SYTRD_PERSIST="1", VIVIT_QR_PERSIST="1",
                            VIVIT_PERSIST_FAULT="1")}


def test_sb2st_persistent_is_bit_identical_to_the_launch_chain(runs):
    assert runs["on"]["sb2st"] == runs["off"]["sb2st"]


@pytest.mark.parametrize("n", [193, 256, 300, 777, 1024, 1280, 1500, 2048])   # (above 1280: all 256 CUs, agent-scope exchanges)
def test_sytrd_persistent_spectrum(runs, n):
    g = torch.Generator().manual_seed(n)
    M = torch.randn(n, n, generator=g)
    S = (M + M.T).double().numpy()
    ref = np.linalg.eigvalsh(S)
    scale = np.abs(ref).max()
    w = {}
    for tag in ("on", "off"):
        r = runs[tag]["sytrd"][str(n)]
        d, e = np.array(r["d"], dtype=np.float64), np.array(r["e"], dtype=np.float64)[: n - 1]
        w[tag] = scipy.linalg.eigvalsh_tridiagonal(d, e)
        assert np.abs(w[tag] - ref).max() <= 5e-6 * scale, tag
    assert np.abs(w["on"] - w["off"]).max() <= 5e-6 * scale
    assert runs["on"]["sytrd"][str(n)] != runs["off"]["sytrd"][str(n)]   # (the knob did select another kernel)


def _band_dense(AB):
    NB = 64
    n = AB.shape[0]
    B = np.zeros((n, n))
    for i in range(n):
        lo = max(0, i - NB)
        B[i, lo: i + 1] = AB[i, lo - i + 2 * NB: 2 * NB + 1]
    return B + np.tril(B, -1).T


@pytest.mark.parametrize("n", [200, 1000, 2500])
def test_panel_qr_persistent_band(runs, n):
    g = torch.Generator().manual_seed(n)
    M = torch.randn(n, n, generator=g)
    ref = np.linalg.eigvalsh(((M + M.T) / 2).double().numpy())
    scale = np.abs(ref).max()
    res = {tag: torch.load(runs[tag]["sy2sb"][str(n)]) for tag in ("on", "off")}
    for tag in ("on", "off"):
        w = np.linalg.eigvalsh(_band_dense(res[tag]["AB"].double().numpy()))
        assert np.abs(w - ref).max() <= 5e-6 * scale, tag
    assert float((res["on"]["AB"] - res["off"]["AB"]).abs().max()) <= 1e-4 * scale
    assert float((res["on"]["tau1"] - res["off"]["tau1"]).abs().max()) <= 2e-3
    assert not torch.equal(res["on"]["AB"], res["off"]["AB"])   # (the knob did select another kernel)


def test_aborted_first_attempt_is_retried_on_the_device(runs):
    """VIVIT_PERSIST_FAULT=1: attempt 0 of every persistent kernel aborts at its arrival gate (nothing written), the second
    launch queued behind it runs: every output is BIT-identical to the undisturbed run."""
    assert runs["retry"]["sb2st"] == runs["on"]["sb2st"]
    assert runs["retry"]["sytrd"] == runs["on"]["sytrd"]
    for n in runs["on"]["sy2sb"]:
        a, b = torch.load(runs["on"]["sy2sb"][n]), torch.load(runs["retry"]["sy2sb"][n])
        assert torch.equal(a["AB"], b["AB"]) and torch.equal(a["tau1"], b["tau1"]), n


def test_timeout_has_its_own_status_and_the_host_retries_on_the_launch_chains(tmp_path):
    """VIVIT_PERSIST_FAULT=3: both attempts give up -> info = VIVIT_INFO_PERSIST_TIMEOUT (include/vivit_hip.h), not the
    non-finite-input status; a solve that still has its input repeats itself on the launch chains (reference semantics
    kept: the eigenvalues ``Tensor.symeig`` returns, vivit/utils/eig.py:35-46)."""
    out = tmp_path / "fault.json"
    subprocess.run([sys.executable, os.path.join(HERE, "persist_fault_child.py"), str(out)],
                   env=dict(os.environ, VIVIT_PERSIST_FAULT="3"), check=True, timeout=600)
    res = json.loads(out.read_text())
    for n, row in res.items():
        assert row["retry_warned"] and row["input_intact"], (n, row)
        assert row["retry_eval_err"] <= 1e-5 and row["retry_residual"] <= 1e-5, (n, row)
        assert row["raised"] == "PersistentKernelTimeout" and row["is_runtime_error"] and not row["message_says_converge"], (n, row)
        assert row["backup_retry_warned"] and row["backup_eval_err"] <= 1e-5, (n, row)
        assert row["after_eval_err"] <= 1e-5, (n, row)
