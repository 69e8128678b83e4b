"""Child process of tests/test_persistent_gpu.py::test_timeout_*: VIVIT_PERSIST_FAULT=3 makes BOTH attempts of every
persistent kernel's arrival gate give up at once (csrc/device_utils.h:persist_arrive), i.e. what a card whose compute
units are held by someone else looks like after 2 x 2 s.  Checks, as JSON:
  * a solve on a copy (overwrite=False) warns, repeats itself on the launch chains and is correct;
  * an in-place solve without a backup raises PersistentKernelTimeout -- a status of its own, not "did not converge";
  * the library stays usable afterwards (the sticky failure word was taken by the failing solve).

usage: python persist_fault_child.py OUT.json
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import kernels  # noqa: E402

DEV = torch.device("cuda:0")


def matrix(n):
    g = torch.Generator().manual_seed(n)
    V = torch.randn(n, n + 50, generator=g) / n ** 0.5
    return (V @ V.T).contiguous()


def main():
    out = {}
    for n in (1000, 2500):   # one-launch tridiagonalisation | band reduction (panel QR) + bulge chase
        S = matrix(n)
        ref = np.linalg.eigvalsh(S.double().numpy())
        G = S.to(DEV)
        row = {}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            w, Z = kernels.symeig(G, eigenvectors=True)                     # copy kept -> launch-chain retry
        row["retry_warned"] = any("launch chains" in str(c.message) for c in caught)
        row["input_intact"] = bool(torch.equal(G.cpu(), S))
        row["retry_eval_err"] = float(np.abs(w.cpu().double().numpy() - ref).max() / ref.max())
        row["retry_residual"] = float((G @ Z - Z * w).abs().max() / w[-1])
        os.environ["VIVIT_PERSIST_BACKUP"] = "0"
        try:
            kernels.symeig(G.clone(), eigenvectors=False, overwrite=True)    # input destroyed, no backup -> the error
            row["raised"] = "nothing"
        except kernels.PersistentKernelTimeout as exc:
            row["raised"] = "PersistentKernelTimeout"
            row["is_runtime_error"] = isinstance(exc, RuntimeError)
            row["message_says_converge"] = "did not converge" in str(exc)
        except RuntimeError as exc:
            row["raised"] = f"RuntimeError: {exc}"
        os.environ["VIVIT_PERSIST_BACKUP"] = "1"
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            w2, _ = kernels.symeig(G.clone(), eigenvectors=False, overwrite=True)   # in place WITH backup -> retry
        row["backup_retry_warned"] = any("launch chains" in str(c.message) for c in caught)
        row["backup_eval_err"] = float(np.abs(w2.cpu().double().numpy() - ref).max() / ref.max())
        del os.environ["VIVIT_PERSIST_BACKUP"]
        with kernels.persistent_kernels(False):                            # the library is still usable
            w3, _ = kernels.symeig(G, eigenvectors=False)
        row["after_eval_err"] = float(np.abs(w3.cpu().double().numpy() - ref).max() / ref.max())
        out[str(n)] = row
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
