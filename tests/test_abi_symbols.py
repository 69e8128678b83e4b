"""The shared library loads and exports every symbol include/vivit_hip.h declares (no GPU needed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported_and_bound():
    from vivit_amd import _lib

    header = open(os.path.join(ROOT, "include", "vivit_hip.h")).read()
    declared = set(re.findall(r"\b(vivit_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in vivit_hip.h but not exported"
    missing = declared - set(_lib.SIGNATURES)
    assert not missing, f"no ctypes prototype for {missing}"
    assert lib.vivit_hip_abi_version() == _lib.ABI_VERSION
    assert lib.vivit_hip_target() == b"gfx950"
    assert b"workspace" in lib.vivit_hip_status_string(-2)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "vivit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"
