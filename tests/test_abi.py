"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/smokehip.h
declares (no compute calls: there is no GPU here), and the product refuses to run without a ROCm device."""
import os
import re

import pytest

from smokephysai_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "smokehip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/smokehip.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "ctypes binding and header disagree"
    want = int(re.search(r"#define\s+SMK_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "smokehip.h")).read()).group(1))
    assert L.smk_abi_version() == want == _lib.ABI_VERSION == 17


def test_no_cpu_fallback():
    from smokephysai_amd.physics import FractalGenerator, NavierStokesSimulator, SmokeSimulator
    for cls in (NavierStokesSimulator, SmokeSimulator):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cls((64, 64), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FractalGenerator(device="cpu")


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under smokephysai_amd/ may import or reference it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "smokephysai_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "smoke_oracle" in txt:
                    bad.append(f)
    assert not bad, bad


def test_graft_entry_build_passes():
    """The driver's "does it build" hook (make is a no-op when the library is current; includes the ABI-version check)."""
    import __graft_entry__ as g
    g.build()
