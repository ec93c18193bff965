"""C-ABI checks that need no GPU: the library loads, exports every symbol include/indm_hip.h declares, the
ctypes binding declares exactly those symbols, and the product path refuses to run without the HIP kernels."""
import os
import re

import pytest
import torch

from indm_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "indm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long|const char\*)\s+(indm_\w+)\s*\(", text)))


def test_header_matches_binding():
    assert header_symbols() == sorted(hip.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert hip.available(), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = hip.lib()
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.axpby(torch.zeros(8), None, 1.0, 0.0)
    with pytest.raises(RuntimeError):
        hip.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "indm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "indm_oracle" not in src and "import oracle" not in src and "_emu" not in src, fn


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's appendix (tools/abi_table.py) names every function the header declares"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    appendix = doc[doc.index("## Appendix A. Every exported entry point"):]
    missing = [s for s in header_symbols() if f"`{s}`" not in appendix]
    assert not missing, missing
