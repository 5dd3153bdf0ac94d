"""The C-ABI library loads on a GPU-less host, exports every symbol the headers declare, and
fails loudly (status + message, no exception across the boundary, no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bce_[a-z0-9_]+)\s*\(", src)) - {"bce_allgather_fn"})


def test_library_builds_and_exports_every_declared_symbol(bce):
    bce.build()
    L = bce.lib()
    for header in ("bce_gpu.h", "bce_circuit.h"):
        names = _declared(header)
        assert len(names) > 10
        for n in names:
            assert hasattr(L, n), "%s declares %s but libbce_amd.so does not export it" % (header, n)
    assert set(bce.ENGINE_SYMBOLS) <= set(_declared("bce_gpu.h"))
    assert set(bce.CIRCUIT_SYMBOLS) <= set(_declared("bce_circuit.h"))


def test_no_gpu_no_fallback(bce):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bce.BceError) as e:
        bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    assert e.value.code == bce.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    # a plaintext-only circuit refuses encrypted mode instead of computing on the host
    c = bce.Circuit()
    c.ReadFile(os.path.join(ROOT, "tests", "golden", "circuits", "adder_2bit.out"))
    c.Reset()
    c.setEncrypted(True)
    with pytest.raises(bce.BceError):
        c.SetInput([[1, 0], [1, 1]])


def test_argument_errors_are_status_codes(bce):
    L = bce.lib()
    h = C.c_void_p()
    assert L.bce_ctx_create(99, bce.GINX, 0, C.byref(h)) == bce.ERR_ARG
    assert L.bce_ctx_create(bce.TOY, 7, 0, C.byref(h)) == bce.ERR_ARG
    assert b"method" in L.bce_last_error(None)
    assert L.bce_eval_gates(None, 0, None) == bce.ERR_ARG


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "openfhe-boolean-circuit-evaluator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "binfhe_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
