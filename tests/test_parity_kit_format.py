"""CPU half of the OpenFHE parity kit: the gate-vector file format (tools/openfhe_export/bce_keyfile.h, "BCEGVEC1")
round-trips through the reader / writer of compare.py, the C header's structs have the documented sizes, and the tool
fails loudly (exit code 2) when the engine has no GPU -- it never falls back to a CPU evaluation."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "openfhe_export"))
import compare as kit  # noqa: E402

PARAMS = dict(n=64, N=512, q=512, Q=134215681, qKS=134215681, baseKS=25, baseG=512, baseR=23)


def _records(rng):
    W = PARAMS["n"] + 1
    ct = lambda: rng.integers(0, PARAMS["q"], W, dtype=np.uint64)
    return [(kit.AND, 3, 1, [ct(), ct(), ct()]), (kit.K_NOT, 1, 0, [ct(), ct()]), (kit.K_ENC_DEFAULT, 1, 1, [ct()]),
            (kit.K_NTT, 0, 0, [rng.integers(0, PARAMS["Q"], 512, dtype=np.uint64)] * 2)]


def test_gate_vector_file_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    recs = _records(rng)
    path = str(tmp_path / "v.bgv")
    kit.write_gatevec(path, 2, PARAMS, recs)
    method, params, back = kit.read_gatevec(path)
    assert method == 2 and params == PARAMS and len(back) == len(recs)
    for (kind, bits, dec, parts), r in zip(recs, back):
        assert (r.kind, r.in_bits, r.decrypted) == (kind, bits, dec)
        assert np.array_equal(r.payload, np.concatenate(parts))
    data = open(path, "rb").read()
    open(path, "wb").write(data[:-8])
    try:
        kit.read_gatevec(path)
        assert False, "a truncated file must be refused"
    except ValueError as e:
        assert "truncated" in str(e)


def test_header_structs_have_the_documented_sizes(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "bce_keyfile.h"\nint main(void) { printf("%zu %zu %zu\\n", '
                   'sizeof(bce_keyfile_header), sizeof(bce_gatevec_header), sizeof(bce_gatevec_record)); return 0; }\n')
    exe = str(tmp_path / "sizes")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "tools", "openfhe_export"), "-o", exe, str(src)])
    assert subprocess.check_output([exe]).split() == [b"104", b"88", b"16"]


def test_compare_without_a_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        return                                           # covered by tests/test_gpu_parity_kit.py on a GPU box
    rng = np.random.default_rng(1)
    vec, key = str(tmp_path / "v.bgv"), str(tmp_path / "k.bce")
    kit.write_gatevec(vec, 2, PARAMS, _records(rng))
    import struct
    with open(key, "wb") as f:                           # header only: the context creation fails first
        f.write(b"BCEKEYS1" + struct.pack("<II", 1, 2) + struct.pack("<8Q", *[PARAMS[k] for k in kit.PARAM_FIELDS]))
        f.write(struct.pack("<QQII", 0, 0, 0, 0))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "openfhe_export", "compare.py"), key, vec],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "compare.py:" in r.stderr, r.stdout + r.stderr
