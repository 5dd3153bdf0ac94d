"""Key material through the on-disk exchange format (tools/openfhe_export/bce_keyfile.h), SURVEY 8(f1).

The producer the format exists for is tools/openfhe_export/export_keys.cpp, which dumps the keys of an OpenFHE
BinFHEContext; OpenFHE is absent here, so the file is written from the CPU oracle's keys by an independent
Python writer of the same layout.  What is pinned: the loader (header checks, streaming import, device-side
transform) and that keys loaded from such a file evaluate gates bit for bit like the oracle that made them.
Parity against OpenFHE's own keys stays unpinned (no OpenFHE in this environment)."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def write_keyfile(path, o, with_z=True):
    p = o.params
    s, z, bsk, ksk = o.sk(), o.z(), o.bsk(), o.ksk()
    with open(path, "wb") as f:
        f.write(b"BCEKEYS1")
        f.write(struct.pack("<II", 1, p["method"]))
        f.write(struct.pack("<8Q", p["n"], p["N"], p["q"], p["Q"], p["qKS"], p["baseKS"], p["baseG"], p["baseR"]))
        f.write(struct.pack("<QQ", bsk.size, ksk.size))
        f.write(struct.pack("<II", 1 if with_z else 0, 0))
        f.write(s.astype("<i4").tobytes())
        if with_z:
            f.write(z.astype("<i4").tobytes())
        if f.tell() % 8:
            f.write(b"\0" * (8 - f.tell() % 8))
        f.write(bsk.astype("<u8").tobytes())
        f.write(ksk.astype("<u4").tobytes())


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_keys_from_file_evaluate_like_the_oracle(bce, orc, tmp_path, method):
    o = orc.Oracle(orc.TOY, getattr(orc, method))
    o.keygen(20240)
    path = str(tmp_path / "keys.bce")
    write_keyfile(path, o, with_z=(method == "GINX"))           # an OpenFHE export carries no ring secret
    c = bce.BinFHEContext(bce.TOY, getattr(bce, method))
    c.import_keys_file(path)
    ca, cb = o.encrypt(1, 0), o.encrypt(0, 1)
    c.pool_reserve(6)
    c.lwe_write([0, 1], np.stack([ca, cb]))
    c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3), (bce.AND, 0, 1, 4, 0, 1), (bce.OP_REFRESH, 0, 0, 5)])
    out = c.lwe_read([2, 3, 4, 5])
    assert np.array_equal(out[0], o.eval_bingate(orc.NAND, ca, cb))
    assert np.array_equal(out[1], o.eval_bingate(orc.OR, ca, cb))
    assert np.array_equal(out[2], o.eval_bingate(orc.AND, ca, o.eval_not(cb)))
    assert np.array_equal(out[3], o.bootstrap(ca))
    assert list(c.Decrypt([2, 3, 4, 5])) == [1, 1, 1, 1]
    # the engine writes the same format: export -> import into a fresh context -> identical keys and ciphertexts
    path2 = str(tmp_path / "keys2.bce")
    if method == "GINX":
        c.export_keys_file(path2)
        assert open(path2, "rb").read() == open(path, "rb").read()
        d = bce.BinFHEContext(bce.TOY, bce.GINX)
        d.import_keys_file(path2)
        d.pool_reserve(3)
        d.lwe_write([0, 1], np.stack([ca, cb]))
        d.EvalGates([(bce.NAND, 0, 1, 2)])
        assert np.array_equal(d.lwe_read([2])[0], out[0])


def test_bad_key_files_are_rejected_with_a_message(bce, orc, tmp_path):
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(3)
    good = str(tmp_path / "good.bce")
    write_keyfile(good, o)
    data = open(good, "rb").read()
    c = bce.BinFHEContext(bce.TOY, bce.GINX)

    def expect_error(blob, text):
        p = str(tmp_path / "bad.bce")
        open(p, "wb").write(blob)
        with pytest.raises(bce.BceError) as e:
            c.import_keys_file(p)
        assert text in str(e.value), str(e.value)

    expect_error(b"NOTAKEYF" + data[8:], "bad magic")
    expect_error(data[:8] + struct.pack("<I", 9) + data[12:], "version")
    expect_error(data[:len(data) // 2], "truncated")
    expect_error(data[:16] + struct.pack("<Q", 65) + data[24:], "parameter n")            # n of another parameter set
    bad_s = bytearray(data); bad_s[104:108] = struct.pack("<i", 2)
    expect_error(bytes(bad_s), "LWE secret")
    with pytest.raises(bce.BceError):
        c.import_keys_file(str(tmp_path / "does_not_exist.bce"))
    with pytest.raises(bce.BceError) as e:
        bce.BinFHEContext(bce.TOY, bce.AP).import_keys_file(good)                          # GINX keys into an AP context
    assert "method" in str(e.value)
    c.import_keys_file(good)                                                                # and the good file still loads
