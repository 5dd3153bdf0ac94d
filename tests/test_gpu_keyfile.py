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


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_evaluation_form_keys_import_without_a_transform(bce, orc, tmp_path, method):
    """SURVEY 8(f1), the part testable without OpenFHE: the bootstrapping key as OpenFHE holds it after BTKeyGen --
    EVALUATION form, bit-reversed order of its Cooley-Tukey transform for the minimal primitive 2N-th root, which is the
    oracle's (and the engine's) own convention (oracle/binfhe_oracle.c: bo_ntt_forward) -- goes in through
    bce_import_keys_eval and through a key file with bsk_format = 1, and evaluates bit for bit like the oracle."""
    import os
    import stat
    o = orc.Oracle(orc.TOY, getattr(orc, method))
    o.keygen(777)
    N = o.params["N"]
    bsk_eval = np.concatenate([o.ntt_forward(p) for p in o.bsk().reshape(-1, N)])
    ca, cb = o.encrypt(1, 0), o.encrypt(1, 1)

    def check(c):
        c.pool_reserve(4)
        c.lwe_write([0, 1], np.stack([ca, cb]))
        c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3, 1, 0)])
        out = c.lwe_read([2, 3])
        assert np.array_equal(out[0], o.eval_bingate(orc.NAND, ca, cb))
        assert np.array_equal(out[1], o.eval_bingate(orc.OR, o.eval_not(ca), cb))

    c = bce.BinFHEContext(bce.TOY, getattr(bce, method))
    c.import_keys_eval(o.sk(), o.z(), bsk_eval, o.ksk())
    check(c)
    assert np.array_equal(c.export_bsk_eval(), bsk_eval)          # what went in comes out (folded layouts undone)
    assert np.array_equal(c.export_bsk(), o.bsk())                # and its coefficient form is the oracle's key
    # the same through a file whose header says "evaluation form"
    path = str(tmp_path / "eval.bce")
    write_keyfile(path, o, with_z=False)
    blob = bytearray(open(path, "rb").read())
    blob[100:104] = struct.pack("<I", 1)
    off = (104 + 4 * o.params["n"] + 7) & ~7
    blob[off:off + 8 * bsk_eval.size] = bsk_eval.astype("<u8").tobytes()
    open(path, "wb").write(bytes(blob))
    d = bce.BinFHEContext(bce.TOY, getattr(bce, method))
    d.import_keys_file(path)
    check(d)
    # a context without the ring secret exports a file that says so (has_z = 0), readable by its owner only
    out_path = str(tmp_path / "reexport.bce")
    d.export_keys_file(out_path)
    assert stat.S_IMODE(os.stat(out_path).st_mode) == 0o600
    hdr = open(out_path, "rb").read(104)
    assert struct.unpack("<II", hdr[96:104]) == (0, 0)
    e = bce.BinFHEContext(bce.TOY, getattr(bce, method))
    e.import_keys_file(out_path)
    check(e)
    bad = bytearray(blob); bad[100:104] = struct.pack("<I", 5)
    open(path, "wb").write(bytes(bad))
    with pytest.raises(bce.BceError):
        bce.BinFHEContext(bce.TOY, getattr(bce, method)).import_keys_file(path)
