"""The OpenFHE parity kit: tools/openfhe_export/export_keys.cpp records what OpenFHE returns for
EvalBinGate / EvalNOT / Bootstrap / Encrypt / Decrypt (+ tail and NTT probes) next to the keys of the same context, and
tools/openfhe_export/compare.py replays that file on the engine through the C ABI and compares every word.

OpenFHE is absent here, so the vector file of these tests is written FROM THE ORACLE in the same format (by the Python
writer inside compare.py): that pins the consumer side -- reader, replay, comparison, exit codes, localisation of a
difference -- and that an oracle-made file passes.  An OpenFHE-made file has not gone through it: parity against OpenFHE
itself stays UNPINNED until one does (INTEGRATION.md lists the two commands)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_keyfile import write_keyfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "openfhe_export"))
import compare as kit  # noqa: E402

pytestmark = pytest.mark.gpu


def oracle_vectors(orc, o, gates=12, seed=1):
    """The records export_keys.cpp writes, produced by the oracle: (kind, in_bits, decrypted, [arrays])."""
    rng = np.random.default_rng(seed)
    P = o.params
    n, N, q, Q = o.n, o.N, P["q"], P["Q"]
    recs, live, idx = [], [], 0
    for _ in range(4):
        bit = int(rng.integers(0, 2))
        fresh = o.encrypt(bit, idx)
        idx += 1
        recs.append((kit.K_ENC_FRESH, bit, o.decrypt(fresh), [fresh]))
        live.append((fresh, bit))
        dflt = o.bootstrap(o.encrypt(bit, idx))          # v1.0.x Encrypt(sk, bit): fresh + Bootstrap
        idx += 1
        recs.append((kit.K_ENC_DEFAULT, bit, o.decrypt(dflt), [dflt]))
        live.append((dflt, bit))
    for g in range(gates):
        gate = g % 6
        i, j = (int(v) for v in rng.choice(len(live), 2, replace=False))
        out = o.eval_bingate(gate, live[i][0], live[j][0])
        dec = o.decrypt(out)
        recs.append((gate, live[i][1] | (live[j][1] << 1), dec, [live[i][0], live[j][0], out]))
        live.append((out, dec & 1))
    for _ in range(3):
        ct, bit = live[int(rng.integers(0, len(live)))]
        nt, bt = o.eval_not(ct), o.bootstrap(ct)
        recs.append((kit.K_NOT, bit, o.decrypt(nt), [ct, nt]))
        recs.append((kit.K_BOOTSTRAP, bit, o.decrypt(bt), [ct, bt]))
    Q8 = Q // 8 + 1
    for _ in range(2):                                    # ModSwitch(qKS, ctQ), KeySwitch, ModSwitch(q, .)
        a = rng.integers(0, Q, N, dtype=np.uint64)
        b = int(rng.integers(0, Q))
        acc = np.zeros(2 * N, dtype=np.uint64)
        acc[0] = a[0]
        acc[N - np.arange(1, N)] = (Q - a[1:]) % Q
        acc[N] = (b - Q8) % Q
        lweN = o.extract_modswitch(acc)
        ks = o.keyswitch(lweN)
        recs.append((kit.K_TAIL, 0, 0, [np.concatenate([a, [b]]).astype(np.uint64), lweN, ks, o.modswitch_final(ks)]))
    coef = rng.integers(0, Q, N, dtype=np.uint64)
    recs.append((kit.K_NTT, 0, 0, [coef, o.ntt_forward(coef)]))
    return recs


def file_params(o):
    return {k: o.params[k] for k in kit.PARAM_FIELDS}


class Sink:
    def __init__(self):
        self.lines = []

    def write(self, s):
        self.lines.append(s)

    def flush(self):
        pass

    def text(self):
        return "".join(self.lines)


@pytest.mark.parametrize("ps,method", [("TOY", "GINX"), ("TOY", "AP"), ("STD128_OPT", "GINX")])
def test_oracle_made_vector_file_replays_word_for_word(bce, orc, tmp_path, ps, method):
    o = orc.Oracle(getattr(orc, ps), getattr(orc, method))
    o.keygen(777)
    keys, vecs = str(tmp_path / "keys.bce"), str(tmp_path / "vectors.bgv")
    write_keyfile(keys, o, with_z=False)                 # an OpenFHE export carries no ring secret
    recs = oracle_vectors(orc, o, gates=12 if ps == "TOY" else 6)
    kit.write_gatevec(vecs, o.params["method"], file_params(o), recs)
    m, p, back = kit.read_gatevec(vecs)
    assert m == o.params["method"] and p == file_params(o) and len(back) == len(recs)
    sink = Sink()
    assert kit.compare(keys, vecs, out=sink) == 0, sink.text()
    assert "RESULT: every record identical" in sink.text()
    o.close()


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_vector_file_of_a_37_bit_modulus_context(bce, orc, tmp_path, method):
    """The kit on the 64-bit-modulus path (STD192's ring: N = 2048, 37-bit Q, three digits base 2^13; small n so that the
    oracle writes the file in seconds): compare.py builds the context from the parameters in the files, the gates run on the
    folded fp64 kernels (N^-1 in the key, 8-byte twiddles), the tail probes on the 64-bit tail kernels, the NTT probe on
    the 64-bit transform."""
    L = orc.lib()
    N = 2048
    Q = L.bo_previous_prime(L.bo_first_prime(37, 2 * N), 2 * N)
    params = (16, N, 512, Q, 1 << 15, 32, 1 << 13, 23)
    o = orc.Oracle(method=getattr(orc, method), custom=params)
    o.keygen(4242)
    keys, vecs = str(tmp_path / "keys.bce"), str(tmp_path / "vectors.bgv")
    write_keyfile(keys, o, with_z=False)
    kit.write_gatevec(vecs, o.params["method"], file_params(o), oracle_vectors(orc, o, gates=6, seed=9))
    sink = Sink()
    assert kit.compare(keys, vecs, out=sink) == 0, sink.text()
    assert "RESULT: every record identical" in sink.text() and "Q %d" % Q in sink.text()
    o.close()


def test_a_single_flipped_word_fails_and_is_localised(bce, orc, tmp_path):
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(778)
    keys = str(tmp_path / "keys.bce")
    write_keyfile(keys, o, with_z=False)
    recs = oracle_vectors(orc, o, gates=6)
    W = o.n + 1

    def run(mutate):
        mine = [(k, b, d, [np.array(p, copy=True) for p in parts]) for k, b, d, parts in recs]
        mutate(mine)
        path = str(tmp_path / "bad.bgv")
        kit.write_gatevec(path, o.params["method"], file_params(o), mine)
        sink = Sink()
        return kit.compare(keys, path, out=sink), sink.text()

    first_gate = next(i for i, r in enumerate(recs) if r[0] <= kit.XNOR_FAST)

    def flip_gate_output(m):
        m[first_gate][3][2][5] ^= 1
    bad, text = run(flip_gate_output)
    assert bad == 1 and "DIFFERENT at word 5" in text and "engine stages" in text and "tail records: 2; differing in ModSwitch(Q->qKS) 0, KeySwitch 0" in text

    def flip_keyswitch(m):
        t = next(r for r in m if r[0] == kit.K_TAIL)
        t[3][2][W - 1] = (int(t[3][2][W - 1]) + 1) % o.params["qKS"]
    bad, text = run(flip_keyswitch)
    assert bad == 1 and "KeySwitch DIFFERENT" in text and "gate records: 12 of 12 identical" in text

    def flip_ntt(m):
        t = next(r for r in m if r[0] == kit.K_NTT)
        t[3][1][0] ^= 1
    bad, text = run(flip_ntt)
    assert bad == 1 and "DIFFERS from" in text

    def wrong_decrypt(m):
        k, b, d, parts = m[first_gate]
        m[first_gate] = (k, b, d ^ 1, parts)
    bad, text = run(wrong_decrypt)
    assert bad == 1 and "Decrypt" in text
    o.close()


def test_the_command_line_and_its_exit_codes(bce, orc, tmp_path):
    """`python tools/openfhe_export/compare.py keys vectors`: 0 equal, 1 different, 2 unusable input."""
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(779)
    keys, vecs, bad = str(tmp_path / "k.bce"), str(tmp_path / "v.bgv"), str(tmp_path / "b.bgv")
    write_keyfile(keys, o, with_z=False)
    recs = oracle_vectors(orc, o, gates=4)
    kit.write_gatevec(vecs, o.params["method"], file_params(o), recs)
    recs[-2][3][3][0] ^= 1                                # final word of a tail record
    kit.write_gatevec(bad, o.params["method"], file_params(o), recs)
    tool = os.path.join(ROOT, "tools", "openfhe_export", "compare.py")
    r = subprocess.run([sys.executable, tool, keys, vecs], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "every record identical" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, keys, bad], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "ModSwitch(qKS -> q) DIFFERENT" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, vecs, keys], capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "not a BCEKEYS1 file" in r.stderr, r.stdout + r.stderr
    o.close()


def test_evaluation_form_key_file_and_vectors(bce, orc, tmp_path):
    """bsk_format = 1 (the key as OpenFHE holds it after BTKeyGen) + the NTT record that pins its order."""
    import struct
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(780)
    keys, vecs = str(tmp_path / "ke.bce"), str(tmp_path / "v.bgv")
    p = o.params
    bsk = o.bsk().reshape(-1, o.N)
    bsk_eval = np.stack([o.ntt_forward(row) for row in bsk]).ravel()
    ksk, s = o.ksk(), o.sk()
    with open(keys, "wb") as f:
        f.write(b"BCEKEYS1" + struct.pack("<II", 1, p["method"]))
        f.write(struct.pack("<8Q", *[p[k] for k in kit.PARAM_FIELDS]))
        f.write(struct.pack("<QQII", bsk_eval.size, ksk.size, 0, 1))
        f.write(s.astype("<i4").tobytes())
        if f.tell() % 8:
            f.write(b"\0" * (8 - f.tell() % 8))
        f.write(bsk_eval.astype("<u8").tobytes())
        f.write(ksk.astype("<u4").tobytes())
    kit.write_gatevec(vecs, p["method"], file_params(o), oracle_vectors(orc, o, gates=6))
    sink = Sink()
    assert kit.compare(keys, vecs, out=sink) == 0, sink.text()
    # the transform-check trailer (what export_keys.cpp appends in evaluation mode): matching pairs are accepted, a producer
    # whose evaluation order differs is refused at import with a message that names the way out
    body = open(keys, "rb").read()
    rng = np.random.default_rng(2)
    coef = rng.integers(0, p["Q"], o.N, dtype=np.uint64)
    good = body + b"BCENTTCK" + struct.pack("<II", 1, 0) + coef.astype("<u8").tobytes() + o.ntt_forward(coef).astype("<u8").tobytes()
    open(keys, "wb").write(good)
    c = bce.BinFHEContext(bce.TOY, bce.GINX)
    c.import_keys_file(keys)
    wrong = np.roll(o.ntt_forward(coef), 1)                       # "another order"
    open(keys, "wb").write(body + b"BCENTTCK" + struct.pack("<II", 1, 0) + coef.astype("<u8").tobytes() + wrong.astype("<u8").tobytes())
    with pytest.raises(bce.BceError) as e:
        c.import_keys_file(keys)
    assert "EVALUATION representation is not this engine's" in str(e.value)
    c.close()
    o.close()
