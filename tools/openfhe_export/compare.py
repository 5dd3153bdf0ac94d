#!/usr/bin/env python3
"""compare.py -- replays an OpenFHE-made gate-vector file on the MI355X engine, word for word.

    python tools/openfhe_export/compare.py keys.bce vectors.bgv [--device 0] [--verbose]

`keys.bce` and `vectors.bgv` come from tools/openfhe_export/export_keys.cpp run on a machine with OpenFHE
(`export_keys STD128_OPT GINX keys.bce --vectors vectors.bgv 64`; formats: bce_keyfile.h).  The keys are imported
through the C ABI (bce_import_keys_file), every recorded input ciphertext is written into the device pool
(bce_lwe_write), the recorded calls run as ONE batched frontier (bce_eval_gates -- the call that replaces
cc.EvalBinGate / cc.EvalNOT of /root/reference/src/gate.cpp:112,133,146,172,198-202), and every output word is compared
with what OpenFHE returned.  Exit code 0: every word of every record equal (and every Decrypt equal); 1: a mismatch
(the first one is printed with the engine's stage outputs, bce_debug_eval_stages); 2: unusable input.

The context is built from the parameters IN THE FILES (bce_ctx_create_custom), so a release of OpenFHE whose parameter
table differs from the engine's built-in one is still compared on its own terms -- the difference is reported.

This tool uses only the product (libbce_amd.so through its ctypes binding) and numpy; it never touches oracle/.
"""
import argparse
import importlib
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "openfhe-boolean-circuit-evaluator_amd"

KEY_MAGIC, VEC_MAGIC = b"BCEKEYS1", b"BCEGVEC1"
OR, AND, NOR, NAND, XOR_FAST, XNOR_FAST = range(6)
K_NOT, K_BOOTSTRAP, K_ENC_DEFAULT, K_ENC_FRESH, K_TAIL, K_NTT = 16, 17, 32, 33, 48, 64
GATE_NAMES = {0: "OR", 1: "AND", 2: "NOR", 3: "NAND", 4: "XOR_FAST", 5: "XNOR_FAST", 16: "EvalNOT", 17: "Bootstrap",
              32: "Encrypt(default)", 33: "Encrypt(FRESH)", 48: "tail", 64: "NTT"}
PARAM_FIELDS = ["n", "N", "q", "Q", "qKS", "baseKS", "baseG", "baseR"]


class Record:
    __slots__ = ("kind", "in_bits", "decrypted", "payload")

    def __init__(self, kind, in_bits, decrypted, payload):
        self.kind, self.in_bits, self.decrypted, self.payload = kind, in_bits, decrypted, payload


def write_gatevec(path, method, params, records):
    """records: iterable of (kind, in_bits, decrypted, [uint64 arrays])  -- the writer the tests use; export_keys.cpp
    is the OpenFHE-side one."""
    records = list(records)
    with open(path, "wb") as f:
        f.write(VEC_MAGIC)
        f.write(struct.pack("<II", 1, method))
        f.write(struct.pack("<8Q", *[params[k] for k in PARAM_FIELDS]))
        f.write(struct.pack("<Q", len(records)))
        for kind, in_bits, dec, parts in records:
            words = np.concatenate([np.ascontiguousarray(p, dtype="<u8").ravel() for p in parts])
            f.write(struct.pack("<IIII", kind, in_bits, dec, words.size))
            f.write(words.tobytes())


def read_header(path, magic):
    with open(path, "rb") as f:
        head = f.read(80)
    if len(head) < 80 or head[:8] != magic:
        raise ValueError("%s: not a %s file" % (path, magic.decode()))
    version, method = struct.unpack_from("<II", head, 8)
    params = dict(zip(PARAM_FIELDS, struct.unpack_from("<8Q", head, 16)))
    return version, method, params


def read_gatevec(path):
    version, method, params = read_header(path, VEC_MAGIC)
    if version != 1:
        raise ValueError("%s: unsupported version %d" % (path, version))
    data = open(path, "rb").read()
    (count,) = struct.unpack_from("<Q", data, 80)
    pos, records = 88, []
    for _ in range(count):
        if pos + 16 > len(data):
            raise ValueError("%s: truncated" % path)
        kind, in_bits, dec, words = struct.unpack_from("<IIII", data, pos)
        pos += 16
        if pos + 8 * words > len(data):
            raise ValueError("%s: truncated payload" % path)
        records.append(Record(kind, in_bits, dec, np.frombuffer(data, dtype="<u8", count=words, offset=pos).astype(np.uint64)))
        pos += 8 * words
    return method, params, records


def _first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return int(d[0]) if d.size else -1


def centred_noise(ct, s, q, bit):
    """b - <a, s> - bit * q/4, centred (what Decrypt rounds away)"""
    n = s.size
    v = (int(ct[n]) - int(np.dot(ct[:n].astype(np.int64), s.astype(np.int64))) - bit * (q // 4)) % q
    return v - q if v >= q // 2 else v


def compare(key_path, vec_path, device=0, verbose=False, out=sys.stdout):
    """returns (mismatching records, report lines)"""
    sys.path.insert(0, ROOT)
    bce = importlib.import_module(PKG)
    _, kmethod, kparams = read_header(key_path, KEY_MAGIC)
    method, params, records = read_gatevec(vec_path)
    if (method, params) != (kmethod, kparams):
        raise ValueError("key file and vector file were written for different contexts: %r vs %r" % ((kmethod, kparams), (method, params)))
    say = lambda *a: print(*a, file=out)
    cc = bce.BinFHEContext(method=method, device=device, custom=tuple(params[k] for k in PARAM_FIELDS))
    n, N, q, Q = cc.n, cc.N, params["q"], params["Q"]
    W = n + 1
    cc.import_keys_file(key_path)
    s, _ = cc.export_sk()
    say("context: method %s, n %d, N %d, q %d, Q %d, qKS %d, baseKS %d, baseG %d, baseR %d; psi %d; %d records"
        % ("AP" if method == 1 else "GINX", n, N, q, Q, params["qKS"], params["baseKS"], params["baseG"], params["baseR"],
           cc.params["psi"], len(records)))

    bad = 0
    cc.pool_reserve(max(8, 3 * sum(1 for r in records if r.kind <= K_BOOTSTRAP), len(records)))
    # ---- replayable records: one batched frontier ------------------------------------------------------------------
    gates = [r for r in records if r.kind <= XNOR_FAST or r.kind in (K_NOT, K_BOOTSTRAP)]
    if gates:
        slots, cts, descs = [], [], []
        for i, r in enumerate(gates):
            two = r.kind <= XNOR_FAST
            need = (3 if two else 2) * W
            if r.payload.size != need:
                raise ValueError("record of kind %d has %d payload words, expected %d" % (r.kind, r.payload.size, need))
            slots += [3 * i, 3 * i + 1]
            cts += [r.payload[:W], r.payload[W:2 * W] if two else r.payload[:W]]
            op = r.kind if two else (bce.OP_NOT if r.kind == K_NOT else bce.OP_REFRESH)
            descs.append((op, 3 * i, 3 * i + 1, 3 * i + 2))
        cc.lwe_write(slots, np.stack(cts))
        cc.EvalGates(descs)
        got = cc.lwe_read([3 * i + 2 for i in range(len(gates))])
        dec = cc.Decrypt([3 * i + 2 for i in range(len(gates))])
        first = None
        for i, r in enumerate(gates):
            want = r.payload[-W:]
            same = np.array_equal(got[i], want)
            if verbose or not same or int(dec[i]) != r.decrypted:
                say("  %-10s #%d: %s%s" % (GATE_NAMES[r.kind], i, "equal" if same else "DIFFERENT at word %d" % _first_diff(got[i], want),
                                            "" if int(dec[i]) == r.decrypted else "; Decrypt %d vs OpenFHE's %d" % (dec[i], r.decrypted)))
            if not same or int(dec[i]) != r.decrypted:
                bad += 1
                if first is None and r.kind != K_NOT:
                    first = i
        say("gate records: %d of %d identical to OpenFHE's output, word for word" % (len(gates) - sum(
            1 for i, r in enumerate(gates) if not np.array_equal(got[i], r.payload[-W:])), len(gates)))
        if first is not None:   # stage outputs of the first differing bootstrapped record, for a maintainer with OpenFHE's debug prints
            acc, lweN, ks = cc.debug_eval_stages([descs[first]])
            r = gates[first]
            say("first differing record (#%d, %s): engine stages" % (first, GATE_NAMES[r.kind]))
            say("  accumulator after blind rotation (coefficient form) acc[0][:4] = %s acc[1][:4] = %s" % (acc[0][:4].tolist(), acc[0][N:N + 4].tolist()))
            say("  after extract + ModSwitch(Q -> qKS): a[:4] = %s b = %d" % (lweN[0][:4].tolist(), int(lweN[0][N])))
            say("  after KeySwitch: a[:4] = %s b = %d" % (ks[0][:4].tolist(), int(ks[0][n])))
            say("  final: engine a[:4] = %s b = %d | OpenFHE a[:4] = %s b = %d"
                % (got[first][:4].tolist(), int(got[first][n]), r.payload[-W:][:4].tolist(), int(r.payload[-1])))
            say("  noise of the two results against OpenFHE's own Decrypt: engine %d, OpenFHE %d (q/8 = %d)"
                % (centred_noise(got[first], s, q, r.decrypted), centred_noise(r.payload[-W:], s, q, r.decrypted), q // 8))

    # ---- tail probes: ModSwitch / KeySwitch / ModSwitch on OpenFHE-made inputs --------------------------------------
    tails = [r for r in records if r.kind == K_TAIL]
    if tails:
        accs = []
        Q8 = Q // 8 + 1
        for r in tails:
            if r.payload.size != 2 * (N + 1) + 2 * W:
                raise ValueError("tail record has %d payload words" % r.payload.size)
            a, b = r.payload[:N].astype(object), int(r.payload[N])
            acc = np.zeros(2 * N, dtype=object)           # the accumulator whose transpose + extract gives (a, b)
            acc[0] = a[0]
            for i in range(1, N):
                acc[N - i] = (Q - a[i]) % Q
            acc[N] = (b - Q8) % Q
            accs.append(acc.astype(np.uint64))
        lweN, ks = cc.debug_tail(np.stack(accs), list(range(len(tails))))
        fin = cc.lwe_read(list(range(len(tails))))
        stage_bad = [0, 0, 0]
        for i, r in enumerate(tails):
            want = (r.payload[N + 1:2 * N + 2], r.payload[2 * N + 2:2 * N + 2 + W], r.payload[-W:])
            for k, (g, w) in enumerate(zip((lweN[i], ks[i], fin[i]), want)):
                if not np.array_equal(g, w):
                    stage_bad[k] += 1
                    say("  tail #%d: %s DIFFERENT at word %d" % (i, ("ModSwitch(Q -> qKS)", "KeySwitch", "ModSwitch(qKS -> q)")[k], _first_diff(g, w)))
                    break
        say("tail records: %d; differing in ModSwitch(Q->qKS) %d, KeySwitch %d, ModSwitch(qKS->q) %d" % (len(tails), *stage_bad))
        bad += sum(stage_bad)

    # ---- transform order ----------------------------------------------------------------------------------------------
    ntts = [r for r in records if r.kind == K_NTT]
    for i, r in enumerate(ntts):
        got_eval = cc.debug_ntt(r.payload[:N].reshape(1, N))[0]
        same = np.array_equal(got_eval, r.payload[N:2 * N])
        say("NTT record #%d: engine's evaluation form %s OpenFHE's" % (i, "equals" if same else "DIFFERS from"))
        bad += 0 if same else 1

    # ---- encryptions: plaintext and noise (cannot be replayed) ---------------------------------------------------------
    for kind in (K_ENC_FRESH, K_ENC_DEFAULT):
        encs = [r for r in records if r.kind == kind]
        if not encs:
            continue
        cc.lwe_write(list(range(len(encs))), np.stack([r.payload for r in encs]))
        dec = cc.Decrypt(list(range(len(encs))))
        wrong = sum(1 for d, r in zip(dec, encs) if int(d) != r.decrypted or r.decrypted != (r.in_bits & 1))
        noise = np.array([centred_noise(r.payload, s, q, r.in_bits & 1) for r in encs], dtype=np.float64)
        say("%s records: %d, wrong plaintexts %d, noise rms %.2f (fresh encryptions have sigma 3.19; a bootstrapped default is wider)"
            % (GATE_NAMES[kind], len(encs), wrong, float(np.sqrt(np.mean(noise ** 2)))))
        bad += wrong
    say("RESULT: %s" % ("every record identical" if bad == 0 else "%d record(s) differ" % bad))
    cc.close()
    return bad


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("keys")
    ap.add_argument("vectors")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args(argv)
    try:
        bad = compare(a.keys, a.vectors, a.device, a.verbose)
    except (ValueError, OSError, RuntimeError) as e:        # RuntimeError: bce.BceError (no GPU, rejected key file, ...)
        print("compare.py: %s" % e, file=sys.stderr)
        return 2
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
