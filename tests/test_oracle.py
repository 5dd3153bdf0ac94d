"""The CPU oracle against what pins it: number-theory values, algebraic identities, gate truth
tables and the committed golden vectors.  Ciphertext-level parity with OpenFHE itself is
UNPINNED (OpenFHE is absent; the reference stores no ciphertext)."""
import json
import os

import numpy as np
import pytest

from kat import GOLDEN


def test_prime_search_and_roots(orc):
    L = orc.lib()
    # FirstPrime / minimal primitive 2N-th roots as computed in SURVEY.md App. D.1
    assert L.bo_first_prime(27, 2048) == 134246401 and L.bo_min_primitive_root(134246401, 2048) == 455622
    assert L.bo_first_prime(27, 1024) == 134224897 and L.bo_min_primitive_root(134224897, 1024) == 341565
    assert L.bo_first_prime(37, 4096) == 137439006721 and L.bo_min_primitive_root(137439006721, 4096) == 13167220
    # GenerateBinFHEContext uses PreviousPrime(FirstPrime(bits, 2N), 2N): the 27-bit NTT prime 2^27 - 2^11 + 1
    assert L.bo_previous_prime(134246401, 2048) == 134215681 == (1 << 27) - (1 << 11) + 1
    assert L.bo_previous_prime(134224897, 1024) == 134215681


@pytest.mark.parametrize("ps,exp", [
    ("TOY", dict(n=64, N=512, q=512, Q=134215681, qKS=134215681, baseKS=25, dKS=6, baseG=512, dG=3, baseR=23, dR=2)),
    ("STD128_OPT", dict(n=502, N=1024, q=1024, Q=134215681, qKS=16384, baseKS=128, dKS=2, baseG=128, dG=4, baseR=32, dR=2)),
    ("STD128", dict(n=512, N=1024, q=1024, Q=134215681, qKS=16384, baseKS=128, dKS=2, baseG=128, dG=4)),
    ("STD192", dict(n=1024, N=2048, q=1024, Q=137438822401, qKS=1 << 19, baseKS=28, dKS=4, baseG=1 << 13, dG=3, baseR=32, dR=2)),
])
def test_parameter_sets(orc, ps, exp):
    o = orc.Oracle(getattr(orc, ps), orc.GINX)
    for k, v in exp.items():
        assert o.params[k] == v, (ps, k)


def _negacyclic_schoolbook(a, b, Q):
    N = len(a)
    out = [0] * N
    for i in range(N):
        if a[i] == 0:
            continue
        for j in range(N):
            k = i + j
            v = a[i] * b[j]
            if k >= N:
                out[k - N] = (out[k - N] - v) % Q
            else:
                out[k] = (out[k] + v) % Q
    return out


def test_ntt_is_negacyclic_convolution(orc):
    o = orc.Oracle(orc.TOY, orc.GINX)
    Q, N = o.params["Q"], o.N
    rng = np.random.default_rng(7)
    a = rng.integers(0, Q, N, dtype=np.uint64)
    b = np.zeros(N, dtype=np.uint64)
    idx = rng.choice(N, 12, replace=False)
    b[idx] = rng.integers(0, Q, 12, dtype=np.uint64)
    fa, fb = o.ntt_forward(a), o.ntt_forward(b)
    prod = np.array([(int(x) * int(y)) % Q for x, y in zip(fa, fb)], dtype=np.uint64)
    got = o.ntt_inverse(prod)
    want = _negacyclic_schoolbook([int(x) for x in b], [int(x) for x in a], Q)
    assert [int(x) for x in got] == want
    assert np.array_equal(o.ntt_inverse(fa), a)


def _truth(gate, a, b):
    return [a | b, a & b, 1 - (a | b), 1 - (a & b), a ^ b, 1 - (a ^ b)][gate]


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_truth_tables_toy(orc, method):
    o = orc.Oracle(orc.TOY, getattr(orc, method))
    o.keygen(2024)
    q = o.params["q"]
    idx = 0
    worst = 0
    for gate in range(6):
        for a in (0, 1):
            for b in (0, 1):
                ca, cb = o.encrypt(a, idx), o.encrypt(b, idx + 1)
                idx += 2
                r = o.eval_bingate(gate, ca, cb)
                want = _truth(gate, a, b)
                assert o.decrypt(r) == want
                assert o.decrypt(o.eval_not(r)) == 1 - want
                worst = max(worst, abs(o.noise(r, want)))
                # staged entry points compose to EvalBinGate
                acc = o.blind_rotate(gate, o.gate_prep(gate, ca, cb))
                assert np.array_equal(o.modswitch_final(o.keyswitch(o.extract_modswitch(acc))), r)
    assert worst < q // 8, "bootstrapped noise must stay below the decryption threshold q/8"
    c1 = o.encrypt(1, 999)
    assert o.decrypt(o.bootstrap(c1)) == 1 and o.decrypt(o.bootstrap(o.eval_not(c1))) == 0


def test_xor_as_the_reference_builds_it(orc):
    """XOR = OR(AND(a, !b), AND(!a, b)), src/gate.cpp:198-202"""
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(5)
    for a in (0, 1):
        for b in (0, 1):
            ca, cb = o.encrypt(a, 10 + 2 * a + b), o.encrypt(b, 20 + 2 * a + b)
            t1 = o.eval_bingate(orc.AND, ca, o.eval_not(cb))
            t2 = o.eval_bingate(orc.AND, o.eval_not(ca), cb)
            assert o.decrypt(o.eval_bingate(orc.OR, t1, t2)) == a ^ b


def test_batched_eval_matches_single(orc):
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(11)
    W = o.n + 1
    pool = np.zeros((8, W), dtype=np.uint64)
    pool[0], pool[1] = o.encrypt(1, 0), o.encrypt(0, 1)
    descs = [(orc.AND, 0, 1, 2, 0, 1), (orc.AND, 0, 1, 3, 1, 0), (orc.OP_NOT, 0, 0, 4, 0, 0),
             (orc.OP_REFRESH, 1, 1, 5, 0, 0), (orc.OP_COPY, 0, 0, 6, 0, 0)]
    nb = o.eval_gates(pool, descs, nthreads=2)
    assert nb == 3
    assert np.array_equal(pool[2], o.eval_bingate(orc.AND, pool[0], o.eval_not(pool[1])))
    assert np.array_equal(pool[3], o.eval_bingate(orc.AND, o.eval_not(pool[0]), pool[1]))
    assert np.array_equal(pool[4], o.eval_not(pool[0]))
    assert np.array_equal(pool[5], o.bootstrap(pool[1]))
    assert np.array_equal(pool[6], pool[0])


def test_committed_golden_vectors(orc):
    g = json.load(open(os.path.join(GOLDEN, "oracle_golden_toy.json")))
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(g["seed"])
    assert o.params == g["params"]
    assert o.sk().tolist() == g["sk"]
    ca, cb = o.encrypt(1, 0), o.encrypt(0, 1)
    assert ca.tolist() == g["ct_a_bit1_idx0"] and cb.tolist() == g["ct_b_bit0_idx1"]
    for name, want in g["gates"].items():
        assert o.eval_bingate(getattr(orc, name), ca, cb).tolist() == want, name
    assert o.eval_not(ca).tolist() == g["not_a"]
    assert o.bootstrap(ca).tolist() == g["bootstrap_a"]


def test_custom_context_with_the_survey_modulus(orc):
    """SURVEY.md App. D.1 quotes Q = FirstPrime(27, 2N) = 134,246,401 (4 digits of 2^9 for TOY);
    the algorithm is modulus-agnostic, so that variant must work too."""
    o = orc.Oracle(method=orc.GINX, custom=(64, 512, 512, 134224897, 0, 25, 1 << 9, 23))
    assert o.params["dG"] == 4 and o.params["psi"] == 341565
    o.keygen(3)
    ca, cb = o.encrypt(1, 0), o.encrypt(1, 1)
    assert o.decrypt(o.eval_bingate(orc.AND, ca, cb)) == 1
    assert o.decrypt(o.eval_bingate(orc.NAND, ca, cb)) == 0


def test_std128_single_gate(orc):
    o = orc.Oracle(orc.STD128_OPT, orc.GINX)
    o.keygen(1)
    ca, cb = o.encrypt(1, 0), o.encrypt(0, 1)
    r = o.eval_bingate(orc.OR, ca, cb)
    assert o.decrypt(r) == 1 and abs(o.noise(r, 1)) < 128


@pytest.mark.parametrize("ps,exact", [("STD128_OPT", True), ("STD128", True), ("STD192", True), ("TOY", False)])
def test_signed_digit_decompose_is_exact_where_the_engine_folds_the_lowest_digit(orc, ps, exact):
    """The HIP kernels of the STD128* and STD192 sets never transform the lowest gadget digit: they use
    sum_l B^l dct_l = acc (mod Q) and read NTT(dct_0) off the evaluation-form accumulator.  That identity
    holds iff SignedDigitDecompose (rgsw-acc.cpp) is EXACT, i.e. the carry it drops after the last digit is zero for
    every centred residue.  Pinned here on the oracle's decomposition: exact for 27-bit Q with 4 digits base 2^7 and
    for 37-bit Q with 3 digits base 2^13; NOT exact for TOY (27-bit Q, 3 digits base 2^9: the top 0.1 % of the
    positive residues lose a carry), which is why the engine keeps the plain key there.  The closed-form predicate
    below is the one engine.cpp evaluates at context creation."""
    o = orc.Oracle(getattr(orc, ps), orc.GINX)
    Q, B, dG, N = o.params["Q"], o.params["baseG"], o.params["dG"], o.N
    span = sum(B ** l for l in range(dG))
    hi, lo = (B // 2 - 1) * span, (B // 2) * span      # largest / smallest (negated) value dG signed digits reach
    assert ((Q >> 1) <= hi + 1 and Q - (Q >> 1) <= lo) == exact
    rng = np.random.default_rng(11)
    edge = [0, 1, 2, Q - 1, Q - 2, (Q >> 1) - 1, Q >> 1, (Q >> 1) + 1, (Q >> 1) - 2, hi % Q, (hi + 1) % Q, (Q - lo) % Q]
    edge += [(B ** l * k) % Q for l in range(1, dG) for k in (1, B // 2 - 1, B // 2, B // 2 + 1, B - 1)]
    edge += [(Q - B ** l * k) % Q for l in range(1, dG) for k in (1, B // 2, B // 2 + 1)]
    top = (Q >> 1) - 1 - rng.integers(0, 1 << 12, size=256)                 # the residues just below Q/2
    vals = np.concatenate([np.array(edge, dtype=np.uint64), top.astype(np.uint64),
                           rng.integers(0, Q, size=2 * N - len(edge) - 256, dtype=np.uint64)])
    ct = vals.reshape(2, N)
    dct = o.signed_digit_decompose(ct)
    rec = np.zeros((2, N), dtype=object)
    for l in range(dG):
        for j in range(2):
            rec[j] = (rec[j] + dct[2 * l + j].astype(object) * (B ** l)) % Q
    bad = rec != ct.astype(object)
    if exact:
        assert not bad.any()
    else:
        assert bad.any()
        # only residues above the largest representable value lose their carry
        assert all(int(v) > hi and int(v) < (Q >> 1) for v in ct[bad])
    o.close()


def _negacyclic(a, b, N, Q):
    """a * b mod (X^N + 1, Q) from the definition (np.convolve, no transform).  a: small signed digits (|a| < 2^10),
    b: residues < 2^28, so every partial sum stays below 2^48 in int64."""
    full = np.convolve(a.astype(np.int64), b.astype(np.int64))
    lo, hi = full[:N].copy(), full[N:]
    lo[:hi.size] -= hi
    return lo % Q


def _rotate_minus_one(p, k, N, Q):
    """p(X) * (X^k - 1) mod (X^N + 1, Q), k in [0, 2N)"""
    k %= 2 * N
    sign = 1
    if k >= N:
        k -= N
        sign = -1
    r = np.empty_like(p)
    r[k:] = p[:N - k]
    r[:k] = -p[N - k:] if k else p[:0]
    return (sign * r - p) % Q


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_blind_rotation_equals_its_coefficient_domain_definition(orc, method):
    """An independent restatement of BootstrapGateCore + EvalAcc with NO number-theoretic transform anywhere: the
    accumulator stays in coefficient form, SignedDigitDecompose is re-implemented here in numpy, every RGSW product is a
    negacyclic convolution by definition (np.convolve) against the COEFFICIENT-form key the oracle exports, the GINX
    monomials are coefficient rotations.  It must reproduce the oracle's accumulator word for word (TOY, a few gates) --
    this pins the oracle's transform-domain machinery (CT/GS order, twiddles, evaluation-form keys and monomials, Barrett
    folds) to the ring arithmetic it stands for.  It does not pin the oracle to OpenFHE (still unpinned: see the header)."""
    o = orc.Oracle(orc.TOY, getattr(orc, method))
    o.keygen(424242)
    P = o.params
    n, N, q, Q, B, dG = P["n"], o.N, P["q"], P["Q"], P["baseG"], P["dG"]
    R, factor = 2 * dG, 2 * N // q
    words = 2 * N
    bsk = o.bsk().astype(np.int64)
    gate_const = {orc.OR: 5 * q // 8, orc.AND: 7 * q // 8, orc.NOR: q // 8, orc.NAND: 3 * q // 8}

    def decompose(acc):
        d = np.where(acc < (Q >> 1), acc, acc - Q).astype(np.int64)          # representative of rgsw-acc.cpp
        out = []
        for _ in range(dG):
            r = ((d + B // 2) % B) - B // 2                                    # signed remainder in [-B/2, B/2)
            d = (d - r) // B
            out.append(r)
        return out                                                             # out[l][c]

    for gate, bits in ((orc.AND, (1, 1)), (orc.NOR, (0, 1)), (orc.NAND, (1, 0))):
        ca, cb = o.encrypt(bits[0], 900 + gate), o.encrypt(bits[1], 950 + gate)
        prep = o.gate_prep(gate, ca, cb)
        b = int(prep[n])
        q1 = gate_const[gate]
        q2 = (q1 + q // 2) % q
        acc = np.zeros((2, N), dtype=np.int64)
        for j in range(q // 2):
            t = (b - j) % q
            inside = (q1 <= t < q2) if q1 < q2 else not (q2 <= t < q1)
            acc[1, j * factor] = (Q - (Q // 8 + 1)) if inside else (Q // 8 + 1)
        for i in range(n):
            aI = (q - int(prep[i])) % q
            if method == "GINX":
                steps = [((i * 2 + 0, i * 2 + 1), aI * factor)]
            else:
                steps, t = [], aI
                for k in range(P["dR"]):
                    a0 = t % P["baseR"]
                    t //= P["baseR"]
                    if a0:
                        steps.append((((i * P["baseR"] + a0) * P["dR"] + k,), None))
            for keys, rot in steps:
                dct = decompose(acc)
                prods = []
                for key in keys:
                    ek = bsk[key * R * words:(key + 1) * R * words].reshape(R, 2, N)
                    p = np.zeros((2, N), dtype=np.int64)
                    for l in range(dG):
                        for c in range(2):
                            for col in range(2):
                                p[col] = (p[col] + _negacyclic(dct[l][c], ek[2 * l + c, col], N, Q)) % Q
                    prods.append(p)
                if method == "GINX":
                    for col in range(2):
                        acc[col] = (acc[col] + _rotate_minus_one(prods[0][col], rot, N, Q)
                                    + _rotate_minus_one(prods[1][col], 2 * N - rot, N, Q)) % Q
                else:
                    acc = prods[0]
        assert np.array_equal(acc.astype(np.uint64), o.blind_rotate(gate, prep).reshape(2, N)), "gate %d" % gate
    o.close()


# ---------------------------------------------------------------------------------------------------------------------
# The oracle ALONE on the reference's own known answers.  tests/oracle_walk.py restates SetInput /
# Clock / Gate::Evaluate on oracle calls (no product code: its own netlist readers, the oracle's batched entry point);
# the inputs and goldens are those of the reference harnesses (tests/kat.py cites them).
# ---------------------------------------------------------------------------------------------------------------------
import kat
import oracle_walk as W


@pytest.fixture(scope="module")
def toy(orc):
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(0xC0FFEE)
    yield o
    o.close()


def test_oracle_alone_adder_2bit_all_sixteen_inputs(orc, toy):
    """examples/simple_ckts/adder_2bit/adder_2bit.out (hand-written, holds the only OR gate of the tree)."""
    nl = W.read_assembler_text(os.path.join(kat.CIRCUITS, "adder_2bit.out"))
    assert [g[0] for g in nl.gates].count("OR") == 1 and len(nl.loads) == 4 and len(nl.stores) == 3
    idx = 0
    for a in range(4):
        for b in range(4):
            out, boots = W.evaluate(orc, toy, nl, [[a & 1, a >> 1], [b & 1, b >> 1]], enc_index=idx)
            idx += 4
            assert kat.to_int(out) == a + b, (a, b, out)
            assert boots == 13 + 4                      # 13 gate bootstraps (+ one refresh per input bit)


def test_oracle_alone_parity_harness_vectors(orc, toy):
    """src/test_parity.cpp:176-206: srand(test_ix) inputs, outputs (even, odd)."""
    nl = W.read_assembler_text(os.path.join(kat.CIRCUITS, "parity.out"))
    for test_ix in range(4):
        ins, want = kat.parity_case(test_ix)
        out, _ = W.evaluate(orc, toy, nl, ins, enc_index=1000 + 16 * test_ix)
        assert out == want, test_ix


def test_oracle_alone_adder_32bit_harness_vectors(orc, toy):
    """src/test_adder.cpp:180-217 on examples/old_bristol_ckts/arith/adder_32bit.txt (310 bootstraps, 127 rounds)."""
    nl = W.read_bristol_old(os.path.join(kat.CIRCUITS, "adder_32bit.txt"))
    assert len(nl.gates) == 375
    for test_ix in (0, 1):
        ins, want = kat.adder_case(test_ix, 32)
        out, boots = W.evaluate(orc, toy, nl, ins, enc_index=2000 + 64 * test_ix)
        assert out == want, test_ix
        assert boots == 310 + 64


@pytest.mark.parametrize("fname", ["comparator_32bit_signed_lt.txt", "comparator_32bit_unsigned_lteq.txt"])
def test_oracle_alone_comparator_harness_vectors(orc, toy, fname):
    """src/test_comparator.cpp:184-269 (test_ix 0 compares a value with itself)."""
    nl = W.read_bristol_old(os.path.join(kat.CIRCUITS, fname))
    for test_ix in (0, 1, 2):
        ins, want = kat.comparator_case(test_ix, fname)
        out, _ = W.evaluate(orc, toy, nl, ins, enc_index=3000 + 64 * test_ix)
        assert out == want, (fname, test_ix)


def test_oracle_alone_aes_expanded_vector_0(orc, toy):
    """src/test_aes.cpp:186-228, first vector, on examples/old_bristol_ckts/crypto/AES-expanded.txt at TOY
    parameters: 27,692 gates, 66,415 gate bootstraps (XOR = NOT, NOT, AND, AND, OR) + 1,536 input refreshes, every one
    on the oracle; the 128 decrypted output bits must be the reference's golden string."""
    nl = W.read_bristol_old(os.path.join(kat.CIRCUITS, "AES-expanded.txt"))
    assert len(nl.gates) == 27692 and len(W.rounds(nl)) == 248
    ins, want = kat.aes_case(kat.AES_VECTORS[0])
    out, boots = W.evaluate(orc, toy, nl, ins, enc_index=10000)
    assert boots == 66415 + 1536
    assert out == want
