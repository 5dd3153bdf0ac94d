"""GPU parity of the HIP engine against the CPU oracle, through the C ABI (include/bce_gpu.h).

Bar: bit-exact (all integer arithmetic; the only floating point is RoundqQ, restated with
the same IEEE double operations).  The oracle is a restatement of OpenFHE's algorithm, not
OpenFHE itself: ciphertext-level parity against OpenFHE is unpinned (see oracle header).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x0FE5EED


def _pair(bce, orc, ps_name, seed=SEED):
    ps_b = getattr(bce, ps_name)
    ps_o = getattr(orc, ps_name)
    o = orc.Oracle(ps_o, orc.GINX)
    o.keygen(seed)
    c = bce.BinFHEContext(ps_b, bce.GINX)
    c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
    return o, c


@pytest.fixture(scope="module")
def toy(bce, orc):
    return _pair(bce, orc, "TOY")


@pytest.fixture(scope="module")
def std128(bce, orc):
    return _pair(bce, orc, "STD128_OPT")


def test_params_match(toy, std128):
    for o, c in (toy, std128):
        assert o.params == c.params


@pytest.mark.parametrize("which", ["toy", "std128"])
def test_ntt_matches_oracle(which, request):
    o, c = request.getfixturevalue(which)
    rng = np.random.default_rng(1)
    polys = rng.integers(0, o.params["Q"], size=(5, o.N), dtype=np.uint64)
    polys[0] = 0
    polys[0, 1] = 1          # X
    polys[1] = o.params["Q"] - 1
    fwd = c.debug_ntt(polys, inverse=False)
    for k in range(polys.shape[0]):
        assert np.array_equal(fwd[k], o.ntt_forward(polys[k])), "forward NTT differs from oracle, poly %d" % k
    back = c.debug_ntt(fwd, inverse=True)
    assert np.array_equal(back, polys)


def _gate_cases(o, base=0):
    """all (gate, a, b) input combinations with fresh oracle encryptions"""
    cases, idx = [], base
    for gate in range(6):
        for a in (0, 1):
            for b in (0, 1):
                cases.append((gate, a, b, o.encrypt(a, idx), o.encrypt(b, idx + 1)))
                idx += 2
    return cases


def _truth(gate, a, b):
    return [a | b, a & b, 1 - (a | b), 1 - (a & b), a ^ b, 1 - (a ^ b)][gate]


def test_toy_all_gates_bit_exact_stages(toy, bce):
    o, c = toy
    cases = _gate_cases(o)
    nb = len(cases)
    c.pool_reserve(3 * nb)
    slots = np.arange(2 * nb, dtype=np.uint32)
    c.lwe_write(slots, np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    bits = c.Decrypt(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        prep = o.gate_prep(g, ca, cb)
        r_acc = o.blind_rotate(g, prep)
        assert np.array_equal(acc[i], r_acc), "accumulator differs, case %d" % i
        r_lweN = o.extract_modswitch(r_acc)
        assert np.array_equal(lweN[i], r_lweN), "extract/modswitch differs, case %d" % i
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(ks[i], r_ks), "keyswitch differs, case %d" % i
        r_out = o.modswitch_final(r_ks)
        assert np.array_equal(out[i], r_out), "final ciphertext differs, case %d" % i
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
        assert bits[i] == _truth(g, a, b) == o.decrypt(out[i])


def test_std128_gates_bit_exact(std128, bce):
    o, c = std128
    cases = [x for x in _gate_cases(o, base=1000) if x[0] in (bce.AND, bce.OR, bce.NAND)][:6]
    nb = len(cases)
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "accumulator differs, case %d" % i
        r_lweN = o.extract_modswitch(r_acc)
        assert np.array_equal(lweN[i], r_lweN)
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(ks[i], r_ks)
        assert np.array_equal(out[i], o.modswitch_final(r_ks))
        assert o.decrypt(out[i]) == _truth(g, a, b)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_std128_every_blind_rotation_kernel_bit_exact(std128, bce, variant, monkeypatch):
    """N = 1024 / dG = 4 has three blind-rotation kernels (1: one wave per inverse transform, plain key; split
    transform with the folded key at 2: one / 3: two workgroups per CU) chosen by launch size; BCE_VARIANT pins one at
    context creation.  Each must reproduce the oracle's accumulator and final ciphertext bit for bit."""
    o, c_auto = std128
    monkeypatch.setenv("BCE_VARIANT", str(variant))
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
    cases = [x for x in _gate_cases(o, base=3000 + 100 * variant) if x[0] in (bce.AND, bce.NOR, bce.XOR_FAST)][1:5]
    nb = len(cases)
    for ctx in (c, c_auto):
        ctx.pool_reserve(3 * nb)
        ctx.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    c_auto.EvalGates(bce.make_descs(descs))
    assert np.array_equal(out, c_auto.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32)))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "accumulator differs, variant %d case %d" % (variant, i)
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
        assert o.decrypt(out[i]) == _truth(g, a, b)


def test_launch_size_does_not_change_ciphertexts(toy, std128, bce):
    """Launch size selects the blind-rotation kernel (<= #CUs workgroups: one per CU) and how many workgroups share
    one bootstrap's key-switch rows (S = 16 ... 1): the same gate must give the same ciphertext alone and inside a
    large launch."""
    for (o, c), big in ((toy, 1500), (std128, 700)):
        ca, cb = o.encrypt(1, 7000), o.encrypt(1, 7001)
        want = o.eval_bingate(bce.NAND, ca, cb)
        c.pool_reserve(2 + big)
        c.lwe_write([0, 1], np.stack([ca, cb]))
        for nb in (1, 3, 40, 300, big):
            c.EvalGates(bce.make_descs([(bce.NAND, 0, 1, 2 + i) for i in range(nb)]))
            out = c.lwe_read(np.arange(2, 2 + nb, dtype=np.uint32))
            assert all(np.array_equal(out[i], want) for i in range(nb)), "launch of %d differs" % nb


def test_folded_not_refresh_and_unary(toy, bce):
    """neg0/neg1 folding == explicit EvalNOT; REFRESH == Bootstrap(); NOT/COPY ops."""
    o, c = toy
    ca, cb = o.encrypt(1, 5000), o.encrypt(0, 5001)
    c.pool_reserve(16)
    c.lwe_write([0, 1], np.stack([ca, cb]))
    c.EvalGates([(bce.AND, 0, 1, 2, 0, 1),      # a AND !b
                 (bce.AND, 0, 1, 3, 1, 0),      # !a AND b
                 (bce.OP_REFRESH, 0, 0, 4),
                 (bce.OP_NOT, 1, 1, 5),
                 (bce.OP_COPY, 0, 0, 6)])
    out = c.lwe_read([2, 3, 4, 5, 6])
    assert np.array_equal(out[0], o.eval_bingate(orc_and(bce), ca, o.eval_not(cb)))
    assert np.array_equal(out[1], o.eval_bingate(orc_and(bce), o.eval_not(ca), cb))
    assert np.array_equal(out[2], o.bootstrap(ca))
    assert np.array_equal(out[3], o.eval_not(cb))
    assert np.array_equal(out[4], ca)
    # XOR the reference's way (src/gate.cpp:198-202): OR of the two ANDs
    c.EvalGates([(bce.OR, 2, 3, 7)])
    x = c.lwe_read([7])[0]
    assert np.array_equal(x, o.eval_bingate(0, out[0], out[1]))
    assert o.decrypt(x) == 1


def test_edge_cases_of_the_frontier_call(toy, bce):
    """Empty frontier; the same ciphertext on both inputs (OpenFHE throws for the same OBJECT and the reference retries
    with a fresh encryption, src/gate.cpp:129-152 -- here handles are values, so it is simply evaluated); output
    written over an input of the same gate; unknown op."""
    o, c = toy
    ca = o.encrypt(1, 8000)
    c.pool_reserve(8)
    c.lwe_write([0], ca[None, :])
    c.EvalGates(bce.make_descs([]))                                 # nothing to do, no error
    c.EvalGates([(bce.AND, 0, 0, 1), (bce.NAND, 0, 0, 2)])          # in0 == in1
    out = c.lwe_read([1, 2])
    assert np.array_equal(out[0], o.eval_bingate(bce.AND, ca, ca)) and o.decrypt(out[0]) == 1
    assert np.array_equal(out[1], o.eval_bingate(bce.NAND, ca, ca)) and o.decrypt(out[1]) == 0
    c.lwe_write([3], out[1][None, :])
    c.EvalGates([(bce.OR, 0, 3, 3)])                                # out == in1: read in the prologue, written by the tail
    assert np.array_equal(c.lwe_read([3])[0], o.eval_bingate(bce.OR, ca, out[1]))
    with pytest.raises(bce.BceError) as e:
        c.EvalGates([(9, 0, 0, 4)])
    assert e.value.code == bce.ERR_ARG


def orc_and(bce):
    return bce.AND  # same numeric value in the oracle (BINGATE order)


def test_keygen_matches_oracle_keygen(bce, orc):
    """Independent host PRNG/sampler implementations + device NTT give identical keys."""
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(1234567)
    c = bce.BinFHEContext(bce.TOY, bce.GINX)
    c.KeyGen(1234567)
    s, z = c.export_sk()
    assert np.array_equal(s, o.sk()) and np.array_equal(z, o.z())
    assert np.array_equal(c.export_ksk(), o.ksk())
    assert np.array_equal(c.export_bsk(), o.bsk())
    # and the engine's own Encrypt uses the same stream as the oracle's
    c.pool_reserve(4)
    c.set_encrypt_seed(1234567)                    # deterministic test mode: stream index = enc_index_base + k
    c.Encrypt([1, 0, 1], [0, 1, 2], enc_index_base=77)
    got = c.lwe_read([0, 1, 2])
    for k, bit in enumerate([1, 0, 1]):
        assert np.array_equal(got[k], o.encrypt(bit, 77 + k))
    assert list(c.Decrypt([0, 1, 2])) == [1, 0, 1]


def test_strided_instances(toy, bce):
    o, c = toy
    K, stride = 3, 8
    c.pool_reserve(K * stride)
    ins = []
    for k in range(K):
        ca, cb = o.encrypt(k & 1, 9000 + 2 * k), o.encrypt(1, 9001 + 2 * k)
        ins.append((ca, cb))
        c.lwe_write([k * stride, k * stride + 1], np.stack([ca, cb]))
    c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3)], instances=K, slot_stride=stride)
    for k in range(K):
        out = c.lwe_read([k * stride + 2, k * stride + 3])
        assert np.array_equal(out[0], o.eval_bingate(bce.NAND, *ins[k]))
        assert np.array_equal(out[1], o.eval_bingate(bce.OR, *ins[k]))


def test_errors_are_reported_not_thrown(bce):
    c = bce.BinFHEContext(bce.TOY, bce.GINX)
    with pytest.raises(bce.BceError) as e:
        c.EvalGates([(bce.AND, 0, 1, 2)])
    assert e.value.code == bce.ERR_NO_KEYS
    c.KeyGen(1)
    c.pool_reserve(2)
    with pytest.raises(bce.BceError) as e:
        c.EvalGates([(bce.AND, 0, 1, 2)])
    assert e.value.code == bce.ERR_POOL
    with pytest.raises(bce.BceError) as e:           # N = 2048 with four gadget digits needs a ring modulus below 2^31 (STD256 has
        bce.BinFHEContext(method=bce.GINX, custom=(16, 2048, 1024, 68719403009, 1 << 14, 128, 1 << 9, 32))   # one); 36 bits: no kernel
    assert e.value.code == bce.ERR_UNSUPPORTED
    with pytest.raises(bce.BceError) as e:           # ring modulus of 40 bits or more (1099511630849 = 1 mod 1024, prime): no kernel
        bce.BinFHEContext(method=bce.GINX, custom=(16, 512, 512, 1099511630849, 1 << 14, 128, 1 << 14, 23))
    assert e.value.code == bce.ERR_UNSUPPORTED


# ---- AP (DM) method: SURVEY 8(a7) -------------------------------------------------------------
@pytest.fixture(scope="module")
def toy_ap(bce, orc):
    o = orc.Oracle(orc.TOY, orc.AP)
    o.keygen(SEED)
    c = bce.BinFHEContext(bce.TOY, bce.AP)
    c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
    return o, c


def test_ap_toy_all_gates_bit_exact_stages(toy_ap, bce):
    o, c = toy_ap
    assert o.params == c.params and o.params["method"] == 1
    cases = _gate_cases(o, base=300)
    nb = len(cases)
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "AP accumulator differs, case %d" % i
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
        assert o.decrypt(out[i]) == _truth(g, a, b)


def test_ap_keygen_matches_oracle(bce, orc):
    o = orc.Oracle(orc.TOY, orc.AP)
    o.keygen(4242)
    c = bce.BinFHEContext(bce.TOY, bce.AP)
    c.KeyGen(4242)
    assert np.array_equal(c.export_ksk(), o.ksk())
    assert np.array_equal(c.export_bsk(), o.bsk())


def test_ap_std128_gate_same_seed_keys(bce, orc):
    """STD128_AP (n=512, N=1024, baseG=2^9): keys are NOT transferred (1.6 GB); both sides derive
    them from the same seed (keygen parity is established above), then ciphertexts must agree."""
    o = orc.Oracle(orc.STD128_AP, orc.AP)
    o.keygen(99)
    c = bce.BinFHEContext(bce.STD128_AP, bce.AP)
    c.KeyGen(99)
    ca, cb = o.encrypt(1, 0), o.encrypt(1, 1)
    c.pool_reserve(4)
    c.lwe_write([0, 1], np.stack([ca, cb]))
    c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3)])
    out = c.lwe_read([2, 3])
    assert np.array_equal(out[0], o.eval_bingate(bce.NAND, ca, cb))
    assert np.array_equal(out[1], o.eval_bingate(bce.OR, ca, cb))
    assert list(c.Decrypt([2, 3])) == [0, 1]


@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_medium_28bit_modulus_non_lazy_path_bit_exact_stages(bce, orc, method):
    """MEDIUM (n = 422, N = 1024, 28-bit Q, base 2^10: 3 gadget digits): 22 Q does not fit 32 bits, so the kernels take
    the non-lazy Harvey forms (values in [0, 4Q), canonical digits, exact Barrett) -- a code path no other parameter
    set of the tests reaches.  Keys from the same seed on both sides."""
    o = orc.Oracle(orc.MEDIUM, getattr(orc, method))
    o.keygen(606)
    c = bce.BinFHEContext(bce.MEDIUM, getattr(bce, method))
    c.KeyGen(606)
    assert o.params == c.params and o.params["Q"] >= (1 << 27) and o.params["dG"] == 3
    cases = [x for x in _gate_cases(o, base=500) if x[0] in (bce.AND, bce.NOR, bce.XOR_FAST)][2:6]
    nb = len(cases)
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "accumulator differs, case %d" % i
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
        assert o.decrypt(out[i]) == _truth(g, a, b)
    o.close()
    c.close()


@pytest.fixture(scope="module")
def std128opt_ap_oracle(orc):
    o = orc.Oracle(orc.STD128_OPT, orc.AP)
    o.keygen(4242)
    yield o
    o.close()


@pytest.mark.parametrize("variant", [0, 1, 3])
def test_std128_opt_ap_every_kernel_same_seed_keys(bce, std128opt_ap_oracle, variant, monkeypatch):
    """`-s STD128_OPT -m AP` of the reference's command line (src/utils.cpp:167-185): N = 1024 with 4 gadget digits, so
    the AP method also runs on the split-transform kernel (automatic choice: one workgroup per CU for this small
    launch; 3: two per CU) besides the one-wave-per-transform kernel (1).  2.1 GB key from the same seed on both sides."""
    o = std128opt_ap_oracle
    monkeypatch.setenv("BCE_VARIANT", str(variant))
    c = bce.BinFHEContext(bce.STD128_OPT, bce.AP)
    c.KeyGen(4242)
    ca, cb = o.encrypt(1, 10), o.encrypt(0, 11)
    c.pool_reserve(5)
    c.lwe_write([0, 1], np.stack([ca, cb]))
    c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3), (bce.XOR_FAST, 0, 1, 4)])
    out = c.lwe_read([2, 3, 4])
    assert np.array_equal(out[0], o.eval_bingate(bce.NAND, ca, cb))
    assert np.array_equal(out[1], o.eval_bingate(bce.OR, ca, cb))
    assert np.array_equal(out[2], o.eval_bingate(bce.XOR_FAST, ca, cb))
    assert list(c.Decrypt([2, 3, 4])) == [1, 1, 1]
    c.close()


# ---- 64-bit ring modulus (STD192: Q ~ 2^37, N = 2048): kernels64.hip -------------------------------
def _custom64(orc, base_g_bits=13, N=512):
    L = orc.lib()
    Q = L.bo_previous_prime(L.bo_first_prime(37, 2 * N), 2 * N)    # 37-bit prime = 1 mod 2N
    #       n   N  q    Q  qKS      baseKS baseG             baseR
    return (16, N, 512, Q, 1 << 15, 32,    1 << base_g_bits, 23)   # base 2^13: 3 gadget digits, 2^10: 4


@pytest.mark.parametrize("dg,N", [(3, 512), (4, 512), (3, 1024), (3, 2048)])
@pytest.mark.parametrize("arith", ["fp64", "int64"])
@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_q64_custom_context_bit_exact_stages(bce, orc, method, arith, dg, N, monkeypatch):
    """The 64-bit-modulus path has two blind-rotation kernels: exact-integer doubles (default for Q < 2^39) and
    64-bit integer Shoup arithmetic (BCE_FP64=0); both must match the oracle bit for bit at every stage.
    N = 2048 is the ring of BASELINE config 5 (STD192): the doubles kernel runs there as its 512-thread SPLIT
    instantiation (inverse transforms on 8 waves, LDS twiddle mirror), which no smaller ring reaches."""
    monkeypatch.setenv("BCE_FP64", "1" if arith == "fp64" else "0")
    if N == 2048 and arith == "fp64":
        # N = 2048 doubles kernel: 8-wave build (GINX default) and 16-wave build (AP default); force the OTHER one for the
        # second half of the gate cases below so that all four (method, build) pairs are compared with the oracle
        monkeypatch.setenv("BCE_VARIANT", "0")
    params = _custom64(orc, 13 if dg == 3 else 10, N)
    o = orc.Oracle(method=getattr(orc, method), custom=params)
    o.keygen(31337)
    c = bce.BinFHEContext(method=getattr(bce, method), custom=params)
    assert o.params == c.params and o.params["Q"] > (1 << 36) and o.params["dG"] == dg
    # NTT
    rng = np.random.default_rng(3)
    polys = rng.integers(0, o.params["Q"], size=(3, o.N), dtype=np.uint64)
    fwd = c.debug_ntt(polys, inverse=False)
    for k in range(3):
        assert np.array_equal(fwd[k], o.ntt_forward(polys[k]))
    assert np.array_equal(c.debug_ntt(fwd, inverse=True), polys)
    # engine keygen == oracle keygen (u64 words), then gates with imported keys
    c.KeyGen(31337)
    assert np.array_equal(c.export_bsk(), o.bsk()) and np.array_equal(c.export_ksk(), o.ksk())
    c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
    cases = _gate_cases(o, base=40)
    nb = len(cases)
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    if N == 2048 and arith == "fp64":
        # the other build of the N = 2048 kernel (8 <-> 16 waves) on a context of its own: identical outputs
        monkeypatch.setenv("BCE_VARIANT", "2" if method == "AP" else "3")
        c2 = bce.BinFHEContext(method=getattr(bce, method), custom=params)
        c2.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
        c2.pool_reserve(3 * nb)
        c2.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
        acc2, lweN2, ks2 = c2.debug_eval_stages(descs)
        assert np.array_equal(acc2, acc) and np.array_equal(lweN2, lweN) and np.array_equal(ks2, ks)
        assert np.array_equal(c2.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32)), out)
        c2.close()
        # both builds keep the key with the lowest gadget digit folded in (4 forward transforms per step); the plain-key
        # instantiations (6 transforms; what an inexact gadget would fall back to) must give the same words
        assert c.forward_transforms_per_step() == 4
        monkeypatch.setenv("BCE_FOLD", "0")
        for var in ("2", "3"):
            monkeypatch.setenv("BCE_VARIANT", var)
            c3 = bce.BinFHEContext(method=getattr(bce, method), custom=params)
            assert c3.forward_transforms_per_step() == 6
            c3.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
            c3.pool_reserve(3 * nb)
            c3.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
            acc3, lweN3, ks3 = c3.debug_eval_stages(descs)
            assert np.array_equal(acc3, acc) and np.array_equal(lweN3, lweN) and np.array_equal(ks3, ks)
            c3.close()
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "64-bit accumulator differs, case %d" % i
        r_lweN = o.extract_modswitch(r_acc)
        assert np.array_equal(lweN[i], r_lweN)
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(ks[i], r_ks)
        assert np.array_equal(out[i], o.modswitch_final(r_ks))
        assert o.decrypt(out[i]) == _truth(g, a, b)


def _custom_narrow(orc, N):
    L = orc.lib()
    Q = L.bo_previous_prime(L.bo_first_prime(29, 2 * N), 2 * N)    # the ring modulus of STD256 when N = 2048
    #       n   N  q          Q  qKS      baseKS baseG   baseR
    return (24, N, min(2 * N, 2048), Q, 1 << 14, 128,  1 << 8, 46)    # base 2^8: four gadget digits; baseR 46 as in STD256


@pytest.mark.parametrize("N", [1024, 2048])
@pytest.mark.parametrize("method", ["GINX", "AP"])
def test_four_digit_29_bit_modulus_narrow_kernel_bit_exact_stages(bce, orc, method, N):
    """STD256's parameter class (29-bit Q >= 2^28, four gadget digits, N = 2048; and its N = 1024 sibling): the integer
    64-bit kernel with 32-bit digit rows (kernels64.hip NARROW -- ten 64-bit rows of N = 2048 would not fit the LDS).
    Small n so that both methods' keys are quick: NTT, keygen (every key word) and every stage of EvalBinGate against the
    oracle, both methods (AP with OpenFHE's base 46 for this set: the digit of a is taken by division)."""
    params = _custom_narrow(orc, N)
    o = orc.Oracle(method=getattr(orc, method), custom=params)
    o.keygen(2718)
    c = bce.BinFHEContext(method=getattr(bce, method), custom=params)
    assert o.params == c.params and (1 << 28) <= o.params["Q"] < (1 << 29) and o.params["dG"] == 4
    assert c.forward_transforms_per_step() == 8
    rng = np.random.default_rng(5)
    polys = rng.integers(0, o.params["Q"], size=(3, o.N), dtype=np.uint64)
    fwd = c.debug_ntt(polys, inverse=False)
    for k in range(3):
        assert np.array_equal(fwd[k], o.ntt_forward(polys[k]))
    assert np.array_equal(c.debug_ntt(fwd, inverse=True), polys)
    c.KeyGen(2718)
    assert np.array_equal(c.export_bsk(), o.bsk()) and np.array_equal(c.export_ksk(), o.ksk())
    cases = _gate_cases(o, base=70)
    nb = len(cases)
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "narrow kernel accumulator differs, case %d" % i
        r_lweN = o.extract_modswitch(r_acc)
        assert np.array_equal(lweN[i], r_lweN)
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(ks[i], r_ks)
        assert np.array_equal(out[i], o.modswitch_final(r_ks))
        assert o.decrypt(out[i]) == _truth(g, a, b)
    o.close()
    c.close()


@pytest.mark.parametrize("method", ["GINX"])
def test_std192_gate_same_seed_keys(bce, orc, method):
    """STD192 (n=1024, N=2048, Q=137438822401, qKS=2^19) with the GINX method; the AP method of BASELINE
    config 5 (12.9 GB key) has its own module, tests/test_gpu_config5.py.
    Keys are derived from the same seed on both sides (keygen parity is established above)."""
    o = orc.Oracle(orc.STD192, getattr(orc, method))
    o.keygen(2718)
    c = bce.BinFHEContext(bce.STD192, getattr(bce, method))
    c.KeyGen(2718)
    assert o.params == c.params
    ca, cb = o.encrypt(1, 0), o.encrypt(0, 1)
    c.pool_reserve(4)
    c.lwe_write([0, 1], np.stack([ca, cb]))
    c.EvalGates([(bce.AND, 0, 1, 2), (bce.OR, 0, 1, 3)])
    out = c.lwe_read([2, 3])
    assert np.array_equal(out[0], o.eval_bingate(bce.AND, ca, cb))
    assert np.array_equal(out[1], o.eval_bingate(bce.OR, ca, cb))
    assert list(c.Decrypt([2, 3])) == [0, 1]
    # 48 gates on distinct ciphertexts in one launch (every operation, folded EvalNOTs) against the oracle's batched evaluation
    rng = np.random.default_rng(192)
    nb = 48
    bits = rng.integers(0, 2, size=2 * nb)
    cts = np.stack([o.encrypt(int(bits[i]), 700 + i) for i in range(2 * nb)])
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), cts)
    descs = [(int(rng.integers(0, 6)), 2 * i, 2 * i + 1, 2 * nb + i, int(rng.integers(0, 2)), int(rng.integers(0, 2))) for i in range(nb)]
    c.EvalGates(bce.make_descs(descs))
    pool = np.zeros((3 * nb, o.params["n"] + 1), dtype=np.uint64)
    pool[:2 * nb] = cts
    o.eval_gates(pool, descs)
    assert np.array_equal(c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32)), pool[2 * nb:])
    assert c.forward_transforms_per_step() == 4
    o.close()
    c.close()


def test_two_host_threads_on_distinct_contexts(bce, orc):
    """SURVEY 8(b) threading row: eval_gates is called from one host thread per context; distinct contexts
    may be driven concurrently from different host threads (ctypes releases the GIL during the calls)."""
    import threading
    results = {}

    def worker(tag, seed):
        o = orc.Oracle(orc.TOY, orc.GINX)
        o.keygen(seed)
        c = bce.BinFHEContext(bce.TOY, bce.GINX)
        c.KeyGen(seed)
        c.pool_reserve(64)
        ok = True
        for rep in range(6):
            ca, cb = o.encrypt(rep & 1, 2 * rep), o.encrypt(1, 2 * rep + 1)
            c.lwe_write([0, 1], np.stack([ca, cb]))
            c.EvalGates([(bce.NAND, 0, 1, 2), (bce.OR, 0, 1, 3), (bce.AND, 0, 1, 4, 1, 0)])
            out = c.lwe_read([2, 3, 4])
            ok &= np.array_equal(out[0], o.eval_bingate(bce.NAND, ca, cb))
            ok &= np.array_equal(out[1], o.eval_bingate(bce.OR, ca, cb))
            ok &= np.array_equal(out[2], o.eval_bingate(bce.AND, o.eval_not(ca), cb))
        results[tag] = bool(ok)

    ts = [threading.Thread(target=worker, args=("t%d" % i, 1000 + i)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert results == {"t0": True, "t1": True, "t2": True}


def test_encryption_randomness_is_fresh_by_default_and_reproducible_on_request(bce, orc):
    """a and e of Encrypt must not be a function of public values.  By default the context
    draws its encryption seed from OS entropy and numbers the streams itself, so the caller's index cannot make
    (a, e) repeat -- also after import_keys, where no key seed exists; set_encrypt_seed() gives the
    deterministic streams the oracle reproduces."""
    o = orc.Oracle(orc.TOY, orc.GINX)
    o.keygen(5)
    ctxs = []
    for how in ("keygen", "import"):
        c = bce.BinFHEContext(bce.TOY, bce.GINX)
        if how == "keygen":
            c.KeyGen(5)
        else:
            c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
        c.pool_reserve(8)
        c.Encrypt([1, 1], [0, 1], enc_index_base=0)
        c.Encrypt([1, 1], [2, 3], enc_index_base=0)           # same caller index again
        got = c.lwe_read([0, 1, 2, 3])
        assert len({got[k].tobytes() for k in range(4)}) == 4, "an (a, e) pair was reused (%s)" % how
        assert not any(np.array_equal(got[k], o.encrypt(1, j)) for k in range(4) for j in range(4)), \
            "default encryption equals the public key-seed stream (%s)" % how
        assert list(c.Decrypt([0, 1, 2, 3])) == [1, 1, 1, 1]
        ctxs.append(got)
    assert not np.array_equal(ctxs[0], ctxs[1])                # two contexts, two entropy draws
    c.set_encrypt_seed(5)
    c.Encrypt([0, 1], [4, 5], enc_index_base=40)
    det = c.lwe_read([4, 5])
    assert np.array_equal(det[0], o.encrypt(0, 40)) and np.array_equal(det[1], o.encrypt(1, 41))
    c.set_encrypt_seed(None)                                   # back to entropy
    c.Encrypt([0], [6], enc_index_base=40)
    assert not np.array_equal(c.lwe_read([6])[0], o.encrypt(0, 40))
    k = bce.BinFHEContext(bce.TOY, bce.GINX)
    k.KeyGen()                                                 # no seed: keys from OS entropy
    s1, _ = k.export_sk()
    k.KeyGen()
    s2, _ = k.export_sk()
    assert not np.array_equal(s1, s2)


def test_std128_fused_tail_every_stage_equals_separate_tail_and_oracle(std128, bce):
    """Saturated launches of the split-transform kernel (more workgroups than CUs) run extract + ModSwitch + KeySwitch +
    ModSwitch in the kernel's epilogue (fused_tail) instead of the k_tail_gather / k_tail_finish kernels.  The same
    gates evaluated in a small launch (separate tail kernels) and inside a 300-gate launch (fused) must agree at every
    stage, and with the oracle."""
    o, c = std128
    base = [x for x in _gate_cases(o, base=9000) if x[0] in (bce.AND, bce.OR, bce.NAND, bce.NOR)][3:9]
    nb_small, nb_big = len(base), 300
    c.pool_reserve(2 * nb_small + nb_big)
    c.lwe_write(np.arange(2 * nb_small, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in base]))
    small = [(g, 2 * i, 2 * i + 1, 2 * nb_small + i) for i, (g, _, _, _, _) in enumerate(base)]
    t0 = c.timing()["fused_tail_launches"]
    acc_s, lweN_s, ks_s = c.debug_eval_stages(small)
    out_s = c.lwe_read(np.arange(2 * nb_small, 3 * nb_small, dtype=np.uint32))
    assert c.timing()["fused_tail_launches"] == t0                      # small launch: separate tail kernels
    big = [(base[i % nb_small][0], 2 * (i % nb_small), 2 * (i % nb_small) + 1, 2 * nb_small + i) for i in range(nb_big)]
    acc_b, lweN_b, ks_b = c.debug_eval_stages(big)
    out_b = c.lwe_read(np.arange(2 * nb_small, 2 * nb_small + nb_big, dtype=np.uint32))
    assert c.timing()["fused_tail_launches"] == t0 + 1                  # saturated launch: tail in the epilogue
    for i in range(nb_big):
        k = i % nb_small
        assert np.array_equal(acc_b[i], acc_s[k]), "accumulator, gate %d" % i
        assert np.array_equal(lweN_b[i], lweN_s[k]), "extract + ModSwitch, gate %d" % i
        assert np.array_equal(ks_b[i], ks_s[k]), "KeySwitch, gate %d" % i
        assert np.array_equal(out_b[i], out_s[k]), "final ciphertext, gate %d" % i
    for k, (g, a, b, ca, cb) in enumerate(base[:3]):
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        r_lweN = o.extract_modswitch(r_acc)
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(acc_s[k], r_acc) and np.array_equal(lweN_s[k], r_lweN) and np.array_equal(ks_s[k], r_ks)
        assert np.array_equal(out_s[k], o.modswitch_final(r_ks)) and o.decrypt(out_s[k]) == _truth(g, a, b)


@pytest.mark.parametrize("variant", [2, 3])
def test_std128_plain_key_split_kernels_and_folded_key_round_trip(std128, bce, variant, monkeypatch):
    """STD128_OPT contexts keep the bootstrapping key with the lowest gadget digit folded in (rows l >= 1 hold
    ek_l - B^l ek_0; 6 forward transforms per step instead of 8).  BCE_FOLD=0 keeps the plain key and the 8-transform
    instantiations of both split-transform builds: same accumulator, same ciphertext, both equal to the oracle's;
    and a folded context exports the canonical key it imported."""
    o, c_fold = std128
    assert c_fold.forward_transforms_per_step() == 6
    monkeypatch.setenv("BCE_FOLD", "0")
    monkeypatch.setenv("BCE_VARIANT", str(variant))
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    assert c.forward_transforms_per_step() == 8
    c.import_keys(o.sk(), o.z(), o.bsk(), o.ksk())
    cases = [x for x in _gate_cases(o, base=7000 + 100 * variant) if x[0] in (bce.OR, bce.NAND, bce.XNOR_FAST)][2:6]
    nb = len(cases)
    for ctx in (c, c_fold):
        ctx.pool_reserve(3 * nb)
        ctx.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack([ca, cb]) for (_, _, _, ca, cb) in cases]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    acc_f, lweN_f, ks_f = c_fold.debug_eval_stages(descs)
    assert np.array_equal(acc, acc_f) and np.array_equal(lweN, lweN_f) and np.array_equal(ks, ks_f)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    assert np.array_equal(out, c_fold.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32)))
    for i, (g, a, b, ca, cb) in enumerate(cases):
        assert np.array_equal(acc[i], o.blind_rotate(g, o.gate_prep(g, ca, cb))), "accumulator differs, case %d" % i
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
    if variant == 2:
        assert np.array_equal(c_fold.export_bsk(), o.bsk())   # un-folded on the way out
    c.close()


def test_toy_keeps_the_plain_key(toy):
    """TOY's gadget (27-bit Q, 3 digits base 2^9) is not exact (tests/test_oracle.py): no folding there."""
    assert toy[1].forward_transforms_per_step() == 6 == 2 * toy[0].params["dG"]


def test_std128_saturated_launch_of_distinct_gates_equals_oracle(std128, bce):
    """One saturated launch (two workgroups per CU, folded key, tail fused into the epilogue) of 768 gates on 1,536
    DISTINCT fresh ciphertexts -- every operation, folded EvalNOTs on either input, refreshes -- against the oracle's
    batched evaluation (OpenMP over gates) of the same descriptors: identical final ciphertexts.  The stage-level tests
    above replicate a handful of inputs; this one walks 768 different accumulator trajectories through the kernel."""
    o, c = std128
    rng = np.random.default_rng(20260)
    nb = 768
    bits = rng.integers(0, 2, size=2 * nb)
    cts = np.stack([o.encrypt(int(bits[i]), 50000 + i) for i in range(2 * nb)])
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), cts)
    ops = [bce.OR, bce.AND, bce.NOR, bce.NAND, bce.XOR_FAST, bce.XNOR_FAST, bce.OP_REFRESH]
    descs = []
    for i in range(nb):
        op = ops[int(rng.integers(0, len(ops)))]
        n0, n1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        descs.append((op, 2 * i, 2 * i + 1, 2 * nb + i, n0, n1 if op != bce.OP_REFRESH else 0))
    t0 = c.timing()["fused_tail_launches"]
    c.EvalGates(bce.make_descs(descs))
    got = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    assert c.timing()["fused_tail_launches"] == t0 + 1
    pool = np.zeros((3 * nb, o.params["n"] + 1), dtype=np.uint64)
    pool[:2 * nb] = cts
    o.eval_gates(pool, descs)
    assert np.array_equal(got, pool[2 * nb:])
    # and they decrypt to the gate functions
    for i in (0, 1, 2, nb - 1):
        op, _, _, _, n0, n1 = descs[i]
        a, b = int(bits[2 * i]) ^ n0, int(bits[2 * i + 1]) ^ n1
        want = a if op == bce.OP_REFRESH else _truth(op, a, b)
        assert o.decrypt(got[i]) == want


def test_multi_round_launches_with_and_without_the_xcd_start_gate(bce, orc, monkeypatch):
    """Launches of more than one round of workgroups start their workgroups in per-XCD cohorts (DevParams::xcd_gate; wave 0
    waits, bounded, for the 64 workgroups that share an XCD's slots).  A launch of 1,100 bootstraps -- two full rounds and
    a partial third, so that the last cohort of every XCD is short and leaves through its count or its time-out -- must leave
    the ciphertexts a gate-less context (BCE_XCD_GATE=0) leaves, and the oracle's on a sample."""
    o = orc.Oracle(orc.STD128_OPT, orc.GINX)
    o.keygen(77)
    nb = 1100
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, size=2 * nb).astype(np.uint8)
    descs = bce.make_descs([(int(rng.choice([bce.AND, bce.OR, bce.NAND, bce.NOR])), 2 * i, 2 * i + 1, 2 * nb + i, int(rng.integers(0, 2)), 0) for i in range(nb)])
    outs = []
    for gate in ("1", "0"):
        monkeypatch.setenv("BCE_XCD_GATE", gate)
        c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
        c.KeyGen(77)
        c.set_encrypt_seed(5)
        c.pool_reserve(3 * nb)
        c.Encrypt(bits, np.arange(2 * nb, dtype=np.uint32), enc_index_base=0)
        c.EvalGates(descs)
        c.EvalGates(descs)                           # a second launch re-arms the counters
        outs.append((c.lwe_read(np.arange(0, 2 * nb, dtype=np.uint32)), c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))))
        c.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ins, got = outs[0]
    for i in (0, 511, 512, 1023, 1024, nb - 1):      # first / last workgroup of every round
        d = descs[i]
        a = o.eval_not(ins[2 * i]) if d.neg0 else ins[2 * i]
        assert np.array_equal(got[i], o.eval_bingate(d.op, a, ins[2 * i + 1])), i
    o.close()


@pytest.mark.parametrize("ps,fwd", [("STD128", 6), ("STD192_OPT", 4), ("STD128_APOPT", 6), ("MEDIUM", 6), ("STD256", 8), ("STD256_OPT", 8)])
def test_other_parameter_sets_of_the_table_same_seed_keys(bce, orc, ps, fwd):
    """The remaining rows of OpenFHE's parameter table that the kernels cover, GINX: STD128 (n = 512) runs the folded
    split-transform kernel like STD128_OPT, STD192_OPT (n = 805, qKS = 2^15) the folded N = 2048 doubles kernel;
    STD128_APOPT (27-bit Q, 3 digits base 2^9: gadget not exact) and MEDIUM (28-bit Q: non-lazy path) stay on the
    one-wave-per-transform kernel with the plain key; STD256 / STD256_OPT (N = 2048, 29-bit Q, FOUR digits base 2^8, q = 2048)
    the integer 64-bit kernel with 32-bit digit rows.  Engine keygen from the seed the oracle uses (keygen parity is
    established above), 12 gates on distinct ciphertexts vs the oracle's batched evaluation."""
    o = orc.Oracle(getattr(orc, ps), orc.GINX)
    o.keygen(1618)
    c = bce.BinFHEContext(getattr(bce, ps), bce.GINX)
    c.KeyGen(1618)
    assert o.params == c.params
    assert c.forward_transforms_per_step() == fwd
    rng = np.random.default_rng(7)
    nb = 12
    bits = rng.integers(0, 2, size=2 * nb)
    cts = np.stack([o.encrypt(int(bits[i]), 300 + i) for i in range(2 * nb)])
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), cts)
    descs = [(i % 6, 2 * i, 2 * i + 1, 2 * nb + i, int(rng.integers(0, 2)), int(rng.integers(0, 2))) for i in range(nb)]
    c.EvalGates(bce.make_descs(descs))
    pool = np.zeros((3 * nb, o.params["n"] + 1), dtype=np.uint64)
    pool[:2 * nb] = cts
    o.eval_gates(pool, descs)
    got = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    assert np.array_equal(got, pool[2 * nb:])
    for i in range(nb):
        op, _, _, _, n0, n1 = descs[i]
        assert o.decrypt(got[i]) == _truth(op, int(bits[2 * i]) ^ n0, int(bits[2 * i + 1]) ^ n1)
    o.close()
    c.close()


def test_scattered_slots_read_and_decrypt_like_dense_ones(toy, bce):
    """bce_lwe_read / bce_decrypt_bits on slots one pool stride apart (the outputs of K lock-step instances, what
    Circuit::Clock decrypts) go through a device gather + one copy; dense requests through one plain copy.  Both must
    return the words the oracle wrote, in request order (unsorted, with a repeated slot)."""
    o, c = toy
    rng = np.random.default_rng(77)
    stride, K, W = 997, 40, 9
    c.pool_reserve(stride * K + 16)
    slots = np.array([k * stride + w for k in range(K) for w in range(W)], dtype=np.uint32)
    bits = rng.integers(0, 2, len(slots)).astype(np.uint8)
    cts = np.stack([o.encrypt(int(b), 5000 + i) for i, b in enumerate(bits)])
    c.lwe_write(slots, cts)
    order = rng.permutation(len(slots))
    req = np.concatenate([slots[order], slots[order][:3]])            # scattered, unsorted, three repeats
    got = c.lwe_read(req)
    assert got.shape == (len(req), c.n + 1)
    assert np.array_equal(got[:len(slots)], cts[order]) and np.array_equal(got[len(slots):], cts[order][:3])
    assert list(c.Decrypt(req)) == list(bits[order]) + list(bits[order][:3])
    # dense request over the same rows (one instance): same words
    dense = np.arange(3 * stride, 3 * stride + W, dtype=np.uint32)
    assert np.array_equal(c.lwe_read(dense), cts[3 * W:4 * W])
    # a slot outside the pool is refused on both paths
    with pytest.raises(bce.BceError):
        c.lwe_read(np.array([0, stride * K + 10_000_000], dtype=np.uint32))
