"""Inputs and golden outputs of the reference's workload harnesses (src/test_*.cpp), restated
as data + a few lines of arithmetic so the CPU and GPU tests share them."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CIRCUITS = os.path.join(GOLDEN, "circuits")
_RAND = json.load(open(os.path.join(GOLDEN, "glibc_rand_bits.json")))
AES_VECTORS = json.load(open(os.path.join(GOLDEN, "aes_vectors.json")))


def hex_bits(h):  # HexStr2UintVec, src/utils.cpp:49-71
    out = []
    for ch in reversed(h):
        v = int(ch, 16)
        out.extend([(v >> b) & 1 for b in range(4)])
    return out


def bin_bits(s):  # BinStr2UintVec, src/utils.cpp:73-89
    return [int(ch) for ch in reversed(s)]


def interleaved_inputs(test_ix, nbits):
    """srand(test_ix); in1[ix]=rand()%2; in2[ix]=rand()%2 (src/test_adder.cpp:180-190)"""
    r = _RAND[str(test_ix)]
    return [r[2 * i] for i in range(nbits)], [r[2 * i + 1] for i in range(nbits)]


def to_int(bits):
    return sum(b << i for i, b in enumerate(bits))


def adder_case(test_ix, nbits):
    a, b = interleaved_inputs(test_ix, nbits)
    s = to_int(a) + to_int(b)
    return [a, b], [(s >> i) & 1 for i in range(nbits + 1)]


def comparator_case(test_ix, fname):
    a, b = interleaved_inputs(test_ix, 32)
    if test_ix == 0:
        b = list(a)                                   # src/test_comparator.cpp:197-200
    ia, ib = to_int(a), to_int(b)
    if "unsigned" not in fname:                       # int32 arithmetic, :252-268
        ia = ia - (1 << 32) if ia >> 31 else ia
        ib = ib - (1 << 32) if ib >> 31 else ib
    out = (ib >= ia) if "lteq" in fname else (ib > ia)
    return [a, b], [int(out)]


def multiplier_case(test_ix):
    a, b = interleaved_inputs(test_ix, 32)
    c = to_int(a) * to_int(b)
    return [a, b], [(c >> i) & 1 for i in range(64)]


def parity_case(test_ix):
    """8 random bits + bit8 = 0 (src/test_parity.cpp:176-206)"""
    r = _RAND[str(test_ix)]
    bits = [r[i] for i in range(8)] + [0]
    odd = sum(bits) & 1
    return [bits], [1 - odd, odd]


def hash_vectors(test_file):
    """(inhex, outhex) pairs of md5-test.txt / sha-256-test.txt"""
    ins, outs = [], []
    for line in open(os.path.join(CIRCUITS, test_file)):
        line = line.strip()
        if line.startswith("in="):
            ins.append(line[3:])
        elif line.startswith("out="):
            outs.append(line[4:])
    return list(zip(ins, outs))


def md5_case(inhex, outhex):
    """bit-reversed input and output (src/test_md5.cpp:237-254)"""
    return [list(reversed(hex_bits(inhex)))], list(reversed(hex_bits(outhex)))


SHA256_IV = "6a09e667bb67ae853c6ef372a54ff53a510e527f9b05688c1f83d9ab5be0cd19"


def sha256_new_case(inhex, outhex):
    """new-format sha256.txt: wire i<512 = bit i (LSB first) of the block, wires 512..767 = IV LSB
    first, output bit i = bit i of the digest"""
    return [hex_bits(inhex), hex_bits(SHA256_IV)], hex_bits(outhex)


def aes_case(v):
    """not reversed (src/test_aes.cpp:262-267)"""
    return [hex_bits(v["inhex1"]), hex_bits(v["inhex2"])], bin_bits(v["outbin"])
