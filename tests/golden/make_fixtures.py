"""Regenerates the small data fixtures under tests/golden/ (run in the build container).

* glibc_rand_bits.json : `srand(seed); rand() % 2` draws the reference harnesses use for their
  inputs (src/test_adder.cpp:180-190, src/test_comparator.cpp:184-201, src/test_multiplier.cpp:183-193,
  src/test_parity.cpp:176-189) so the tests do not depend on libc.
* aes_vectors.json     : the input / expected-output strings of src/test_aes.cpp:186-228 (data only).
* oracle_golden_toy.json : ciphertext-level vectors produced by the CPU oracle (oracle/) for TOY / GINX,
  seed 0x0FE5EED: the committed anchor the HIP path is compared with besides the live oracle.
"""
import ctypes
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def glibc_bits():
    libc = ctypes.CDLL("libc.so.6")
    out = {}
    for seed in range(10):
        libc.srand(seed)
        out[str(seed)] = [libc.rand() % 2 for _ in range(256)]
    return out


def aes_vectors(ref="/root/reference/src/test_aes.cpp"):
    src = open(ref).read()
    body = src[src.index("switch (loop_ix)"):src.index("default:", src.index("switch (loop_ix)"))]

    def strings(block, var):
        vals = []
        for m in re.finditer(var + r"\s*=\s*((?:\"[0-9a-f]*\"\s*)+);", block):
            vals.append("".join(re.findall(r"\"([0-9a-f]*)\"", m.group(1))))
        return vals

    cases = re.split(r"case \d+:", body)[1:]
    out = []
    for ci, blk in enumerate(cases):
        exp_blk, non_blk = blk.split("} else {")
        for name, b in (("AES-expanded", exp_blk), ("AES-non-expanded", non_blk)):
            out.append({"circuit": name, "case": ci, "inhex1": strings(b, "inhex1")[-1],
                        "inhex2": strings(b, "inhex2")[-1], "outbin": strings(b, "outbin")[-1]})
    return out


def oracle_golden():
    from oracle import oracle as O
    o = O.Oracle(O.TOY, O.GINX)
    o.keygen(0x0FE5EED)
    ca, cb = o.encrypt(1, 0), o.encrypt(0, 1)
    g = {"paramset": "TOY", "method": "GINX", "seed": 0x0FE5EED, "params": o.params,
         "sk": o.sk().tolist(), "ct_a_bit1_idx0": ca.tolist(), "ct_b_bit0_idx1": cb.tolist(), "gates": {}}
    for name, gate in (("OR", O.OR), ("AND", O.AND), ("NOR", O.NOR), ("NAND", O.NAND), ("XOR_FAST", O.XOR_FAST),
                       ("XNOR_FAST", O.XNOR_FAST)):
        g["gates"][name] = o.eval_bingate(gate, ca, cb).tolist()
    g["not_a"] = o.eval_not(ca).tolist()
    g["bootstrap_a"] = o.bootstrap(ca).tolist()
    acc = o.blind_rotate(O.AND, o.gate_prep(O.AND, ca, cb))
    g["and_acc_first8"] = acc[:8].tolist()
    g["and_acc_checksum"] = int(acc.sum() % (1 << 61))
    return g


if __name__ == "__main__":
    json.dump(glibc_bits(), open(os.path.join(HERE, "glibc_rand_bits.json"), "w"))
    json.dump(aes_vectors(), open(os.path.join(HERE, "aes_vectors.json"), "w"), indent=1)
    json.dump(oracle_golden(), open(os.path.join(HERE, "oracle_golden_toy.json"), "w"))
    print("fixtures written")
