"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps oracle/_build/libbinfhe_oracle.so (built by oracle/Makefile), the plain-C
restatement of the OpenFHE binfhe path the reference calls at src/gate.cpp:112,133,172,
198-202 and src/circuit.cpp:88-91,506,800.  Pinned to the reference's functional known answers
(tests/test_oracle.py::test_oracle_alone_*); PARITY UNPINNED at ciphertext level against OpenFHE
(see binfhe_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libbinfhe_oracle.so")

TOY, MEDIUM, STD128_AP, STD128_APOPT, STD128, STD128_OPT, STD192, STD192_OPT, STD256, STD256_OPT = range(10)
AP, GINX = 1, 2
OR, AND, NOR, NAND, XOR_FAST, XNOR_FAST = range(6)
OP_NOT, OP_REFRESH, OP_COPY = 16, 17, 18
P_NAMES = ["n", "N", "q", "Q", "qKS", "baseKS", "dKS", "baseG", "dG", "baseR", "dR", "method", "psi"]


class GateDesc(C.Structure):
    _fields_ = [("op", C.c_uint32), ("in0", C.c_uint32), ("in1", C.c_uint32),
                ("out", C.c_uint32), ("neg0", C.c_uint32), ("neg1", C.c_uint32)]


def build():
    """Compile the oracle if the shared object is missing or stale."""
    src = os.path.join(_HERE, "binfhe_oracle.c")
    if (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u64p = C.POINTER(C.c_uint64)
        L.bo_ctx_create.restype = C.c_void_p
        L.bo_ctx_create.argtypes = [C.c_int, C.c_int]
        L.bo_ctx_create_custom.restype = C.c_void_p
        L.bo_ctx_create_custom.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                           C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.bo_ctx_destroy.argtypes = [C.c_void_p]
        L.bo_get_params.argtypes = [C.c_void_p, u64p]
        L.bo_first_prime.restype = C.c_uint64
        L.bo_first_prime.argtypes = [C.c_uint32, C.c_uint64]
        L.bo_previous_prime.restype = C.c_uint64
        L.bo_previous_prime.argtypes = [C.c_uint64, C.c_uint64]
        L.bo_min_primitive_root.restype = C.c_uint64
        L.bo_min_primitive_root.argtypes = [C.c_uint64, C.c_uint64]
        L.bo_keygen.argtypes = [C.c_void_p, C.c_char_p]
        L.bo_export_sk.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_export_z.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_bsk_words.restype = C.c_uint64
        L.bo_bsk_words.argtypes = [C.c_void_p]
        L.bo_export_bsk.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_ksk_words.restype = C.c_uint64
        L.bo_ksk_words.argtypes = [C.c_void_p]
        L.bo_export_ksk.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_encrypt.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
        L.bo_decrypt.restype = C.c_int
        L.bo_decrypt.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_noise.restype = C.c_int64
        L.bo_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.bo_eval_not.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_eval_bingate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_bootstrap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_gate_prep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_blind_rotate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.bo_extract_modswitch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_keyswitch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_modswitch_final.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_ntt_forward.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_ntt_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.bo_signed_digit_decompose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bo_eval_gates.restype = C.c_uint64
        L.bo_eval_gates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def seed_bytes(seed):
    """32-byte key from an int (little-endian) or bytes."""
    if isinstance(seed, (bytes, bytearray)):
        return bytes(seed).ljust(32, b"\0")[:32]
    return int(seed).to_bytes(32, "little")


class Oracle:
    """One BinFHEContext-equivalent (parameters + keys) on the CPU."""

    def __init__(self, paramset=TOY, method=GINX, custom=None):
        L = lib()
        if custom is not None:
            self.h = L.bo_ctx_create_custom(*custom, method)
        else:
            self.h = L.bo_ctx_create(paramset, method)
        if not self.h:
            raise ValueError("bad oracle parameters")
        buf = (C.c_uint64 * len(P_NAMES))()
        L.bo_get_params(self.h, buf)
        self.params = dict(zip(P_NAMES, [int(v) for v in buf]))
        self.n = self.params["n"]
        self.N = self.params["N"]

    def close(self):
        if self.h:
            lib().bo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- keys ---------------------------------------------------------
    def keygen(self, seed=0x0FE5EED):
        lib().bo_keygen(self.h, seed_bytes(seed))

    def sk(self):
        s = np.zeros(self.n, dtype=np.int32)
        lib().bo_export_sk(self.h, _p(s))
        return s

    def z(self):
        z = np.zeros(self.N, dtype=np.int32)
        lib().bo_export_z(self.h, _p(z))
        return z

    def bsk(self):
        w = lib().bo_bsk_words(self.h)
        out = np.zeros(w, dtype=np.uint64)
        lib().bo_export_bsk(self.h, _p(out))
        return out

    def ksk(self):
        w = lib().bo_ksk_words(self.h)
        out = np.zeros(w, dtype=np.uint32)
        lib().bo_export_ksk(self.h, _p(out))
        return out

    # -- LWE ------------------------------------------------------------
    def ct(self):
        return np.zeros(self.n + 1, dtype=np.uint64)

    def encrypt(self, bit, index):
        ct = self.ct()
        lib().bo_encrypt(self.h, int(bit), int(index), _p(ct))
        return ct

    def decrypt(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        return lib().bo_decrypt(self.h, _p(ct))

    def noise(self, ct, bit):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        return lib().bo_noise(self.h, _p(ct), int(bit))

    def eval_not(self, ct):
        out = self.ct()
        lib().bo_eval_not(self.h, _p(np.ascontiguousarray(ct)), _p(out))
        return out

    def eval_bingate(self, gate, a, b):
        out = self.ct()
        lib().bo_eval_bingate(self.h, gate, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def bootstrap(self, a):
        out = self.ct()
        lib().bo_bootstrap(self.h, _p(np.ascontiguousarray(a)), _p(out))
        return out

    # -- staged ---------------------------------------------------------
    def gate_prep(self, gate, a, b):
        out = self.ct()
        lib().bo_gate_prep(self.h, gate, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def blind_rotate(self, gate, prep):
        acc = np.zeros(2 * self.N, dtype=np.uint64)
        lib().bo_blind_rotate(self.h, gate, _p(np.ascontiguousarray(prep)), _p(acc))
        return acc

    def extract_modswitch(self, acc):
        out = np.zeros(self.N + 1, dtype=np.uint64)
        lib().bo_extract_modswitch(self.h, _p(np.ascontiguousarray(acc)), _p(out))
        return out

    def keyswitch(self, lweN):
        out = self.ct()
        lib().bo_keyswitch(self.h, _p(np.ascontiguousarray(lweN)), _p(out))
        return out

    def modswitch_final(self, ks):
        out = self.ct()
        lib().bo_modswitch_final(self.h, _p(np.ascontiguousarray(ks)), _p(out))
        return out

    def ntt_forward(self, x):
        x = np.array(x, dtype=np.uint64)
        lib().bo_ntt_forward(self.h, _p(x))
        return x

    def ntt_inverse(self, x):
        x = np.array(x, dtype=np.uint64)
        lib().bo_ntt_inverse(self.h, _p(x))
        return x

    def signed_digit_decompose(self, ct):
        """ct: [2][N] coefficient form -> [2 dG][N] digits mod Q (row 2l + j = digit l of component j)."""
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, self.N)
        out = np.zeros((2 * self.params["dG"], self.N), dtype=np.uint64)
        lib().bo_signed_digit_decompose(self.h, _p(ct), _p(out))
        return out

    # -- batched ----------------------------------------------------------
    def eval_gates(self, pool, descs, nthreads=0):
        """pool: uint64 [slots, n+1] (modified in place); descs: list of (op,in0,in1,out,neg0,neg1)."""
        arr = (GateDesc * len(descs))(*[GateDesc(*d) for d in descs])
        assert pool.dtype == np.uint64 and pool.flags["C_CONTIGUOUS"]
        return lib().bo_eval_gates(self.h, _p(pool), len(descs), arr, nthreads)
