"""AP method: what the digit-selected keys cost -- one saturated launch of distinct ciphertexts (every bootstrap walks its own
RGSW ciphertexts) against one of identical ciphertexts (all workgroups on the same keys).  Development measurement, GPU only."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
# usage: ap_key_locality.py [PARAMSET [BOOTSTRAPS]]   (default STD128_OPT 6144; STD192 1024 = the config-5 kernel)
ps = sys.argv[1] if len(sys.argv) > 1 else "STD128_OPT"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
c = bce.BinFHEContext(getattr(bce, ps), bce.AP); c.KeyGen(42)
c.pool_reserve(3 * nb)
bits = np.random.default_rng(0).integers(0, 2, 2 * nb).astype(np.uint8)
c.Encrypt(bits, np.arange(2 * nb), enc_index_base=0)
descs = bce.make_descs([(bce.NAND, 2 * i, 2 * i + 1, 2 * nb + i) for i in range(nb)])
def run(tag):
    c.EvalGates(descs); c.synchronize(); c.timing_reset()
    for _ in range(3): c.EvalGates(descs)
    c.synchronize(); t = c.timing()
    print(tag, "%.2f ms per launch -> %.0f bootstraps/s" % (t["blind_rotate_ms"] / 3, nb / (t["blind_rotate_ms"] / 3) * 1e3), flush=True)
run("distinct ciphertexts (digit-selected keys differ per bootstrap):")
two = c.lwe_read(np.array([0, 1], dtype=np.uint32))
c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.tile(two, (nb, 1)))
run("identical ciphertexts (every bootstrap walks the same keys):   ")
