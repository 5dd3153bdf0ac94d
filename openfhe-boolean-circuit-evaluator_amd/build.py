"""In-tree build of libbce_amd.so (HIP kernels + C ABI + host circuit runtime) for gfx950.

hipcc cross-compiles without a GPU.  The shared object stays next to this file so that it
travels with the repo snapshot and shows up as in-tree native code when loaded.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libbce_amd.so")
OBJ = os.path.join(HERE, "_obj")

SOURCES = ["kernels.hip", "kernels64.hip", "keygen.hip", "engine.cpp", "keyfile.cpp", "rccl_xchg.cpp", "bristol.cpp", "circuit.cpp", "circuit_capi.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wall",
         "-Wno-unused-result", "-Wno-unused-value"]
FLAGS += os.environ.get("BCE_EXTRA_FLAGS", "").split()  # development builds only (e.g. -DBCE_PHASE_PROF)


def _deps():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    paths += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    paths.append(os.path.join(HERE, "..", "tools", "openfhe_export", "bce_keyfile.h"))
    return paths


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    newest = max(os.path.getmtime(p) for p in _deps())
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OBJ, s + ".o")
        objs.append(o)
        if (not force) and os.path.exists(o) and os.path.getmtime(o) >= newest:
            continue
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out:
            sys.stderr.write(out.decode())
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs + ["-lpthread", "-ldl"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
