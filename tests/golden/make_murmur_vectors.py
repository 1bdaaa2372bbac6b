"""Generates tests/golden/murmur3_vectors.json.

Independent cross-check for the k-mer hash: Austin Appleby's public-domain
MurmurHash3_x64_128 as shipped inside scikit-learn in this container (NOT part of the pyfastani
reference) is compiled where it lies and run on seeded k-mers for k in {1..33} with seed 42; the low 32
bits of the first output word are what skch::CommonFunc::getHash returns.
"""
import ctypes
import json
import os
import subprocess
import tempfile

import numpy as np
import sklearn

SRC = os.path.join(os.path.dirname(sklearn.__file__), "utils", "src", "MurmurHash3.cpp")
HERE = os.path.dirname(os.path.abspath(__file__))

with tempfile.TemporaryDirectory() as tmp:
    shim = os.path.join(tmp, "shim.cpp")
    with open(shim, "w") as f:
        f.write('#include "MurmurHash3.h"\n#include <stdint.h>\n'
                'extern "C" uint32_t low32(const void* key, int len, uint32_t seed) {\n'
                '  uint64_t out[2]; MurmurHash3_x64_128(key, len, seed, out); return (uint32_t)out[0]; }\n')
    so = os.path.join(tmp, "mm.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.dirname(SRC), SRC, shim, "-o", so])
    lib = ctypes.CDLL(so)
    lib.low32.restype = ctypes.c_uint32
    lib.low32.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32]
    rng = np.random.Generator(np.random.PCG64(42))
    vectors = []
    for k in list(range(1, 34)) + [48, 64, 100]:
        for _ in range(6):
            kmer = bytes(np.frombuffer(b"ACGTNacgtRYKM", dtype=np.uint8)[rng.integers(0, 13, k)])
            vectors.append({"kmer": kmer.decode(), "hash": int(lib.low32(kmer, k, 42))})
    for aa in ["MPFSRRTSTASAAVAF", "LLGACGEGDPVMLEAV", "ACDEFGHIKLMNPQRSTVWY"]:
        vectors.append({"kmer": aa, "hash": int(lib.low32(aa.encode(), len(aa), 42))})
with open(os.path.join(HERE, "murmur3_vectors.json"), "w") as f:
    json.dump(vectors, f, indent=0)
print(len(vectors), "vectors")
