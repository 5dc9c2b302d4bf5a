"""Design study behind K6's chain pass (DESIGN 3 K6 "Round 4"): ratio of the method-14 encoder in the host emulation of the
shipped source, against liblzma's preset 6, over the bytes the chain pass hashes (MZ_LZE_FAR_NGRAM, a build per value) and
the links the block parse follows (a run-time knob of the emulation), on config-4 entries, the reference's C sources, an ELF
and the corpus itself; and liblzma's own fast / normal modes over a hash chain for scale.  Not part of the product or the
test suite.

    python tests/study/k6_chain.py [ngram ...]        (default: 4 6 7 8)
"""
import ctypes as C
import glob
import lzma
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

_u8p = C.POINTER(C.c_uint8)


def build(ngram):
    so = "/tmp/libemul_k6_n%d.so" % ngram
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DMZHIP_HOST_EMUL", "-DMZ_LZE_FAR_NGRAM=%du" % ngram,
                           "-I" + os.path.join(ROOT, "minizip-ng_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "emul", "emul.cpp"), "-o", so])
    L = C.CDLL(so)
    L.emul_lzma_encode_ways.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.emul_set_far_depth.argtypes = [C.c_uint32]
    return L


def ours(L, d, depth):
    L.emul_set_far_depth(depth)
    a = np.frombuffer(d, dtype=np.uint8).copy()
    out = np.zeros(len(d) + len(d) // 8 + 4096, dtype=np.uint8)
    ol, crc = C.c_uint32(), C.c_uint32()
    assert L.emul_lzma_encode_ways(C.cast(a.ctypes.data, _u8p), len(d), 0, 4, C.cast(out.ctypes.data, _u8p), len(out), C.byref(ol), C.byref(crc)) == 0
    z = out[:ol.value].tobytes()
    assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == d
    return len(z)


def main():
    grams = [int(a) for a in sys.argv[1:]] or [4, 6, 7, 8]
    corpus = synth.bench_corpus()[0]
    sets = {"config-4 entries": synth.markov_entries(2, 1 << 20, 77, corpus), "corpus": [corpus]}
    src = b"".join(open(f, "rb").read() for f in sorted(glob.glob("/root/reference/*.c") + glob.glob("/root/reference/*.h")))
    if src:
        sets["reference C sources"] = [src[:4 << 20]]
    if os.path.exists("/usr/bin/python3.10"):
        sets["ELF (python3.10, 4 MiB)"] = [open("/usr/bin/python3.10", "rb").read()[:4 << 20]]
    l6 = {k: sum(len(lzma.compress(d, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "preset": 6}])) for d in v) for k, v in sets.items()}
    tot = {k: sum(len(d) for d in v) for k, v in sets.items()}
    print("liblzma preset 6: " + ", ".join("%s %.4f" % (k, l6[k] / tot[k]) for k in sets))
    for g in grams:
        L = build(g)
        for depth in (1, 4, 8, 16, 32):
            row = {k: sum(ours(L, d, depth) for d in v) for k, v in sets.items()}
            print("%d bytes hashed, %2d links: " % (g, depth) + ", ".join("%s %.4f (x%.3f)" % (k, row[k] / tot[k], row[k] / l6[k]) for k in sets), flush=True)
    ds = sets["config-4 entries"]
    base = {"id": lzma.FILTER_LZMA1, "dict_size": 8 << 20, "lc": 3, "lp": 0, "pb": 2}
    for name, kw in (("fast, hc4, depth 4", dict(mode=lzma.MODE_FAST, mf=lzma.MF_HC4, nice_len=273, depth=4)),
                     ("fast, hc4, depth 16", dict(mode=lzma.MODE_FAST, mf=lzma.MF_HC4, nice_len=273, depth=16)),
                     ("fast, hc4, depth 64", dict(mode=lzma.MODE_FAST, mf=lzma.MF_HC4, nice_len=273, depth=64)),
                     ("normal, hc4, depth 4", dict(mode=lzma.MODE_NORMAL, mf=lzma.MF_HC4, nice_len=64, depth=4)),
                     ("normal, hc4, depth 16", dict(mode=lzma.MODE_NORMAL, mf=lzma.MF_HC4, nice_len=64, depth=16)),
                     ("normal, bt4 (= preset 6)", dict(mode=lzma.MODE_NORMAL, mf=lzma.MF_BT4, nice_len=64))):
        f = dict(base)
        f.update(kw)
        print("liblzma %-26s config-4 entries %.4f" % (name + ":", sum(len(lzma.compress(d, format=lzma.FORMAT_RAW, filters=[f])) for d in ds) / tot["config-4 entries"]))


if __name__ == "__main__":
    main()
