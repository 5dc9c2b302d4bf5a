"""Design study: what one span window of K1 costs in SIMT steps, counted by the host emulation of the shipped kernel
source built with -DMZ_STATS (walk steps = the slowest lane of every pass).  Not part of the product or the test suite.

    python tests/study/k1_steps.py [extra -D flags, e.g. -DMZ_SPAN_PRELIT=0]
"""
import ctypes as C
import os
import random
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402


def main():
    flags = sys.argv[1:]
    so = "/tmp/libemul_stats_%s.so" % (abs(hash(tuple(flags))) % 10 ** 8)
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DMZHIP_HOST_EMUL", "-DMZ_STATS",
                           "-I" + os.path.join(ROOT, "minizip-ng_amd", "csrc"), "-I" + os.path.join(ROOT, "include")] + flags +
                          [os.path.join(ROOT, "tests", "emul", "emul.cpp"), "-o", so])
    L = C.CDLL(so)
    L.emul_stats.restype = C.POINTER(C.c_ulonglong)
    text, desc = synth.bench_corpus()
    rnd = random.Random(5)
    total = 0
    for _ in range(200):
        o = rnd.randrange(0, len(text) - 65536)
        d = text[o:o + 65536]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z = co.compress(d) + co.flush()
        out = C.create_string_buffer(len(d) + 64)
        ol, iu, crc = C.c_uint32(), C.c_uint32(), C.c_uint32()
        st = L.emul_inflate(z, len(z), out, len(d), C.byref(ol), C.byref(iu), C.byref(crc))
        assert st == 0 and out.raw[:ol.value] == d
        total += len(d)
    s = L.emul_stats()
    win, passes, chunks = s[8], s[9], s[15]
    print("%s, 200 x 64 KiB at level 6; flags %s" % (desc, " ".join(flags) or "(default)"))
    print("windows %d (%.0f output bytes each), passes per window %.2f, chunks per window %.2f" % (win, s[10] / win, passes / win, chunks / win))
    print("chunks emitted on a pass the other lanes still count on (one lane per span): %.2f per window" % (s[21] / win))
    print("walk steps per window: %.1f in emitting passes + %.1f in counting passes = %.1f" % (s[7] / win, s[5] / win, (s[7] + s[5]) / win))
    print("block headers: %d code-length symbols through the 64-bit front end in %d steps, %d through the serial loop" % (s[16], s[18], s[17]))
    print("span walks: %d length codes decoded, %.2f %% of them with a distance code longer than the root table" % (s[19], 100.0 * s[20] / max(1, s[19])))
    print("pieces per window %.0f, of them near %.0f, near rounds %.1f" % (s[12] / win, s[13] / win, s[14] / win))


if __name__ == "__main__":
    main()
