"""Probe: LZMA/XZ host encodes in fresh processes, sequences of small inputs (diagnosing a CLI append failure)."""
import ctypes as C, lzma, os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CD = open(os.path.join(ROOT, "tests", "golden", "cd122.bin"), "rb").read()

def child(seq, xz):
    L = C.CDLL(os.path.join(ROOT, "minizip-ng_amd", "_build", "libmzhip.so"))
    fn = L.mzhip_xz_encode_host if xz else L.mzhip_lzma_encode_host
    fn.restype = C.c_int32
    res = []
    for d in seq:
        cap = len(d) + len(d) // 8 + 4096
        out = (C.c_uint8 * cap)()
        ol, crc = C.c_uint32(), C.c_uint32()
        st = fn(d, len(d), out, cap, C.byref(ol), C.byref(crc))
        z = bytes(out[:ol.value])
        try:
            back = lzma.decompress(z) if xz else lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE)
            ok = back == d
        except Exception as e:
            ok = repr(e)
        res.append((len(d), st, ol.value, crc.value == zlib.crc32(d), ok))
    print(res)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        which, xz = sys.argv[1], int(sys.argv[2])
        seqs = {"a": [b"1", CD], "b": [CD], "c": [CD, CD, CD], "d": [b"", b"1", CD, b"1", CD]}
        child(seqs[which], xz)
    else:
        bad = 0
        for rep in range(int(os.environ.get("REPS", "12"))):
            for which in os.environ.get("SEQS", "abcd"):
                for xz in (0, 1):
                    r = subprocess.run([sys.executable, __file__, which, str(xz)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                    line = r.stdout.decode().strip().splitlines()[-1] if r.stdout else ""
                    good = r.returncode == 0 and "False" not in line and "Error" not in line and "LZMAError" not in line
                    if not good:
                        bad += 1
                        print("BAD", rep, which, xz, r.returncode, r.stdout.decode())
        print("done, bad =", bad)
