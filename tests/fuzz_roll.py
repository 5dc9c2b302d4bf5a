"""Differential fuzz of the auto-prime (not a pytest): random archives through the unmodified reader loop on the drop-in against the
all-reference reader -- sizes, entry counts, methods, reader threads, whole-image limit (so that some archives are imaged whole, most
are rolled over in windows of many sizes), imaging through the archive's own descriptor or through the readers' streams, a
clear now and then, entries read front to back, back to front or in a random order, sometimes one flipped payload byte (that entry must fail on both sides, every other one must not).

    python tests/fuzz_roll.py [cases=200] [seed=1] [library]

library: tests/emul/_build/libmockdrop.so (the product's cache and shims on the host emulation: any container) or, on a GPU box,
integration/_build/libmzhipdrop.so (the default when it exists and a device is there)."""
import ctypes as C
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = sys.argv[3] if len(sys.argv) > 3 else None
if lib is None:
    lib = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    try:
        import importlib

        importlib.import_module("minizip-ng_amd").require_gpu()
    except Exception:  # noqa: BLE001
        lib = os.path.join(ROOT, "tests", "emul", "_build", "libmockdrop.so")
on_device = "mzhipdrop" in lib
hip, ref = oracle.MzDriver(lib), oracle.ref()
L = hip.L
if on_device:
    import importlib

    L = importlib.import_module("minizip-ng_amd").lib()
L.mzhip_prime_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
L.mzhip_autoprime_stats.argtypes = [C.POINTER(C.c_uint64)] * 4
L.mzhip_last_error.restype = C.c_char_p
rnd = random.Random(seed)
c = np.frombuffer(synth.corpus(), dtype=np.uint8)
scale = 8 if on_device else 1  # the emulation decodes ~30 MB/s
bad = 0
tot_entries = tot_windows = 0
with tempfile.TemporaryDirectory() as tmp:
    for it in range(cases):
        method = rnd.choice((8, 8, 8, 8, 14, 95))
        n = rnd.randint(3, 120 if method != 8 else 400 * scale)
        top = min(rnd.choice((300, 3000, 20000, 70000 if method == 8 else 30000)) * (scale if method == 8 else 1), len(c) // 2)
        lens = np.array([rnd.choice((0, 1, rnd.randint(0, top), rnd.randint(0, top))) for _ in range(n)], dtype=np.int32)
        offs = np.array([rnd.randint(0, len(c) - int(top) - 1) for _ in range(n)], dtype=np.int64)
        path = os.path.join(tmp, "f%d.zip" % it)
        ref.zip_write(path, c, offs, lens, method=method, level=rnd.choice((1, 6, 9)))
        table = ref.zip_index(path)
        victim, cd_victim = -1, -1
        if rnd.random() < 0.25:
            cand = [i for i in range(n) if table[i, 3] > 24]
            if cand:
                victim = rnd.choice(cand)
                cd_victim = int(table[victim, 6])
                raw = bytearray(open(path, "rb").read())
                raw[int(table[victim, 7]) + rnd.randint(8, int(table[victim, 3]) - 8)] ^= 1 << rnd.randrange(8)
                open(path, "wb").write(raw)
        cd = table[:, 6].copy()
        if rnd.random() < 0.3:
            # not front to back: a random order (windows are primed, evicted and primed again; the thrash guard may give some up:
            # slower, never different), or back to front
            perm = np.array(rnd.sample(range(n), n)) if rnd.random() < 0.6 else np.arange(n)[::-1]
            cd, lens, table = cd[perm], lens[perm], table[perm]
        out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        pad = 1
        if victim >= 0:
            # a stream with a flipped bit may decode to MORE than its declared size before it is refused (nothing bounds a DEFLATE
            # stream's output, mz_strm_zlib.c:116-193), and the driver copies what it is handed: the victim's bytes go behind
            # everybody else's, with room to spare, so that they cannot land in a neighbour's slot
            where = int(np.nonzero(table[:, 6] == cd_victim)[0][0])
            out_off[where] = int(lens.sum())
            pad = int(lens[where]) + (16 << 20)
        o_ref = np.zeros(int(lens.sum()) + pad, dtype=np.uint8)
        o_hip = np.zeros(int(lens.sum()) + pad, dtype=np.uint8)
        _, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=1, own_crc=False, out=o_ref, out_off=out_off)
        limit = rnd.choice(("32k", "64k", "128k", "512k", "2", "512"))
        os.environ["MZHIP_AUTOPRIME"] = limit
        os.environ["MZHIP_AUTOPRIME_FD"] = rnd.choice(("0", "1"))
        threads = rnd.choice((1, 1, 2, 3, 6))
        mapped = rnd.random() < 0.3
        passes = rnd.choice((1, 1, 2))
        for p in range(passes):
            o_hip[:] = 0
            _, crc_h, ulen_h, st_h = hip.zip_read_all(path, cd, nthreads=threads, own_crc=False, out=o_hip, out_off=out_off, mapped=mapped)
            good = st_r == 0
            same = bool(((st_h == 0) == good).all() and (crc_h[good] == crc_r[good]).all() and (ulen_h[good] == ulen_r[good]).all())
            if same:  # bytes of the entries that decode (a failed entry's partial bytes may differ in length)
                for i in np.nonzero(good)[0]:
                    a, b = int(out_off[i]), int(out_off[i]) + int(lens[i])
                    if not np.array_equal(o_hip[a:b], o_ref[a:b]):
                        same = False
                        break
            if victim >= 0 and st_r[victim] == 0:
                pass  # (the flip fell where the stream does not care: both sides decode it)
            if not same:
                bad += 1
                why = "statuses" if not ((st_h == 0) == good).all() else "crc" if not (crc_h[good] == crc_r[good]).all() else "sizes" if not (ulen_h[good] == ulen_r[good]).all() else "bytes"
                idx = [int(i) for i in np.nonzero(good)[0] if not np.array_equal(o_hip[int(out_off[i]):int(out_off[i]) + int(lens[i])], o_ref[int(out_off[i]):int(out_off[i]) + int(lens[i])])][:5]
                print("  differs in: %s; entries with other bytes %s; the failed entries ref %s hip %s; sizes there %s" % (why, idx, np.nonzero(st_r)[0][:5], np.nonzero(st_h)[0][:5], lens[idx] if idx else ""), flush=True)
                print("case %d pass %d: method %d, %d entries, limit %s, %d threads, mapped %s, fd %s: statuses ref %s hip %s, last error %r" % (
                    it, p, method, n, limit, threads, mapped, os.environ["MZHIP_AUTOPRIME_FD"], st_r[st_r != 0][:5], st_h[st_h != 0][:5], L.mzhip_last_error()), flush=True)
        w = [C.c_uint64() for _ in range(4)]
        L.mzhip_autoprime_stats(*[C.byref(x) for x in w])
        tot_entries += n * passes
        tot_windows = w[0].value
        if rnd.random() < 0.3:
            L.mzhip_prime_clear()
        os.remove(path)
    e, h, m = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.mzhip_prime_stats(C.byref(e), C.byref(h), C.byref(m))
print("fuzz_roll: %d archives (%d entry reads, %d windows primed so far, cache hits since the last clear %d), %d differences" % (cases, tot_entries, tot_windows, h.value, bad),
      flush=True)
sys.exit(1 if bad else 0)
