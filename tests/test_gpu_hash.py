"""SURVEY 8(f) row f4 on the C path: the reference's reader verifies an entry's Hash extra field (0x1a51) by hashing what it
reads and comparing at mz_zip_reader_entry_close (mz_zip_rw.c:409-451,462-467).  mzhip_prime_*() computes those digests on
the device in the pass that decodes the archive; libmzhip.so's mz_crypt_sha_* (shim_sha.c) put them where the reference
looks.  The archive is written by the crypto-enabled ALL-REFERENCE build (oracle/_ref/libmzref_crypto.so: its writer adds a
SHA-256 field to every entry); it is read through mz_zip_reader_entry_open / _read / _close of the unmodified mz_zip_rw.c on
integration/_build/libmzhipdrop_crypto.so -- un-primed (the renamed reference SHA behind the shim), primed (device digests),
and with one digest tampered with (MZ_CRC_ERROR on exactly that entry, as the all-reference reader says)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPC = os.path.join(ROOT, "integration", "_build", "libmzhipdrop_crypto.so")
REFC = os.path.join(ROOT, "oracle", "_ref", "libmzref_crypto.so")

# run in a process of its own: libmzhip.so binds the (weak) mz_ref_crypt_sha_* of the drop-in library when it is loaded,
# so the drop-in has to be what loads it
PROG = r"""
import sys, os, json, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np, oracle
hip = oracle.MzDriver(%(drop)r)
L = hip.L
L.mzhip_prime_file.restype = C.c_int64
L.mzhip_prime_file.argtypes = [C.c_char_p]
L.mzhip_sha_primed_digests.restype = C.c_uint64
L.mzhip_prime_hash_stats.argtypes = [C.POINTER(C.c_uint64)] * 2
def stats():
    a, b = C.c_uint64(0), C.c_uint64(0)
    L.mzhip_prime_hash_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value), int(L.mzhip_sha_primed_digests())
out = {}
for name, path in (("good", %(good)r), ("bad", %(bad)r)):
    L.mzhip_prime_clear()
    s0 = stats()
    os.environ["MZHIP_AUTOPRIME"] = "0"                      # (the reader would prime the archive itself: on by default since round 5)
    st, ul = hip.zip_reader_walk(path)                       # nothing primed: the reference's SHA behind the shim
    del os.environ["MZHIP_AUTOPRIME"]
    s1 = stats()
    primed = L.mzhip_prime_file(path.encode())
    st2, ul2 = hip.zip_reader_walk(path)                     # primed: digests from the device
    s2 = stats()
    out[name] = dict(unprimed=[int(x) for x in st], ulen=[int(x) for x in ul], primed_entries=int(primed), primed=[int(x) for x in st2],
                     ulen2=[int(x) for x in ul2], d_unprimed=[s1[i] - s0[i] for i in range(3)], d_primed=[s2[i] - s1[i] for i in range(3)])
L.mzhip_prime_clear()
print(json.dumps(out))
"""


def _tamper_digest(raw, table, e):
    """flip one bit of the digest in the central directory's Hash field of entry e"""
    b = bytearray(raw)
    pos = int(table[e, 6])
    fn = b[pos + 28] | b[pos + 29] << 8
    ex = b[pos + 30] | b[pos + 31] << 8
    q, end = pos + 46 + fn, pos + 46 + fn + ex
    while q + 4 <= end:
        fid, fsz = b[q] | b[q + 1] << 8, b[q + 2] | b[q + 3] << 8
        if fid == 0x1A51:
            b[q + 8 + 5] ^= 0x40
            return bytes(b)
        q += 4 + fsz
    raise AssertionError("entry %d has no Hash extra field" % e)


def test_reader_hash_verification_uses_device_digests(tmp_path):
    if not (os.path.exists(DROPC) and os.path.exists(REFC)):
        pytest.skip("crypto-enabled drop-in / reference libraries missing (built where /root/reference exists)")
    import importlib

    importlib.import_module("minizip-ng_amd").require_gpu()
    refc = oracle.MzDriver(REFC)
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(5)
    n = 40
    lens = rnd.randint(1, 200000, size=n).astype(np.int32)
    lens[:3] = (1, 65535, 65536)
    offs = rnd.randint(0, len(c) - 200000, size=n).astype(np.int64)
    good = str(tmp_path / "good.zip")
    refc.zip_write(good, c, offs, lens, method=8, level=6)     # the reference writer with crypto: SHA-256 field per entry
    table = refc.zip_index(good)
    raw = open(good, "rb").read()
    bad = str(tmp_path / "bad.zip")
    victim = 7
    open(bad, "wb").write(_tamper_digest(raw, table, victim))
    st_ref, ul_ref = refc.zip_reader_walk(good)
    st_bad, _ = refc.zip_reader_walk(bad)
    assert (st_ref == 0).all() and (ul_ref == lens).all()
    want_bad = [0] * n
    want_bad[victim] = -105                                      # MZ_CRC_ERROR, mz_zip_rw.c:448-449
    assert [int(x) for x in st_bad] == want_bad
    r = subprocess.run([sys.executable, "-c", PROG % dict(root=ROOT, drop=DROPC, good=good, bad=bad)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g, b = got["good"], got["bad"]
    # un-primed: same verdicts through the renamed reference SHA; no device digest involved
    assert g["unprimed"] == [0] * n and g["ulen"] == [int(x) for x in lens] and g["d_unprimed"] == [0, 0, 0]
    assert b["unprimed"] == want_bad
    # primed: every entry's digest was computed and verified on the device, and all n mz_crypt_sha_end calls were answered with it
    assert g["primed_entries"] == n and g["primed"] == [0] * n and g["ulen2"] == [int(x) for x in lens]
    assert g["d_primed"] == [n, 0, n], g["d_primed"]
    # tampered field: the device finds the mismatch (the entry is not served from the cache), the reader reports MZ_CRC_ERROR on
    # exactly that entry -- and the other n - 1 are answered from device digests
    assert b["primed"] == want_bad and b["d_primed"] == [n, 1, n - 1], b["d_primed"]
