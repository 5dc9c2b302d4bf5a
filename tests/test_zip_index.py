"""CPU tests: the C bulk central-directory indexer (mzhip_zip_index_mem) against the reference's own
entry walk (oracle/_ref: mz_zip_goto_first/next_entry + mz_zip_entry_get_info + local-header skip)."""
import importlib
import os
import tempfile
import zipfile

import numpy as np
import pytest

import oracle
from tests import synth

archive = importlib.import_module("minizip-ng_amd.archive")
needs_ref = pytest.mark.skipif(not (oracle.have_ref() or os.path.exists("/root/reference/mz_zip.c")),
                               reason="oracle/_ref not built and /root/reference absent")


@needs_ref
def test_index_matches_reference_walk():
    ref = oracle.ref()
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(8)
    with tempfile.TemporaryDirectory() as tmp:
        for method, level, n, size in ((8, 6, 300, 20000), (0, 0, 1000, 3000), (14, 6, 10, 40000), (8, 1, 1, 0)):
            lens = rnd.randint(0, size + 1, size=n).astype(np.int32)
            offs = rnd.randint(0, len(c) - size - 1, size=n).astype(np.int64)
            path = os.path.join(tmp, "a%d_%d.zip" % (method, n))
            ref.zip_write(path, c, offs, lens, method=method, level=level)
            want = ref.zip_index(path)
            got = archive.index_file(path)
            assert got.shape == want.shape and (got == want).all(), (method, n)


@needs_ref
def test_index_zip64_many_entries():
    """> 65 535 entries forces the ZIP64 end-of-central-directory records (mz_zip.c:1011-1059)."""
    ref = oracle.ref()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "many.zip")
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
            for i in range(70000):
                z.writestr("e/%06d" % i, b"x" * (i % 7))
        got = archive.index_file(path)
        assert len(got) == 70000
        want = ref.zip_index(path)
        assert (got == want).all()


def test_index_python_zipfile_and_errors():
    c = synth.corpus()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "py.zip")
        with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
            for i in range(50):
                z.writestr("f%d.txt" % i, c[i * 1000:i * 1000 + 5000 + i])
            z.comment = b"archive comment " * 10
        t = archive.index_file(path)
        raw = open(path, "rb").read()
        with zipfile.ZipFile(path) as z:
            for i, info in enumerate(z.infolist()):
                assert t[i, archive.COL_CRC] == info.CRC and t[i, archive.COL_CSIZE] == info.compress_size
                assert t[i, archive.COL_USIZE] == info.file_size and t[i, archive.COL_METHOD] == 8
                p = int(t[i, archive.COL_PAYLOAD])
                import zlib
                assert zlib.decompress(raw[p:p + info.compress_size], -15) == c[i * 1000:i * 1000 + 5000 + i]
        mz = importlib.import_module("minizip-ng_amd")
        with pytest.raises(mz.MzHipError):
            archive.index_bytes(b"not a zip file at all, just some bytes............")


def test_shard_bounds_balanced():
    t = np.zeros((1000, 8), dtype=np.int64)
    rnd = np.random.RandomState(1)
    t[:, archive.COL_CSIZE] = rnd.randint(0, 30000, 1000)
    t[:, archive.COL_USIZE] = t[:, archive.COL_CSIZE] * 3
    for world in (1, 2, 3, 8):
        b = archive.shard_bounds(t, world)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all() and len(b) == world + 1
        w = (t[:, archive.COL_CSIZE] + t[:, archive.COL_USIZE]).astype(float)
        loads = [w[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(loads) <= 1.15 * (sum(loads) / world) + 1


def test_hash_extrafield_parse():
    """Hash extrafield (0x1a51, doc/mz_extrafield.md) of each central-directory record: first one wins, other
    fields are skipped, entries without one report algorithm 0 (mz_zip_reader_entry_get_first_hash)."""
    import hashlib
    import io
    import struct
    import zipfile

    archive = importlib.import_module("minizip-ng_amd.archive")
    d = b"hello hash"
    hx = struct.pack("<HHHH", 0x1A51, 36, 23, 32) + hashlib.sha256(d).digest()
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w") as z:
        for name, ex in (("a", hx), ("b", b""), ("c", struct.pack("<HHI", 0x7875, 4, 0) + hx),
                         ("d", struct.pack("<HHHH", 0x1A51, 24, 20, 20) + hashlib.sha1(d).digest() + hx)):
            zi = zipfile.ZipInfo(name)
            zi.extra = ex
            z.writestr(zi, d)
    raw = buf.getvalue()
    t = archive.index_bytes(raw)
    alg, dsz, dig = archive.hash_fields(raw, t)
    assert alg.tolist() == [23, 0, 23, 20] and dsz.tolist() == [32, 0, 32, 20]
    assert dig[0, :32].tobytes() == hashlib.sha256(d).digest() == dig[2, :32].tobytes()
    assert dig[3, :20].tobytes() == hashlib.sha1(d).digest()
