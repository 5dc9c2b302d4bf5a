"""GPU parity tests for SURVEY 8(f) rows 3 and 4: the .xz kernel (ZIP method 95: container + LZMA2 chunks + block
checks, CRC-32 of the output fused) through the C ABI (mzhip_xz_batch / mzhip_xz_host), through the drop-in
mz_stream_lzma and through the unmodified mz_zip reader; and the SHA-1 / SHA-224 / SHA-256 batch kernel.
Checker: the oracle restatement (pinned to the live reference and the xz.zip fixture in tests/test_oracle.py),
the compiled reference where oracle/_ref travelled, hashlib for the digests."""
import ctypes as C
import hashlib
import os
import random
import tempfile
import zlib

import numpy as np
import pytest

import oracle
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_xz_batch.restype = C.c_int32
    L.mzhip_xz_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
    L.mzhip_sha_batch.restype = C.c_int32
    L.mzhip_sha_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_uint32] + [C.c_void_p] * 2
    return gpu_util


def run_xz(gpu, pays, caps, max_out=None):
    import torch

    b = gpu.make_batch(pays, caps)
    n = len(pays)
    dev = b["d_in"].device
    out_len, in_used, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    mo = torch.tensor(max_out if max_out is not None else [-1] * n, dtype=torch.int64, device=dev)
    rc = gpu.mz.lib().mzhip_xz_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                     b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                     mo.data_ptr(), n, out_len.data_ptr(), in_used.data_ptr(), crc.data_ptr(),
                                     status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    return (b, b["d_out"].cpu().numpy(), out_len.cpu().numpy(), in_used.cpu().numpy(), gpu.mz.u32(crc),
            status.cpu().numpy())


def _unsupported(name):
    return "lp4" in name or "lc4" in name or "lc1lp3" in name


def test_xz_batch_cases_and_fixture(gpu, fixtures):
    cases = synth.xz_cases()
    fx = [e for e in fixtures if e["method"] == 95]
    pays = [x + b"tail" for _, _, x in cases] + [e["payload"] for e in fx]
    caps = [len(d) + 64 for _, d, _ in cases] + [e["usize"] + 4 for e in fx]
    b, h_out, out_len, in_used, crc, status = run_xz(gpu, pays, caps)
    for i, (name, d, x) in enumerate(cases):
        so, uo, oo = oracle.xz_decode(pays[i], caps[i])
        assert (status[i], in_used[i], out_len[i]) == (so, uo, len(oo)) == (0, len(x), len(d)), name
        assert gpu.entry_bytes(b, h_out, i, len(d)) == d, name
        assert crc[i] == zlib.crc32(d), name
    for j, e in enumerate(fx):
        i = len(cases) + j
        assert (status[i], in_used[i], out_len[i], crc[i]) == (0, e["csize"], e["usize"], e["crc"])
        assert in_used[i] == e["ref"]["total_in"] and out_len[i] == e["ref"]["total_out"]
    assert len(fx) >= 1
    assert sum(n.startswith("filter/") for n, _, _ in cases) >= 70     # Delta / BCJ chains in front of LZMA2 included


def test_xz_filter_chains_liblzma_refuses(gpu):
    """A misaligned BCJ start offset, an unknown filter, LZMA2 in front, Delta last: LZMA_OPTIONS_ERROR in liblzma,
    MZ_DATA_ERROR through mz_stream_lzma_read (mz_strm_lzma.c:236-237), -3 here."""
    bad = synth.xz_bad_chain_cases()
    b, h_out, out_len, in_used, crc, status = run_xz(gpu, [x for _, x in bad], [10000] * len(bad))
    for i, (name, x) in enumerate(bad):
        assert status[i] == -3 and oracle.xz_decode(x, 10000)[0] == -3, name


def test_xz_clamp_and_out_cap(gpu):
    name, d, x = synth.xz_cases()[0]
    b, h_out, out_len, in_used, crc, status = run_xz(gpu, [x, x, x], [len(d) + 8, len(d) - 1, len(d)], [3000, -1, -1])
    assert (status[0], out_len[0], crc[0]) == (0, 3000, zlib.crc32(d[:3000]))       # TOTAL_OUT_MAX, mz_strm_lzma.c:214-215
    assert status[1] == -200 and status[2] == 0 and crc[2] == zlib.crc32(d)


def test_xz_differential_fuzz_batch(gpu):
    """2500 corrupted / truncated .xz streams in one launch vs the oracle: the same accept / reject decision and
    error class; on accept the same bytes, consumed input and CRC."""
    rnd = random.Random(21)
    bases = [x for n, d, x in synth.xz_cases() if 0 < len(d) <= 100000 and not _unsupported(n)]
    pays = []
    for it in range(2500):
        x = bytearray(rnd.choice(bases))
        k = rnd.randrange(5)
        if k == 0:
            x[rnd.randrange(len(x))] ^= 1 << rnd.randrange(8)
        elif k == 1:
            x[rnd.randrange(len(x))] = rnd.randrange(256)
        elif k == 2:
            del x[rnd.randrange(1, len(x)):]
        elif k == 3:
            x[rnd.randrange(min(len(x), 40))] = rnd.randrange(256)
        else:
            x[-rnd.randrange(1, 40)] = rnd.randrange(256)
        pays.append(bytes(x))
    b, h_out, out_len, in_used, crc, status = run_xz(gpu, pays, [200000] * len(pays))
    n_ok = 0
    for i, x in enumerate(pays):
        so, uo, oo = oracle.xz_decode(x, 200000)
        if so == 0:
            n_ok += 1
            assert (status[i], in_used[i], out_len[i], crc[i]) == (0, uo, len(oo), zlib.crc32(oo)), i
            assert gpu.entry_bytes(b, h_out, i, len(oo)) == oo, i
        elif so == -109:
            assert status[i] in (-109, -3), (i, status[i])
        else:
            assert status[i] == so or (status[i], so) == (-109, -3), (i, status[i], so)
    assert n_ok >= 3


def test_sha_batch(gpu):
    import torch

    rnd = np.random.RandomState(8)
    kat = b"the quick and lazy fox did his thang"       # the reference's KAT string, test/test_crypt.cc:26
    datas = [b"", b"a", b"abc", kat, rnd.bytes(55), rnd.bytes(56), rnd.bytes(63), rnd.bytes(64), rnd.bytes(65),
             rnd.bytes(119), rnd.bytes(120), synth.corpus()[:200000]]
    datas += [rnd.bytes(int(n)) for n in rnd.randint(0, 3000, size=700)] + synth.slices(300, 65536, 5)
    b = gpu.make_batch(datas, [1] * len(datas), align=1)
    n = len(datas)
    for alg, fn in ((20, hashlib.sha1), (22, hashlib.sha224), (23, hashlib.sha256), (24, hashlib.sha384), (25, hashlib.sha512)):
        stride = 64 if alg >= 24 else 32
        dg = torch.zeros(n * stride, dtype=torch.uint8, device=b["d_in"].device)
        assert gpu.mz.lib().mzhip_sha_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(), n, alg,
                                            dg.data_ptr(), None) == 0
        torch.cuda.synchronize()
        h = dg.cpu().numpy().reshape(n, stride)
        sz = fn().digest_size
        for i, d in enumerate(datas):
            assert h[i, :sz].tobytes() == fn(d).digest() and not h[i, sz:].any(), (alg, i, len(d))
    h = hashlib  # KATs of test/test_crypt.cc:50-116
    assert h.sha1(kat).hexdigest() == "3efb8392b6cd8e14bd76bd08081521dc73df418c"
    assert h.sha256(kat).hexdigest() == "7a31ea0848525f7ebfeec9ee532bcc5d6d26772427e097b86cf440a56546541c"
    assert gpu.mz.lib().mzhip_sha_batch(None, None, None, 1, 10, None, None) == -109       # MD5: not served


@pytest.fixture(scope="module")
def libs(gpu):
    if not os.path.exists(DROP) or not oracle.have_ref():
        pytest.skip("drop-in / reference builds missing (built where /root/reference exists)")
    return oracle.MzDriver(DROP), oracle.ref()


def test_xz_stream_parity(libs):
    """mz_stream_lzma (method 95) on the HIP backend vs on liblzma: read() sequences, TOTAL_IN/OUT, close/error."""
    hip, ref = libs
    keys = ("rets", "out", "total_in", "total_out", "close", "open")
    for name, d, x in synth.xz_cases():
        if _unsupported(name) and len(d) > 1:
            continue
        for extra, kw in ((b"", {}), (b"tail" * 9000, {}), (b"", dict(max_in=len(x), max_out=len(d)))):
            a = hip.stream_decode(95, x + extra, len(d) + 64, **kw)
            b = ref.stream_decode(95, x + extra, len(d) + 64, **kw)
            assert {k: a[k] for k in keys} == {k: b[k] for k in keys}, (name, len(extra), kw)
            assert a["out"] == d and (a["error"] != 0) == (b["error"] != 0)
    name, d, x = synth.xz_cases()[0]
    third = len(x) // 3
    for bad in (x[:len(x) // 2], x[:20], x[:5], x[:-1], x[:third] + bytes([x[third] ^ 0x55]) + x[third + 1:],
                x[:-4] + bytes([x[-4] ^ 1]) + x[-3:], b"\xfd7zXY\x00" + x[6:]):
        a = hip.stream_decode(95, bad, len(d) + 70000)
        b = ref.stream_decode(95, bad, len(d) + 70000)
        assert a["rets"][-1] == b["rets"][-1] == -3 and a["close"] == b["close"] == -112, len(bad)


def test_xz_archive_through_unmodified_mz_zip(libs):
    """An archive written by the reference writer with method 95 (liblzma stream encoder, CRC64 check) is extracted
    by the unmodified mz_zip reader on the HIP codecs; mz_zip's own CRC verification must pass."""
    hip, ref = libs
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(14)
    n, size = 10, 60000
    lens = rnd.randint(0, size + 1, size=n).astype(np.int32)
    lens[:3] = (0, 1, size)
    offs = rnd.randint(0, len(c) - size, size=n).astype(np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.zip")
        ref.zip_write(path, c, offs, lens, method=95, level=6)
        t_ref, t_hip = ref.zip_index(path), hip.zip_index(path)
        assert (t_ref == t_hip).all() and (t_ref[:, 0] == 95).all()
        cd = t_ref[:, 6].copy()
        out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        o_ref = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
        o_hip = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
        _, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=2, out=o_ref, out_off=out_off)
        _, crc_h, ulen_h, st_h = hip.zip_read_all(path, cd, nthreads=2, out=o_hip, out_off=out_off)
        assert (st_r == 0).all() and (st_h == 0).all(), st_h
        assert (crc_r == crc_h).all() and (ulen_h == lens).all() and (o_ref == o_hip).all()
        assert (crc_h == t_ref[:, 2].astype(np.uint32)).all()
