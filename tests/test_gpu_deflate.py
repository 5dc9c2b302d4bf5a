"""GPU parity tests for K4 (DEFLATE encode, fixed-Huffman blocks, fused CRC-32 of the input) through the C ABI
(mzhip_deflate_batch / mzhip_deflate_host_a) and through the drop-in mz_stream_zlib WRITE path.  Compressor
output is not a format property, so parity = the reference side (oracle restatement, zlib 1.2.11, the compiled
reference's mz_stream_zlib READ) inflates the bytes back to the input and every CRC agrees."""
import ctypes as C
import importlib
import os
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
    L.mzhip_deflate_batch.restype = C.c_int32
    L.mzhip_deflate_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4
    return gpu_util


def run_deflate(gpu, datas, final=None):
    import torch

    caps = [len(d) + len(d) // 8 + 64 for d in datas]
    b = gpu.make_batch(datas, caps)
    n = len(datas)
    dev = b["d_in"].device
    out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    fin = torch.tensor(final, dtype=torch.uint8, device=dev) if final is not None else None
    rc = gpu.mz.lib().mzhip_deflate_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                          b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                          fin.data_ptr() if fin is not None else None, n, out_len.data_ptr(),
                                          crc.data_ptr(), status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    h = b["d_out"].cpu().numpy()
    ol = out_len.cpu().numpy()
    return [gpu.entry_bytes(b, h, i, int(ol[i])) for i in range(n)], gpu.mz.u32(crc), status.cpu().numpy()


def test_deflate_batch_roundtrip(gpu):
    c = synth.corpus()
    rnd = np.random.RandomState(5)
    datas = [b"", b"a", b"abc", b"abcd", b"aaaa", b"A" * 1000, b"ab" * 500, c[:100], c[:65536], rnd.bytes(5000),
             c[:200000], bytes(70000), b"x" * 63, b"x" * 64, b"x" * 65, c[:70000] + c[:70000]]
    datas += synth.slices(300, 65536, 1234) + synth.slices(500, 8192, 1235)
    zs, crc, status = run_deflate(gpu, datas)
    assert (status == 0).all()
    tot_in = tot_out = 0
    for i, d in enumerate(datas):
        assert zlib.decompress(zs[i], -15) == d, i
        assert crc[i] == zlib.crc32(d), i
        tot_in += len(d)
        tot_out += len(zs[i])
    assert tot_out < 0.55 * tot_in
    for i in range(0, len(datas), 37):                       # the slow restatement on a sample
        so, used, out = oracle.inflate_raw(zs[i], len(datas[i]) + 8)
        assert (so, used, out) == (0, len(zs[i]), datas[i]) and oracle.crc32(out) == crc[i]
    # and back through the device decoder (K4 -> K1+K2)
    b = gpu.make_batch(zs, [len(d) + 8 for d in datas])
    out_len, in_used, crc2, st2 = gpu.run_inflate(b)
    assert (st2 == 0).all() and (crc2 == crc).all() and (out_len == np.array([len(d) for d in datas])).all()


def test_deflate_nonfinal_pieces_concatenate(gpu):
    c = synth.corpus()
    parts = [c[:30000], c[30000:90000], b"", c[90000:100000]]
    zs, crc, status = run_deflate(gpu, parts, final=[0, 0, 0, 1])
    assert (status == 0).all()
    assert zlib.decompress(b"".join(zs), -15) == b"".join(parts)


def test_zlib_stream_write_dropin(gpu):
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing (built where /root/reference exists)")
    hip, ref = oracle.MzDriver(DROP), oracle.ref()
    c = synth.corpus()
    rnd = np.random.RandomState(9)
    for d in (c[:877], b"", b"test data", c[:150000], rnd.bytes(70000), c * 20):   # c*20 = 9.2 MB: > one segment
        z, info = hip.stream_encode(8, d, level=1, chunk=65535)
        # same contract the reference's own round-trip test checks (test_stream_compress.cc:78-82)
        assert info["open"] == 0 and info["close"] == 0 and info["error"] == 0
        assert info["total_in"] == len(d) and info["total_out"] == len(z)
        r = ref.stream_decode(8, z, len(d) + 64, chunk=16384)     # decoded by the REFERENCE (zlib 1.2.11)
        assert r["out"] == d and r["error"] == 0 and r["total_in"] == len(z)
        h = hip.stream_decode(8, z, len(d) + 64, chunk=16384)     # and by the HIP stream
        assert h["out"] == d and h["rets"] == r["rets"]


def test_archives_written_through_unmodified_mz_zip(gpu):
    """mz_zip_writer (reference, unmodified) on top of the HIP zlib stream writes an archive that the
    all-reference reader extracts with CRC verification (mz_zip.c:2116-2128)."""
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing (built where /root/reference exists)")
    hip, ref = oracle.MzDriver(DROP), oracle.ref()
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(4)
    n, size = 120, 65536
    lens = rnd.randint(0, size + 1, size=n).astype(np.int32)
    lens[:3] = (0, 1, size)
    offs = rnd.randint(0, len(c) - size, size=n).astype(np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "hip_written.zip")
        hip.zip_write(path, c, offs, lens, method=8, level=1)
        t = ref.zip_index(path)
        assert len(t) == n and (t[:, 0] == 8).all() and (t[:, 4] == lens).all()
        out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        o = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
        _, crc, ulen, st = ref.zip_read_all(path, t[:, 6].copy(), nthreads=2, out=o, out_off=out_off)
        assert (st == 0).all() and (ulen == lens).all()
        for i in range(n):
            assert o[out_off[i]:out_off[i] + lens[i]].tobytes() == c[offs[i]:offs[i] + lens[i]].tobytes()
            assert crc[i] == zlib.crc32(c[offs[i]:offs[i] + lens[i]].tobytes()) == t[i, 2]


def test_full_size_encode_decode_roundtrip_property(gpu):
    """Size-independent property at config-5 / config-2 scale, all on the device: for 100 000 x 64 KiB entries
    inflate(deflate(x)) has the length and the CRC-32 of x, the CRC the encoder fused over its input equals the CRC the
    decoder fused over its output, and both equal zlib's (checked on the unique slices the entries are tiled from)."""
    import torch

    n_unique, n_total, size = 1024, 100000, 65536
    datas = synth.slices(n_unique, size, 4321)
    want_u = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)
    idx = np.random.RandomState(9).randint(0, n_unique, size=n_total)
    dev = torch.device("cuda:0")
    blob = torch.from_numpy(np.frombuffer(b"".join(datas), dtype=np.uint8).copy()).to(dev)
    in_off = torch.from_numpy(idx.astype(np.int64) * size).to(dev)              # entries share the unique inputs
    in_len = torch.full((n_total,), size, dtype=torch.int32, device=dev)
    cap = size + size // 8 + 64
    z = torch.empty(n_total * cap, dtype=torch.uint8, device=dev)
    z_off = torch.arange(n_total, dtype=torch.int64, device=dev) * cap
    z_cap = torch.full((n_total,), cap, dtype=torch.int32, device=dev)
    z_len, e_crc, e_st = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(3))
    L = gpu.mz.lib()
    assert L.mzhip_deflate_batch(blob.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), z.data_ptr(), z_off.data_ptr(),
                                 z_cap.data_ptr(), None, n_total, z_len.data_ptr(), e_crc.data_ptr(), e_st.data_ptr(),
                                 None) == 0
    out = torch.empty(n_total * size, dtype=torch.uint8, device=dev)
    o_off = torch.arange(n_total, dtype=torch.int64, device=dev) * size
    o_len, used, d_crc, d_st = gpu.mz.inflate_batch(z, z_off, z_len, out, o_off, in_len)
    torch.cuda.synchronize()
    assert int((e_st != 0).sum()) == 0 and int((d_st != 0).sum()) == 0
    assert bool((o_len == size).all()) and bool((used == z_len).all())
    assert bool((e_crc == d_crc).all())
    want = torch.from_numpy(want_u[idx].view(np.int32).copy()).to(dev)
    assert bool((d_crc == want).all())
    assert float(z_len.sum()) / (n_total * size) < 0.40
    for e in (0, n_total // 2, n_total - 1):          # and bytes, on a few entries
        assert out[e * size:(e + 1) * size].cpu().numpy().tobytes() == datas[idx[e]]


@pytest.mark.parametrize("level", [1, 4, 6, 7, 9, -1])
def test_compress_level_is_honoured(gpu, level):
    """COMPRESS_LEVEL reaches the encoder (the reference hands it to deflateInit2, mz_strm_zlib.c:87,339-343): levels 1-3
    take the fast class, 4-6 and -1 four candidates per hash bucket + the lazy rule, 7-9 the cost parse on top.  Same
    round-trip bar at every level -- the REFERENCE's inflate returns the input -- plus a ratio bar per class against
    zlib 1.2.11 at the same level on the same 64 KiB slices of the bench corpus (levels 7-9 <= 0.310)."""
    import torch

    L = gpu.mz.lib()
    L.mzhip_deflate_batch_level.restype = C.c_int32
    L.mzhip_deflate_batch_level.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_int32, C.c_int32] + [C.c_void_p] * 4
    text, _ = synth.bench_corpus()
    rnd = np.random.RandomState(77)
    datas = [text[o:o + 65536] for o in rnd.randint(0, len(text) - 65536, size=400)]
    datas += [b"", b"abc", b"A" * 70000, text[:100], text[:200000], rnd.bytes(3000)]
    caps = [len(d) + len(d) // 8 + 64 for d in datas]
    b = gpu.make_batch(datas, caps)
    n = len(datas)
    dev = b["d_in"].device
    out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    rc = L.mzhip_deflate_batch_level(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(), b["d_out"].data_ptr(),
                                     b["out_off"].data_ptr(), b["out_cap"].data_ptr(), None, n, level, 15, out_len.data_ptr(),
                                     crc.data_ptr(), status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    h = b["d_out"].cpu().numpy()
    ol = out_len.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    zs = [gpu.entry_bytes(b, h, i, int(ol[i])) for i in range(n)]
    ref = oracle.ref() if oracle.have_ref() else None
    for i, d in enumerate(datas):
        assert zlib.decompress(zs[i], -15) == d and gpu.mz.u32(crc)[i] == zlib.crc32(d), (level, i)
        if ref is not None and i % 25 == 0:
            r = ref.stream_decode(8, zs[i], len(d) + 64, chunk=16384)          # mz_stream_zlib_read of the reference
            assert r["out"] == d and r["error"] == 0 and r["total_in"] == len(zs[i]), (level, i)
    ratio = sum(len(z) for z in zs[:400]) / (400 * 65536)
    zl = sum(len(zlib.compress(d, level)) - 6 for d in datas[:400]) / (400 * 65536)
    print("level %d: ratio %.4f (zlib at the same level: %.4f)" % (level, ratio, zl))
    if 1 <= level <= 3:
        assert ratio <= zl, (level, ratio, zl)                     # the fast class beats zlib-1
    elif level >= 7:
        assert ratio <= 0.310 and ratio <= 1.04 * zl, (level, ratio, zl)
    else:
        assert ratio <= 0.320 and ratio <= 1.07 * zl, (level, ratio, zl)


def test_stream_level_property_changes_the_output():
    """MZ_STREAM_PROP_COMPRESS_LEVEL through the drop-in stream: level 9 output is smaller than level 1 output, both
    decode on the reference."""
    mz = importlib.import_module("minizip-ng_amd")
    mz.require_gpu()
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing (built where /root/reference exists)")
    hip, ref = oracle.MzDriver(DROP), oracle.ref()
    text, _ = synth.bench_corpus()
    d = text[:300000]
    z1, i1 = hip.stream_encode(8, d, level=1, chunk=65535)
    z9, i9 = hip.stream_encode(8, d, level=9, chunk=65535)
    assert i1["error"] == 0 and i9["error"] == 0 and len(z9) < 0.95 * len(z1)
    for z in (z1, z9):
        r = ref.stream_decode(8, z, len(d) + 64, chunk=16384)
        assert r["out"] == d and r["error"] == 0 and r["total_in"] == len(z)
