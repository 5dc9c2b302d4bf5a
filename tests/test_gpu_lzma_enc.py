"""GPU parity tests for LZMA / XZ ENCODE: mzhip_lzma_encode_batch, the host-buffer
entry points, and the drop-in mz_stream_lzma WRITE path (methods 14 and 95).  Compressor output is not a format
property, so parity = the reference side (oracle restatement, liblzma through Python, the compiled reference's
mz_stream_lzma READ and its unmodified mz_zip reader) decodes the bytes back to the input and every CRC agrees."""
import ctypes as C
import lzma
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
_u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_lzma_encode_batch.restype = C.c_int32
    L.mzhip_lzma_encode_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4
    for fn in (L.mzhip_lzma_encode_host, L.mzhip_xz_encode_host):
        fn.restype = C.c_int32
        fn.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return gpu_util


def _cases():
    c = synth.corpus()
    rnd = np.random.RandomState(3)
    return [b"", b"a", b"ab", b"a" * 20, c[:100], c[:5000], c[:65536], c[:200000], rnd.bytes(3000), bytes(100000),
            c[:70000] + rnd.bytes(500) + c[:70000], b"abcabcabc" * 1000, c[:65535], c[:65537], c + c[:100000]]


def test_lzma_encode_batch_roundtrip(gpu):
    import torch

    datas = _cases() + synth.slices(200, 65536, 1234) + synth.slices(300, 8192, 1235)
    caps = [len(d) + len(d) // 8 + 1024 for d in datas]
    b = gpu.make_batch(datas, caps)
    n = len(datas)
    dev = b["d_in"].device
    out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    rc = gpu.mz.lib().mzhip_lzma_encode_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                              max(len(d) for d in datas), b["d_out"].data_ptr(), b["out_off"].data_ptr(),
                                              b["out_cap"].data_ptr(), None, n, out_len.data_ptr(), crc.data_ptr(),
                                              status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    h = b["d_out"].cpu().numpy()
    ol, st, k = out_len.cpu().numpy(), status.cpu().numpy(), gpu.mz.u32(crc)
    assert (st == 0).all()
    tot_in = tot_out = 0
    for i, d in enumerate(datas):
        z = gpu.entry_bytes(b, h, i, int(ol[i]))
        assert k[i] == zlib.crc32(d), i
        # the header's dictionary: 64 KiB for a stream of one block, 8 MiB (the reach of the chain pass's links) beyond
        assert z[:9] == bytes([9, 20, 5, 0, 0x5D]) + (bytes([0, 0, 1, 0]) if len(d) <= 65536 else bytes([0, 0, 0x80, 0])), i
        if i < 40 or i % 23 == 0:
            assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == d, i
        if i < 20 or i % 57 == 0:
            assert oracle.lzma_zip_decode(z, len(d) + 64, -1) == (0, len(z), d), i
        if len(d) == 65536:
            tot_in += len(d)
            tot_out += len(z)
    assert tot_out < 0.40 * tot_in          # it does compress text (liblzma preset 6 reaches 0.28 on the same slices)


def test_host_entry_points(gpu):
    L = gpu.mz.lib()
    for d in _cases():
        for fn, kind in ((L.mzhip_lzma_encode_host, 14), (L.mzhip_xz_encode_host, 95)):
            cap = len(d) + len(d) // 8 + 4096
            out = np.zeros(cap, dtype=np.uint8)
            ol, crc = C.c_uint32(), C.c_uint32()
            assert fn(d, len(d), out.ctypes.data, cap, C.byref(ol), C.byref(crc)) == 0
            z = out[:ol.value].tobytes()
            assert crc.value == zlib.crc32(d)
            if kind == 14:
                assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == d
            else:
                assert lzma.decompress(z, format=lzma.FORMAT_XZ) == d
                assert oracle.xz_decode(z + b"tail", len(d) + 64) == (0, len(z), d)


def test_preset_selects_the_parse_class(gpu):
    """COMPRESS_LEVEL reaches the encoder as liblzma's preset (mz_strm_lzma.c:81): presets 0-3 = the one-candidate parse,
    4-9 and the default four candidates + lazy rule, the chain followed deeper as the preset goes up -- every stream decodes
    with liblzma, the default class is smaller."""
    L = gpu.mz.lib()
    for f in (L.mzhip_lzma_encode_host_preset, L.mzhip_xz_encode_host_preset):
        f.restype = C.c_int32
        f.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    c = synth.corpus()
    sizes = {}
    for preset in (0, 3, 4, 6, 9, -1):
        tot = 0
        for d in (c[:65536], c[100000:300000], b"", b"abc" * 5000):
            for fn, kind in ((L.mzhip_lzma_encode_host_preset, 14), (L.mzhip_xz_encode_host_preset, 95)):
                cap = len(d) + len(d) // 8 + 4096
                out = np.zeros(cap, dtype=np.uint8)
                ol, crc = C.c_uint32(), C.c_uint32()
                assert fn(d, len(d), preset, out.ctypes.data, cap, C.byref(ol), C.byref(crc)) == 0
                z = out[:ol.value].tobytes()
                assert crc.value == zlib.crc32(d)
                if kind == 14:
                    assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == d
                else:
                    assert lzma.decompress(z, format=lzma.FORMAT_XZ) == d
                tot += len(z)
        sizes[preset] = tot
    # the classes: 0-3 one candidate and one link of the chain; 4-5 four candidates + 4 links, 6 (= the default) 8 links, 7-9 16
    # links (MZ_LZE_DEPTH_FOR_PRESET) -- the chain only matters for the streams of more than one block (one case of four)
    assert sizes[0] == sizes[3] and sizes[6] == sizes[-1] and sizes[9] <= sizes[6] <= sizes[4]
    assert sizes[6] < 0.95 * sizes[3]
    # ratio bars against liblzma itself at preset 6.  One 64 KiB piece: the same history for both,
    # what differs is the parse (liblzma prices every choice; K6 = four hash candidates + inheritance + lazy rule).
    # A longer stream: K6 looks back 32 KiB across its 64 KiB blocks, liblzma over the whole stream (8 MiB dictionary):
    # held against liblzma on the same 64 KiB pieces for the parse, against liblzma on the whole stream for the record.
    def raw6(b):
        return len(lzma.compress(b, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "preset": 6}]))
    for d, bar in ((c[:65536], 1.15), (c[100000:300000], 1.12)):
        cap = len(d) + len(d) // 8 + 4096
        out = np.zeros(cap, dtype=np.uint8)
        ol, crc = C.c_uint32(), C.c_uint32()
        assert L.mzhip_lzma_encode_host_preset(d, len(d), 6, out.ctypes.data, cap, C.byref(ol), C.byref(crc)) == 0
        pieces = sum(raw6(d[o:o + 65536]) for o in range(0, len(d), 65536))
        print("LZMA preset 6, %d bytes: %d (liblzma on the same 64 KiB pieces: %d = x%.3f; on the whole stream: %d = x%.3f)"
              % (len(d), ol.value, pieces, ol.value / pieces, raw6(d), ol.value / raw6(d)))
        assert ol.value <= bar * pieces, (len(d), ol.value, pieces)


def test_long_history_ratio_on_the_config4_corpus(gpu):
    """VERDICT r3 item 8 / missing 5: mz_strm_lzma.c:81 hands preset 6 (8 MiB dictionary) to liblzma; K6's matches reach
    back 8 MiB as well since round 4 (the chain pass, lzma_enc_core.h mz_lz_chain) -- ratio <= 0.30 on the 1 MiB entries
    of BASELINE.json configs[3] (round 3: 0.42; liblzma preset 6: 0.245), every stream decoded by liblzma, the batch
    entry point and the host entry point making the same bytes."""
    import torch

    L = gpu.mz.lib()
    L.mzhip_lzma_encode_host_preset.restype = C.c_int32
    L.mzhip_lzma_encode_host_preset.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mzhip_lzma_encode_batch_preset.restype = C.c_int32
    L.mzhip_lzma_encode_batch_preset.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_int32] + [C.c_void_p] * 4
    datas = synth.markov_entries(6, 1 << 20, 77, synth.bench_corpus()[0]) + [synth.corpus()[:300000], b"q" * 70000]
    caps = [len(d) + len(d) // 8 + 1024 for d in datas]
    b = gpu.make_batch(datas, caps)
    n = len(datas)
    dev = b["d_in"].device
    out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    assert L.mzhip_lzma_encode_batch_preset(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                            max(len(d) for d in datas), b["d_out"].data_ptr(), b["out_off"].data_ptr(),
                                            b["out_cap"].data_ptr(), None, n, 6, out_len.data_ptr(), crc.data_ptr(),
                                            status.data_ptr(), None) == 0
    torch.cuda.synchronize()
    h = b["d_out"].cpu().numpy()
    ol, st, k = out_len.cpu().numpy(), status.cpu().numpy(), gpu.mz.u32(crc)
    assert (st == 0).all()
    tin = tout = t6 = 0
    for i, d in enumerate(datas):
        z = gpu.entry_bytes(b, h, i, int(ol[i]))
        assert k[i] == zlib.crc32(d), i
        assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == d, i
        if i in (0, n - 2, n - 1):  # the host entry point (one stream per call) makes the same bytes
            out = np.zeros(caps[i], dtype=np.uint8)
            o2, c2 = C.c_uint32(), C.c_uint32()
            assert L.mzhip_lzma_encode_host_preset(d, len(d), 6, out.ctypes.data, caps[i], C.byref(o2), C.byref(c2)) == 0
            assert out[:o2.value].tobytes() == z, i
        if i < 6:
            tin += len(d)
            tout += len(z)
            t6 += len(lzma.compress(d, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "preset": 6}]))
    print("config-4 corpus, preset 6: ratio %.4f (liblzma preset 6: %.4f)" % (tout / tin, t6 / tin))
    assert tout <= 0.30 * tin, (tout, tin)
    # method 95 the same way: the .xz block is parsed as one stream, its LZMA2 chunks keep the dictionary (round 3: every
    # 48 KiB chunk was a stream of its own and reset it -- 0.40 on these entries)
    L.mzhip_xz_encode_host_preset.restype = C.c_int32
    L.mzhip_xz_encode_host_preset.argtypes = [C.c_char_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    xin = xout = 0
    for d in datas[:2] + [datas[-2]]:
        cap = len(d) + len(d) // 8 + 4096
        out = np.zeros(cap, dtype=np.uint8)
        o2, c2 = C.c_uint32(), C.c_uint32()
        assert L.mzhip_xz_encode_host_preset(d, len(d), 6, out.ctypes.data, cap, C.byref(o2), C.byref(c2)) == 0
        x = out[:o2.value].tobytes()
        assert c2.value == zlib.crc32(d) and lzma.decompress(x, format=lzma.FORMAT_XZ) == d
        assert oracle.xz_decode(x + b"tail", len(d) + 64) == (0, len(x), d)
        if len(d) == 1 << 20:
            xin += len(d)
            xout += len(x)
    print("config-4 corpus, method 95 at preset 6: ratio %.4f" % (xout / xin))
    assert xout <= 0.30 * xin, (xout, xin)


@pytest.fixture(scope="module")
def libs(gpu):
    if not os.path.exists(DROP) or not oracle.have_ref():
        pytest.skip("drop-in / reference builds missing (built where /root/reference exists)")
    return oracle.MzDriver(DROP), oracle.ref()


def test_stream_write_is_read_by_the_reference(libs):
    """mz_stream_lzma WRITE on the HIP backend, READ on liblzma (and on the HIP backend)."""
    hip, ref = libs
    for d in _cases():
        for method in (14, 95):
            z, info = hip.stream_encode(method, d)
            assert (info["total_in"], info["total_out"], info["close"], info["error"], info["open"]) == (len(d), len(z), 0, 0, 0)
            kw = dict(max_in=len(z), max_out=len(d)) if method == 14 else {}
            b = ref.stream_decode(method, z, len(d) + 64, **kw)
            assert (b["out"], b["total_in"], b["close"]) == (d, len(z), 0), (len(d), method)
            a = hip.stream_decode(method, z, len(d) + 64, **kw)
            assert (a["out"], a["total_in"], a["close"]) == (d, len(z), 0), (len(d), method)


def test_archives_written_on_hip_are_extracted_by_the_reference(libs):
    """mz_zip_writer (unmodified) on the HIP codecs writes method-14 and method-95 archives; the all-reference
    reader extracts them and mz_zip's own CRC verification passes."""
    hip, ref = libs
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(23)
    with tempfile.TemporaryDirectory() as tmp:
        for method in (14, 95):
            n, size = 12, 90000
            lens = rnd.randint(0, size + 1, size=n).astype(np.int32)
            lens[:3] = (0, 1, size)
            offs = rnd.randint(0, len(c) - size, size=n).astype(np.int64)
            path = os.path.join(tmp, "w%d.zip" % method)
            hip.zip_write(path, c, offs, lens, method=method, level=6)
            t = ref.zip_index(path)
            assert (t[:, 0] == method).all() and (t[:, 4] == lens).all()
            out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
            o_ref = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
            _, crc_r, ulen_r, st_r = ref.zip_read_all(path, t[:, 6].copy(), nthreads=2, out=o_ref, out_off=out_off)
            assert (st_r == 0).all(), (method, st_r)
            assert (crc_r == t[:, 2].astype(np.uint32)).all() and (ulen_r == lens).all()
            for i in range(n):
                assert o_ref[out_off[i]:out_off[i] + lens[i]].tobytes() == c[offs[i]:offs[i] + lens[i]].tobytes()


def test_encode_decode_roundtrip_property_on_device(gpu):
    """lzma_decode(lzma_encode(x)) == x at scale, all on the device (K6 -> K3): 9216 x 64 KiB entries, lengths, consumed
    input and fused CRCs of both directions agree with each other and with zlib's CRC of the unique slices."""
    import torch

    L = gpu.mz.lib()
    L.mzhip_lzma_batch.restype = C.c_int32
    L.mzhip_lzma_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
    n_unique, n_total, size = 512, 9216, 65536
    datas = synth.slices(n_unique, size, 99)
    want_u = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)
    idx = np.random.RandomState(10).randint(0, n_unique, size=n_total)
    dev = torch.device("cuda:0")
    blob = torch.from_numpy(np.frombuffer(b"".join(datas), dtype=np.uint8).copy()).to(dev)
    in_off = torch.from_numpy(idx.astype(np.int64) * size).to(dev)
    in_len = torch.full((n_total,), size, dtype=torch.int32, device=dev)
    cap = size + size // 8 + 1024
    z = torch.empty(n_total * cap, dtype=torch.uint8, device=dev)
    z_off = torch.arange(n_total, dtype=torch.int64, device=dev) * cap
    z_cap = torch.full((n_total,), cap, dtype=torch.int32, device=dev)
    z_len, e_crc, e_st = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(3))
    assert L.mzhip_lzma_encode_batch(blob.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), size, z.data_ptr(),
                                     z_off.data_ptr(), z_cap.data_ptr(), None, n_total, z_len.data_ptr(), e_crc.data_ptr(),
                                     e_st.data_ptr(), None) == 0
    out = torch.empty(n_total * size, dtype=torch.uint8, device=dev)
    o_off = torch.arange(n_total, dtype=torch.int64, device=dev) * size
    o_len, used, d_crc, d_st = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(4))
    mo = torch.full((n_total,), size, dtype=torch.int64, device=dev)
    assert L.mzhip_lzma_batch(z.data_ptr(), z_off.data_ptr(), z_len.data_ptr(), out.data_ptr(), o_off.data_ptr(),
                              in_len.data_ptr(), mo.data_ptr(), n_total, o_len.data_ptr(), used.data_ptr(),
                              d_crc.data_ptr(), d_st.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int((e_st != 0).sum()) == 0 and int((d_st != 0).sum()) == 0
    assert bool((o_len == size).all()) and bool((used == z_len).all()) and bool((e_crc == d_crc).all())
    assert bool((d_crc == torch.from_numpy(want_u[idx].view(np.int32).copy()).to(dev)).all())
    for e in (0, n_total // 2, n_total - 1):
        assert out[e * size:(e + 1) * size].cpu().numpy().tobytes() == datas[idx[e]]


def test_xz_write_in_blocks_is_read_by_the_reference(gpu):
    """Method 95 WRITE in bounded memory (shim_lzma.c): an entry larger than one segment becomes one .xz stream of several
    blocks (a block per segment, index and footer at close()); liblzma behind the all-reference mz_stream_lzma READ decodes
    it, and so does this backend's own .xz kernel."""
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing")
    hip, ref = oracle.MzDriver(DROP), oracle.ref()
    L = hip.L
    L.mzhip_set_write_segment.argtypes = [C.c_int64]
    L.mzhip_set_write_segment.restype = None
    text, _ = synth.bench_corpus()
    d = (text[:450000] + bytes(100000) + text[:300000]) * 2 + bytes(range(256)) * 100
    try:
        L.mzhip_set_write_segment(128 << 10)                          # segments of 128 KiB: 14 blocks
        for lvl, chunk in ((1, 65535), (6, 1000), (6, 400000)):
            z, info = hip.stream_encode(95, d, level=lvl, chunk=chunk)
            assert info["close"] == 0 and info["total_in"] == len(d) and info["total_out"] == len(z)
            assert lzma.decompress(z, format=lzma.FORMAT_XZ) == d     # liblzma itself (Python's module)
            b = ref.stream_decode(95, z, len(d) + 64)
            assert b["out"] == d and b["close"] == 0 and b["error"] == 0
            a = hip.stream_decode(95, z, len(d) + 64)                 # ... and the .xz kernel of this backend
            assert a["out"] == d and a["close"] == 0 and a["error"] == 0
    finally:
        L.mzhip_set_write_segment(0)
