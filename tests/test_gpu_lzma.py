"""GPU parity tests for K3 (raw-LZMA1 range decode + fused CRC-32) through the C ABI
(mzhip_lzma_batch / mzhip_lzma_host) against the oracle restatement, the lzma.zip golden fixture
and -- where oracle/_ref travelled -- the compiled reference (mz_stream_lzma over liblzma 5.2.5)."""
import ctypes as C
import zlib

import numpy as np
import pytest

import oracle
from tests import synth
from tests.test_oracle import _zip_lzma

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_lzma_batch.restype = C.c_int32
    L.mzhip_lzma_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
    return gpu_util


def run_lzma(gpu, pays, caps, max_out):
    import torch

    b = gpu.make_batch(pays, caps)
    n = len(pays)
    dev = b["d_in"].device
    out_len, in_used, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    mo = torch.tensor(max_out, dtype=torch.int64, device=dev)
    rc = gpu.mz.lib().mzhip_lzma_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                       b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                       mo.data_ptr(), n, out_len.data_ptr(), in_used.data_ptr(), crc.data_ptr(),
                                       status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    return (b, b["d_out"].cpu().numpy(), out_len.cpu().numpy(), in_used.cpu().numpy(), gpu.mz.u32(crc),
            status.cpu().numpy())


def test_lzma_batch_vs_oracle(gpu):
    c = synth.corpus()
    rnd = np.random.RandomState(11)
    datas = [b"", b"a", c[:1000], c[:150000], rnd.bytes(5000), b"A" * 100000, c[1000:70000] + rnd.bytes(3000) + c[:50000]]
    datas += synth.slices(40, 20000, 77)
    pays = [_zip_lzma(d) for d in datas]
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu, pays, [len(d) + 16 for d in datas], [len(d) for d in datas])
    for i, d in enumerate(datas):
        so, uo, oo = oracle.lzma_zip_decode(pays[i], len(d) + 16, len(d))
        assert status[i] == so == 0 and in_used[i] == uo == len(pays[i]) and out_len[i] == len(d), i
        assert gpu.entry_bytes(b, h_out, i, len(d)) == d == oo, i
        assert crc[i] == oracle.crc32(d) == zlib.crc32(d), i


def test_lzma_malformed_and_clamp(gpu):
    c = synth.corpus()
    z = _zip_lzma(c[:30000])
    third = len(z) // 3
    bads = [z[:len(z) // 2], z[:20], z[:9], z[:5], z[:third] + bytes([z[third] ^ 0x55]) + z[third + 1:],
            z[:9] + b"\x01" + z[10:], z]
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu, bads, [100000] * len(bads), [-1] * (len(bads) - 1) + [12345])
    for i, bad in enumerate(bads[:-1]):
        so, uo, oo = oracle.lzma_zip_decode(bad, 100000, -1)
        assert so == -3 and status[i] in (-3, -5), (i, status[i])   # both surface as MZ_DATA_ERROR (mz_strm_lzma.c:236)
    # TOTAL_OUT_MAX clamp (mz_strm_lzma.c:214-215): length and CRC cover the clamped prefix
    assert status[-1] == 0 and out_len[-1] == 12345 and crc[-1] == zlib.crc32(c[:12345])


def test_lzma_fixture_and_reference(gpu, fixtures):
    ents = [e for e in fixtures if e["method"] == 14]
    assert ents
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu, [e["payload"] for e in ents], [e["usize"] + 8 for e in ents],
                                                        [e["usize"] for e in ents])
    for i, e in enumerate(ents):
        assert (status[i], out_len[i], in_used[i], crc[i]) == (0, e["usize"], e["csize"], e["crc"])
        assert in_used[i] == e["ref"]["total_in"] and out_len[i] == e["ref"]["total_out"]
    if oracle.have_ref():
        ref = oracle.ref()
        datas = synth.slices(6, 50000, 5)
        for d in datas:
            z, info = ref.stream_encode(14, d)          # written by the reference's own mz_stream_lzma WRITE
            st, used, out, k = gpu_lzma_host(gpu, z, len(d) + 8, len(d))
            r = ref.stream_decode(14, z, len(d) + 8, max_in=len(z), max_out=len(d))
            assert (st, used, out) == (0, r["total_in"], r["out"]) and k == ref.crc32(d)


def gpu_lzma_host(gpu, z, cap, max_out):
    L = gpu.mz.lib()
    L.mzhip_lzma_host.restype = C.c_int32
    L.mzhip_lzma_host.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64] + [C.POINTER(C.c_uint32)] * 3
    out = C.create_string_buffer(cap + 8)
    ol, iu, crc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = L.mzhip_lzma_host(z, len(z), out, cap, max_out, C.byref(ol), C.byref(iu), C.byref(crc))
    return st, iu.value, out.raw[:ol.value], crc.value


def test_lzma_every_lc_lp_pb(gpu):
    """lc / lp / pb over the whole space lzma_alone_decoder accepts (mz_strm_lzma.c:126), lc + lp = 4 included: the
    literal model's upper half then lives in a per-wave scratch in HBM (12 KiB) instead of the LDS slice."""
    import lzma as pylzma

    c = synth.corpus()
    d = c[:90000] + bytes(range(256)) * 30
    combos = [(4, 0, 2), (3, 1, 2), (0, 4, 0), (2, 2, 4), (1, 3, 1), (0, 0, 0), (3, 0, 4), (3, 0, 2)]
    pays = []
    for lc, lp, pb in combos:
        raw = pylzma.compress(d, format=pylzma.FORMAT_ALONE, filters=[dict(id=pylzma.FILTER_LZMA1, preset=6, lc=lc, lp=lp, pb=pb)])
        pays.append(bytes([5, 2, 5, 0]) + raw[:5] + raw[13:])
    pays = pays * 40                      # more entries than one wave: the scratch is per resident wave
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu, pays, [len(d) + 8] * len(pays), [len(d)] * len(pays))
    for i, z in enumerate(pays):
        assert (status[i], out_len[i], in_used[i], crc[i]) == (0, len(d), len(z), zlib.crc32(d)), (i, combos[i % len(combos)])
    assert gpu.entry_bytes(b, h_out, 0, len(d)) == d and gpu.entry_bytes(b, h_out, len(combos) + 1, len(d)) == d
    if oracle.have_ref():
        r = oracle.ref().stream_decode(14, pays[0], len(d) + 8, max_in=len(pays[0]), max_out=len(d))
        assert r["out"] == d and r["error"] == 0


def test_lzma_every_properties_byte(gpu):
    """All 256 values of the properties byte in front of two streams (a run of one byte: decodes to the same bytes under any
    lc / lp; text): refused with the header where liblzma refuses it (>= 225, lc + lp > 4: lzma_lzma_lclppb_decode), decoded
    or refused like the oracle everywhere else -- the oracle itself is pinned to the compiled reference on the same 256
    headers (tests/test_oracle.py::test_lzma_properties_byte_rules)."""
    import lzma as pylzma

    pays, datas = [], []
    for d in (b"q" * 5000, synth.corpus()[:20000]):
        raw = pylzma.compress(d, format=pylzma.FORMAT_ALONE, filters=[dict(id=pylzma.FILTER_LZMA1, preset=6)])
        z = bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]
        for props in range(256):
            pays.append(z[:4] + bytes([props]) + z[5:])
            datas.append(d)
    caps = [len(d) + 70000 for d in datas]
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu, pays, caps, [-1] * len(pays))
    refused = 0
    for i, z in enumerate(pays):
        props = z[4]
        st, used, out = oracle.lzma_zip_decode(z, caps[i], -1)
        if props >= 225 or props % 9 + (props // 9) % 5 > 4:
            assert st == -3 and status[i] != 0, (i, props, st, status[i])
            refused += 1
            continue
        assert (status[i] == 0) == (st == 0), (i, props, status[i], st)
        if st == 0:
            assert out_len[i] == len(out) and gpu.entry_bytes(b, h_out, i, len(out)) == out and crc[i] == zlib.crc32(out), (i, props)
    assert refused > 200
