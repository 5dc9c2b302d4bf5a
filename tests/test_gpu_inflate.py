"""GPU parity tests for K1+K2 (raw-DEFLATE decode + fused CRC-32) through the C ABI
(mzhip_inflate_batch / mzhip_inflate_host_a), checked against the oracle (oracle/*.c), zlib-made
expected bytes, the golden fixtures, and -- where oracle/_ref travelled -- the compiled reference."""
import zlib

import numpy as np
import pytest

import oracle
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    return gpu_util


def test_edge_payloads_batch(gpu):
    cases = synth.edge_payloads()
    batch = gpu.make_batch([z + b"\x00trailing-garbage" for _, _, z in cases], [len(d) + 8 for _, d, _ in cases])
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    h_out = batch["d_out"].cpu().numpy()
    for i, (name, data, z) in enumerate(cases):
        st, used, ref_out = oracle.inflate_raw(z + b"\x00trailing-garbage", len(data) + 8)
        assert status[i] == st == 0, name
        assert in_used[i] == used == len(z), name
        assert out_len[i] == len(data), name
        assert gpu.entry_bytes(batch, h_out, i, len(data)) == data, name
        assert crc[i] == oracle.crc32(data) == zlib.crc32(data), name


def test_long_codes_batch(gpu):
    """Dynamic codes of 13-15 bits over the full alphabet (second-level tables at capacity, long distance codes)."""
    cases = synth.long_code_payloads()
    batch = gpu.make_batch([z for _, _, z in cases], [len(d) + 8 for _, d, _ in cases])
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    h_out = batch["d_out"].cpu().numpy()
    for i, (name, data, z) in enumerate(cases):
        assert (status[i], in_used[i], out_len[i], crc[i]) == (0, len(z), len(data), zlib.crc32(data)), name
        assert gpu.entry_bytes(batch, h_out, i, len(data)) == data, name


def test_fixtures_golden(gpu, fixtures):
    ents = [e for e in fixtures if e["method"] == 8]
    batch = gpu.make_batch([e["payload"] for e in ents], [e["usize"] + 4 for e in ents], align=1)
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    for i, e in enumerate(ents):
        assert status[i] == 0 and out_len[i] == e["usize"] and in_used[i] == e["csize"], (e["archive"], e["entry"])
        assert crc[i] == e["crc"], (e["archive"], e["entry"])          # the CRC the archive's CD pins
        assert in_used[i] == e["ref"]["total_in"] and out_len[i] == e["ref"]["total_out"]


def test_status_parity_malformed(gpu):
    """Error class (and exact in_used on success) for truncated / bit-flipped streams vs the oracle."""
    names, pays, caps = [], [], []
    for name, data, z in synth.edge_payloads():
        if len(z) < 16:
            continue
        for cname, bad in synth.corruptions(z):
            names.append((name, cname))
            pays.append(bad)
            caps.append(len(data) + 70000)
    # an unused code of an incomplete distance set, the input ending right behind it: Z_DATA_ERROR on its one bit, at every cut
    # (tests/test_oracle.py INCOMPLETE_DISTANCE_SET: the two streams of the device fuzz on which the restatement was wrong)
    from tests.test_oracle import INCOMPLETE_DISTANCE_SET
    for k, z in enumerate(INCOMPLETE_DISTANCE_SET):
        for n in range(1, len(z) + 1):
            names.append(("incomplete distance set %d" % k, "cut %d" % n))
            pays.append(z[:n])
            caps.append(100000)
    batch = gpu.make_batch(pays, caps)
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    h_out = batch["d_out"].cpu().numpy()
    for i, nm in enumerate(names):
        st, used, out = oracle.inflate_raw(pays[i], caps[i])
        assert status[i] == st, (nm, status[i], st)
        if st == 0:
            assert in_used[i] == used and out_len[i] == len(out), nm
            assert gpu.entry_bytes(batch, h_out, i, len(out)) == out, nm
            assert crc[i] == oracle.crc32(out), nm


def test_out_cap_exceeded(gpu):
    data = synth.corpus()[:50000]
    z = synth.deflate_raw(data)
    batch = gpu.make_batch([z, z], [len(data) - 1, len(data)])
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    assert status[0] == -200 and status[1] == 0 and crc[1] == zlib.crc32(data)


def test_many_entries_64k_and_8k(gpu):
    """A few thousand corpus slices (configs 2 and 3 shapes), every byte and CRC compared."""
    for size, n, seed in ((65536, 1500, 1234), (8192, 4000, 1235)):
        datas = synth.slices(n, size, seed)
        pays = [synth.deflate_raw(d) for d in datas]
        batch = gpu.make_batch(pays, [size] * n)
        out_len, in_used, crc, status = gpu.run_inflate(batch)
        h_out = batch["d_out"].cpu().numpy()
        assert (status == 0).all()
        assert (out_len == size).all()
        assert (in_used == np.array([len(p) for p in pays])).all()
        want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)
        assert (crc == want).all()
        for i in range(0, n, 7):
            assert gpu.entry_bytes(batch, h_out, i, size) == datas[i]
        # oracle spot-check (the restatement is slow: a sample)
        for i in range(0, n, 211):
            st, used, out = oracle.inflate_raw(pays[i], size)
            assert st == 0 and out == datas[i] and used == in_used[i] and oracle.crc32(out) == crc[i]


def test_stream_beyond_256_mib(gpu):
    """One entry whose compressed stream is longer than a 32-bit bit cursor can address (272 MiB; the reference streams
    any size, mz_strm_zlib.c:116-193): 4350 stored blocks of noise, then Huffman blocks of text that start behind bit
    2^31, next to an ordinary entry in the same batch.  Every byte compared."""
    rnd = np.random.RandomState(3)
    noise = rnd.randint(0, 256, size=65535, dtype=np.uint8)
    nblk = 4350
    hdr = np.array([0, 0xFF, 0xFF, 0x00, 0x00], dtype=np.uint8)          # BFINAL=0 BTYPE=00, LEN=65535, NLEN=0
    blocks = np.empty((nblk, 5 + 65535), dtype=np.uint8)
    blocks[:, :5] = hdr
    for i in range(nblk):                                                # every block its own rotation of the noise
        blocks[i, 5:] = np.roll(noise, i * 7)
    text = (synth.corpus() * 30)[:6000000]
    tail = synth.deflate_raw(text)                                       # final blocks, byte aligned behind the stored ones
    z = blocks.tobytes() + tail
    assert len(z) > (1 << 28) + (1 << 20)
    want = blocks[:, 5:].tobytes() + text
    small = synth.corpus()[:40000]
    batch = gpu.make_batch([z, synth.deflate_raw(small)], [len(want), len(small)])
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    assert status.tolist() == [0, 0]
    assert out_len.tolist() == [len(want), len(small)] and in_used.tolist() == [len(z), len(synth.deflate_raw(small))]
    assert crc[0] == zlib.crc32(want) and crc[1] == zlib.crc32(small)
    h_out = batch["d_out"].cpu().numpy()
    assert gpu.entry_bytes(batch, h_out, 0, len(want)) == want
    # cut inside the text blocks, behind bit 2^31: input exhausted, like the reference
    batch = gpu.make_batch([z[:len(z) - 1000]], [len(want)])
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    assert status[0] == -5


def test_unaligned_offsets(gpu):
    datas = synth.slices(64, 5000, 99)
    pays = [synth.deflate_raw(d) for d in datas]
    batch = gpu.make_batch(pays, [5000 + (i % 5) for i in range(64)], align=1)
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    h_out = batch["d_out"].cpu().numpy()
    for i in range(64):
        assert status[i] == 0 and gpu.entry_bytes(batch, h_out, i, 5000) == datas[i] and crc[i] == zlib.crc32(datas[i])


def test_inflate_host_entry_point(gpu):
    data = synth.corpus()[7000:7000 + 65536]
    z = synth.deflate_raw(data)
    st, used, out, crc = gpu.mz.inflate_host(z + b"xx", len(data))
    assert (st, used, out, crc) == (0, len(z), data, zlib.crc32(data))
    st, used, out, crc = gpu.mz.inflate_host(z[:len(z) // 2], len(data))
    assert st == -5


def test_crc32_batch_and_host(gpu):
    import torch

    rnd = np.random.RandomState(3)
    sizes = [0, 1, 2, 15, 16, 17, 1023, 1024, 1025, 4096, 65535, 65536, 300001]
    bufs = [rnd.bytes(s) for s in sizes]
    off = np.cumsum([0] + [len(b) + 3 for b in bufs[:-1]]).astype(np.int64)
    blob = np.zeros(int(off[-1]) + sizes[-1] + 16, dtype=np.uint8)
    for o, b in zip(off, bufs):
        blob[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    d = torch.from_numpy(blob).cuda()
    init = rnd.randint(0, 2**31, size=len(sizes)).astype(np.int32)
    crc = gpu.mz.crc32_batch(d, torch.from_numpy(off).cuda(), torch.tensor(sizes, dtype=torch.int32).cuda(),
                             torch.from_numpy(init).cuda())
    torch.cuda.synchronize()
    got = gpu.mz.u32(crc)
    for i, b in enumerate(bufs):
        assert got[i] == zlib.crc32(b, int(init[i])) == oracle.crc32(b, int(init[i])), sizes[i]
    big = rnd.bytes(3 * (1 << 20) + 12345)
    assert gpu.mz.crc32_host(big) == zlib.crc32(big)
    assert gpu.mz.crc32_host(big[:1000], 0xDEADBEEF) == zlib.crc32(big[:1000], 0xDEADBEEF)
    assert gpu.mz.crc32_host(b"123456789") == 0xCBF43926


def test_differential_fuzz_batch(gpu):
    """3000 randomly corrupted / truncated streams in one launch: status class, and bytes + consumed input +
    CRC wherever the stream still decodes, against the oracle."""
    import random

    rnd = random.Random(777)
    c = synth.corpus()
    bases = [synth.deflate_raw(c[o:o + n], level=lv) for o, n, lv in ((100, 3000, 6), (5000, 20000, 9), (70000, 9000, 1))]
    bases.append(synth.deflate_raw(c[:6000], strategy=zlib.Z_FIXED))
    bases.append(synth.stored_blocks(c[:3000], block=1000))
    pays = []
    for it in range(3000):
        z = bytearray(rnd.choice(bases))
        kind = rnd.randrange(4)
        if kind == 0:
            z[rnd.randrange(len(z))] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            z[rnd.randrange(len(z))] = rnd.randrange(256)
        elif kind == 2:
            del z[rnd.randrange(1, len(z)):]
        else:
            z[rnd.randrange(min(len(z), 40))] = rnd.randrange(256)
        pays.append(bytes(z))
    cap = 120000
    batch = gpu.make_batch(pays, [cap] * len(pays), align=1)
    out_len, in_used, crc, status = gpu.run_inflate(batch)
    h_out = batch["d_out"].cpu().numpy()
    n_ok = 0
    for i, z in enumerate(pays):
        so, uo, oo = oracle.inflate_raw(z, cap)
        assert status[i] == so, (i, status[i], so)
        if so == 0:
            assert in_used[i] == uo and out_len[i] == len(oo) and crc[i] == oracle.crc32(oo), i
            assert gpu.entry_bytes(batch, h_out, i, len(oo)) == oo, i
            n_ok += 1
    assert n_ok > 100


def test_crc32_device_fault_is_loud():
    """mz_crypt_crc32_update has no error channel (mz_crypt.h:20) and used to degrade silently when the device
    failed under it.  With a failure injected into the first device checksum of a fresh process (MZHIP_FAULT_INJECT=crc): the
    value is still the right CRC-32 (the library's own host fold), mzhip_last_error() of the thread names the symbol,
    mzhip_crc_faults() counts it, stderr carries one line, and the thread's next codec-stream read fails with MZ_STREAM_ERROR
    instead of carrying on behind a device that does not answer."""
    import os
    import subprocess
    import sys

    from tests import gpu_util

    gpu_util.mz.require_gpu()
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drop = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not os.path.exists(drop):
        pytest.skip("drop-in library missing (built where /root/reference exists)")
    code = r"""
import ctypes as C, sys, zlib
sys.path.insert(0, %r)
import numpy as np
import oracle
from tests import synth
hip = oracle.MzDriver(%r)
L = C.CDLL(%r)
L.mzhip_last_error.restype = C.c_char_p
L.mzhip_crc_faults.restype = C.c_uint64
data = synth.corpus()[:200000]
assert L.mzhip_crc_faults() == 0
got = hip.crc32(data)
assert got == zlib.crc32(data), (got, zlib.crc32(data))
assert L.mzhip_crc_faults() == 1
msg = L.mzhip_last_error().decode()
assert "mz_crypt_crc32_update" in msg and "host" in msg, msg
z = synth.deflate_raw(data[:70000])
r = hip.stream_decode(8, z, 70000)
assert r["rets"][0] == -1, r["rets"]            # MZ_STREAM_ERROR: the fault the checksum call could not report
r = hip.stream_decode(8, z, 70000)              # ... reported once; the device answers again
assert r["out"] == data[:70000] and r["close"] == 0
assert hip.crc32(data) == zlib.crc32(data) and L.mzhip_crc_faults() == 1
print("fault path ok")
""" % (ROOT, drop, gpu_util.mz.LIB_PATH)
    env = dict(os.environ, MZHIP_FAULT_INJECT="crc")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "fault path ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "mzhip: mz_crypt_crc32_update: device failure" in r.stderr
