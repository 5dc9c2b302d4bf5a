"""Concurrent use of the batch ABI: several host threads, each on its own HIP stream, running the two encoders that
need per-launch device scratch (K4 tokens, K6 tokens) back to back without synchronising in between.  The scratch
cache (scratch_acquire in mzhip_kernels.hip) must never hand two in-flight launches on different streams the same
buffer; every result is checked by inflating / LZMA-decoding it on the CPU."""
import ctypes as C
import lzma
import threading
import zlib

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_deflate_batch.restype = C.c_int32
    L.mzhip_deflate_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4
    L.mzhip_lzma_encode_batch.restype = C.c_int32
    L.mzhip_lzma_encode_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4
    return gpu_util


def test_encoders_on_concurrent_streams(gpu):
    import torch

    L = gpu.mz.lib()
    n_threads, rounds = 4, 5
    errors, work = [], {}

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            jobs = []
            with torch.cuda.stream(stream):
                for r in range(rounds):
                    # sizes differ per thread and round so the cache sees growing and shrinking requests
                    n = 40 + 37 * ((t + r) % 4)
                    size = (8192, 65536, 200000, 30000)[(t * 3 + r) % 4]
                    datas = synth.slices(n, size, 9000 + 100 * t + r)
                    caps = [len(d) + len(d) // 8 + 1024 for d in datas]
                    for kind in ("deflate", "lzma"):
                        b = gpu.make_batch(datas, caps)
                        dev = b["d_in"].device
                        out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
                        if kind == "deflate":
                            rc = L.mzhip_deflate_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                                       b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                                       None, n, out_len.data_ptr(), crc.data_ptr(), status.data_ptr(),
                                                       stream.cuda_stream)
                        else:
                            rc = L.mzhip_lzma_encode_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                                           size, b["d_out"].data_ptr(), b["out_off"].data_ptr(),
                                                           b["out_cap"].data_ptr(), None, n, out_len.data_ptr(), crc.data_ptr(),
                                                           status.data_ptr(), stream.cuda_stream)
                        assert rc == 0, (t, r, kind, rc)
                        jobs.append((kind, datas, b, out_len, crc, status))      # no synchronisation between launches
            stream.synchronize()
            work[t] = jobs
        except Exception as e:                                                   # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    checked = 0
    for t, jobs in work.items():
        for kind, datas, b, out_len, crc, status in jobs:
            h = b["d_out"].cpu().numpy()
            ol, st, k = out_len.cpu().numpy(), status.cpu().numpy(), gpu.mz.u32(crc)
            assert (st == 0).all(), (t, kind)
            for i in range(0, len(datas), 7):
                z = gpu.entry_bytes(b, h, i, int(ol[i]))
                assert k[i] == zlib.crc32(datas[i]), (t, kind, i)
                if kind == "deflate":
                    assert zlib.decompress(z, -15) == datas[i], (t, kind, i)
                else:
                    assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == datas[i], (t, kind, i)
                checked += 1
    assert checked > 300


def test_one_window_on_many_waves(gpu):
    """mzhip_inflate_parallel_host itself: one 64 MiB window of a level-6 stream of text (~0.3 compressed) handed over whole
    -- the header search finds its blocks, every block is parsed by a wave of its own, the source map is resolved -- against
    zlib's bytes; the state handed back is the header the serial kernel would go on from."""
    import ctypes as C
    import time
    import zlib

    import numpy as np

    L = gpu.mz.lib()
    text, _ = synth.bench_corpus()
    d = (text + text[::-1][:100000]) * 100                                  # ~57 MB
    z = synth.deflate_raw(d, 6)
    zin = np.frombuffer(z, dtype=np.uint8).copy()
    cap = 64 << 20
    buf = np.zeros(cap, dtype=np.uint8)
    st = (C.c_uint32 * 4)()
    ol, nb, ended, nseg = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    seg = (C.c_uint32 * 2048)()
    L.mzhip_inflate_parallel_host.restype = C.c_int32
    L.mzhip_inflate_parallel_host.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    wcrc, wadl = C.c_uint32(), C.c_uint32()   # CRC-32 / Adler-32 of the window's bytes (what the gzip / zlib trailers run over)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        rc = L.mzhip_inflate_parallel_host(zin.ctypes.data, zin.size, buf.ctypes.data, cap, None, C.byref(st), C.byref(ol), C.byref(nb),
                                           C.byref(ended), C.byref(wcrc), C.byref(wadl), 0, 65535, seg, 2048, C.byref(nseg))
        best = min(best, time.time() - t0)
        assert rc == 0, rc
    print("one window: %d blocks, %d bytes, ended %d, %.1f ms host to host (%.2f GB/s)" % (nb.value, ol.value, ended.value, best * 1e3, ol.value / best / 1e9))
    assert nb.value >= 100 and ol.value <= len(d)
    assert buf[:ol.value].tobytes() == d[:ol.value]
    assert wcrc.value == zlib.crc32(d[:ol.value]) and wadl.value == zlib.adler32(d[:ol.value])
    if ended.value:
        assert ol.value == len(d) and (st[1] + 7) // 8 == len(z)
    else:                                                                   # the chain stopped (the last bits of the stream are the serial kernel's)
        assert st[0] == st[1] and st[2] == ol.value and ol.value > len(d) - (2 << 20)
    n = nseg.value
    assert n == (ol.value + 65534) // 65535
    for i in (0, 1, n - 1):
        assert seg[i] == zlib.crc32(d[i * 65535:min((i + 1) * 65535, ol.value)])


_HEADER_CHECK_PROG = r"""
import ctypes as C, json, random, sys, zlib
import numpy as np
sys.path.insert(0, %r)
from tests import synth, gpu_util
L = gpu_util.mz.lib()
L.mzhip_inflate_parallel_host.restype = C.c_int32
L.mzhip_inflate_parallel_host.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
rnd = random.Random(17)
text, _ = synth.bench_corpus()
binary = bytes((i * 7 + (i >> 3)) & 255 for i in range(200000))
d = text * 6 + binary * 3 + text[::-1] * 2 + bytes(rnd.getrandbits(3) for _ in range(300000))
cap = 32 << 20
buf = np.zeros(cap, dtype=np.uint8)
res = []
for level, strat, mem in [(l, zlib.Z_DEFAULT_STRATEGY, 8) for l in range(1, 10)] + [(6, zlib.Z_FILTERED, 8), (6, zlib.Z_HUFFMAN_ONLY, 8), (6, zlib.Z_RLE, 8),
                                                                              (1, zlib.Z_RLE, 8), (9, zlib.Z_FILTERED, 1), (6, zlib.Z_DEFAULT_STRATEGY, 1),
                                                                              (6, zlib.Z_HUFFMAN_ONLY, 1), (6, zlib.Z_HUFFMAN_ONLY, 4)]:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strat)
    z = co.compress(d) + co.flush()
    zin = np.frombuffer(z, dtype=np.uint8).copy()
    st = (C.c_uint32 * 4)()
    ol, nb, ended = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = L.mzhip_inflate_parallel_host(zin.ctypes.data, zin.size, buf.ctypes.data, cap, None, C.byref(st), C.byref(ol), C.byref(nb), C.byref(ended),
                                       None, None, 0, 0, None, 0, None)
    assert rc == 0, (level, strat, mem, rc)
    assert buf[:ol.value].tobytes() == d[:ol.value], (level, strat, mem)
    res.append([level, strat, mem, ol.value, nb.value, ended.value, len(d)])
print(json.dumps(res))
"""


def test_header_check_keeps_every_real_block(gpu):
    """Behind the header search a lane reads each candidate's header to its end and drops what a decoder would refuse
    (k_check_headers).  A real block that it dropped would end the chain of a window early: streams of every level and
    strategy zlib has -- Huffman only (no distance code at all), RLE (one distance code), filtered, a tiny memLevel (a block
    every few KB, many of them fixed: the chain ends there by design) -- are decoded by ONE many-wave window exactly as far,
    in exactly as many blocks, as with the check switched off (MZHIP_HEADER_CHECK=0, a process of its own), byte for byte; and
    the streams made of dynamic blocks throughout are decoded to their last megabyte."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for check in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _HEADER_CHECK_PROG % root], capture_output=True, text=True, timeout=600, cwd=root,
                           env=dict(os.environ, MZHIP_HEADER_CHECK=check))
        assert r.returncode == 0, r.stderr[-3000:]
        got[check] = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    assert got["1"] == got["0"]
    through = [g for g in got["1"] if g[3] > g[6] - (1 << 20)]
    print("streams decoded through by one window: %d of %d; blocks per stream %s" % (len(through), len(got["1"]), sorted(g[4] for g in got["1"])))
    assert len(through) >= 13 and all(g[4] >= 20 for g in through)


def test_header_search_kernels_agree(gpu):
    """The header search of a window (k_find_blocks: 32 bit offsets per lane from one 16-byte fetch, the loop-free tests made for
    all 32 at once) against the kernel it replaced (one offset per lane through mz_block_header_plausible, the statement of the
    test): the same set of candidates on noise, on streams of text at three levels, on stored blocks and zeros, for ranges that
    start and end anywhere, inputs of every small length and every placement relative to a dword on the device -- and, in
    Python, the dynamic-header rule itself on a sample of the candidates and of the offsets that were refused."""
    import random

    L = gpu.mz.lib()
    L.mzhip_find_blocks_host.restype = C.c_int32
    L.mzhip_find_blocks_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    rnd = random.Random(5)
    text, _ = synth.bench_corpus()

    def find(a, b0, b1, which, mis):
        cap = 1 << 20
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint32()
        rc = L.mzhip_find_blocks_host(a.ctypes.data, a.size, b0, b1, which, mis, out.ctypes.data, cap, C.byref(n))
        assert rc == 0 and n.value <= cap, (rc, n.value)
        return np.sort(out[:n.value])

    def bits(a, p, k):
        v = 0
        for i in range(k):
            v |= ((int(a[(p + i) >> 3]) >> ((p + i) & 7)) & 1) << i
        return v

    def plausible(a, p):
        if (p >> 3) + 24 > a.size:
            return False
        bt = bits(a, p + 1, 2)
        if bt == 0:
            b = (p + 10) >> 3
            return (int(a[b]) | int(a[b + 1]) << 8) ^ (int(a[b + 2]) | int(a[b + 3]) << 8) == 0xFFFF
        if bt != 2 or bits(a, p + 3, 5) > 29 or bits(a, p + 8, 5) > 29:
            return False
        kraft = 0
        for i in range(bits(a, p + 13, 4) + 4):
            l = bits(a, p + 17 + 3 * i, 3)
            kraft += (128 >> l) if l else 0
        return kraft == 128

    stored = b"".join(zlib.compressobj(0, zlib.DEFLATED, -15).compress(bytes(rnd.getrandbits(8) for _ in range(n))) for n in (70000, 3, 65535, 100))
    inputs = [("noise", bytes(rnd.getrandbits(8) for _ in range(1 << 20))),
              ("text level 1", synth.deflate_raw(text * 6, 1)), ("text level 6", synth.deflate_raw(text * 6, 6)), ("text level 9", synth.deflate_raw(text * 3, 9)),
              ("stored", stored + synth.deflate_raw(bytes(rnd.getrandbits(8) for _ in range(200000)), 6)), ("zeros", bytes(300000)), ("ones", b"\xff" * 300000)]
    total = 0
    for name, raw in inputs:
        a = np.frombuffer(raw, dtype=np.uint8).copy()
        nb = 8 * a.size
        ranges = [(0, nb), (1, nb), (31, nb - 7), (rnd.randrange(nb // 2), rnd.randrange(nb // 2, nb)), (nb - 300, nb), (12345, 12345), (12345, 12346)]
        for i, (b0, b1) in enumerate(ranges):
            mis = (0, 1, 2, 3, 5, 8, 15)[i % 7]
            new, old = find(a, b0, b1, 0, mis), find(a, b0, b1, 1, mis)
            assert np.array_equal(new, old), (name, b0, b1, mis, new.size, old.size)
            assert new.size == 0 or (new[0] >= b0 and new[-1] < b1)
            total += new.size
            kept = find(a, b0, b1, 2, mis)   # ... and what the header check behind the search keeps of them
            assert np.isin(kept, new).all() and np.unique(kept).size == kept.size, (name, b0, b1, mis)
            if name == "noise" and i == 0:
                assert kept.size * 20 < new.size + 20, (kept.size, new.size)
        full = find(a, 0, nb, 0, 0)
        yes = set(int(x) for x in full)
        for p in [int(x) for x in full[:: max(1, full.size // 60)]]:
            assert plausible(a, p), (name, p)
        for p in (rnd.randrange(nb) for _ in range(300)):
            assert plausible(a, p) == (p in yes), (name, p)
    assert total > 10000
    for n in list(range(0, 40)) + [63, 64, 65, 100, 257]:   # inputs shorter than a lane's fetch, ranges that hold nothing
        a = np.frombuffer(bytes(rnd.getrandbits(8) for _ in range(n)) + b"\0", dtype=np.uint8)[:n].copy() if n else np.zeros(1, dtype=np.uint8)[:0].copy()
        if n == 0:
            continue
        for mis in (0, 3):
            assert np.array_equal(find(a, 0, 8 * n, 0, mis), find(a, 0, 8 * n, 1, mis)), n


@pytest.mark.parametrize("kind", ["sparse", "text"])
def test_entry_larger_than_any_window_bounded_memory(tmp_path, kind):
    """A ZIP64 entry of 3 GiB (the reference streams any size through 32 767 bytes, mz_strm_zlib.c:51,116-193): the
    drop-in decodes it window by window on the device (64 MiB windows, 32 KiB of history; a wave per DEFLATE block where the
    window starts at a block header, the resumable serial kernel elsewhere) -- same sizes, CRC verdicts and status as the
    all-reference reader, in a process whose peak RSS stays far below the entry.  "sparse" compresses 300:1 (few, huge
    blocks), "text" is 1 GiB of the bench corpus at level 1 (~0.4: thousands of blocks per window)."""
    import json
    import os
    import subprocess
    import sys
    import zipfile

    import oracle

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing")
    path = str(tmp_path / "big.zip")
    total = (3 if kind == "sparse" else 1) * (1 << 30) + 12345
    piece = (b"sparse " * 1024 + bytes(120000)) * 8                      # ~1 MiB, compresses 300:1
    if kind == "text":
        piece = synth.bench_corpus()[0] * 2
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as zf:
        with zf.open("huge.bin", "w", force_zip64=True) as f:
            left = total
            while left > 0:
                k = min(left, len(piece))
                f.write(piece[:k])
                left -= k
        zf.writestr("small.txt", synth.corpus()[:70000])
    ref = oracle.ref()
    table = ref.zip_index(path)
    assert len(table) == 2 and int(table[0, 4]) == total
    cd = table[:, 6].copy()
    sec_r, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=1, own_crc=False)
    assert (st_r == 0).all() and int(ulen_r[0]) == total
    prog = (
        "import sys, json, resource\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, oracle\n"
        "hip = oracle.MzDriver(%r)\n"
        "cd = np.array(%r, dtype=np.int64)\n"
        "sec, crc, ulen, st = hip.zip_read_all(%r, cd, nthreads=1, own_crc=False)\n"
        "print(json.dumps(dict(sec=sec, crc=[int(x) for x in crc], ulen=[int(x) for x in ulen], st=[int(x) for x in st],\n"
        "                      rss_kib=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)))\n" % (ROOT, DROP, [int(x) for x in cd], path))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert got["st"] == [0, 0] and got["ulen"] == [int(x) for x in ulen_r] and got["crc"] == [int(x) for x in crc_r]
    # the same process shape on the small entry alone: what the HIP runtime, numpy and the libraries cost by themselves
    prog0 = prog.replace(repr([int(x) for x in cd]), repr([int(cd[1])]))
    r0 = subprocess.run([sys.executable, "-c", prog0], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r0.returncode == 0, r0.stderr[-2000:]
    base = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][-1])
    print("%.0f GiB %s entry through the drop-in: %.1f s = %.2f GiB/s, peak RSS %.0f MiB (the same process reading a 70 KB entry: %.0f MiB)"
          % (total / (1 << 30), kind, got["sec"], total / got["sec"] / (1 << 30), got["rss_kib"] / 1024, base["rss_kib"] / 1024))
    assert got["rss_kib"] - base["rss_kib"] < 512 * 1024                  # a 64 MiB window, 16 MiB of input, staging: not the entry
    print("the all-reference reader: %.1f s" % sec_r)


def test_xz_entry_larger_than_any_window_bounded_memory(tmp_path):
    """Method 95 in bounded memory (mz_strm_lzma.c:127-128,147-241 streams an .xz entry through 32 767 bytes): a 3 GiB ZIP64
    entry written by the ALL-REFERENCE writer as one .xz stream is extracted through the unmodified mz_zip reader on the
    drop-in mz_stream_lzma READ stream -- the container walked by shim_lzma.c, the LZMA2 chunks decoded window by window by
    k_lzma2_run, the block's CRC-64 carried on the device -- in a process whose peak RSS stays far below the entry: size,
    CRC-32 and status (mz_zip_entry_read_close compares the CRC and, for a data descriptor, TOTAL_IN) as the all-reference
    reader's."""
    import json
    import os
    import subprocess
    import sys

    import oracle

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing")
    path = str(tmp_path / "xbig.zip")
    total = 3 * (1 << 30) + 54321
    piece = (b"sparse " * 1024 + bytes(120000)) * 8 + synth.corpus()[:20000]      # ~1 MiB; one wave decodes long matches fast
    ref = oracle.ref()
    ref.zip_write_repeat(path, piece, total, method=95, level=1)
    table = ref.zip_index(path)
    assert len(table) == 2 and int(table[0, 4]) == total and int(table[0, 0]) == 95
    cd = table[:, 6].copy()
    sec_r, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=1, own_crc=False)
    assert (st_r == 0).all() and int(ulen_r[0]) == total
    prog = (
        "import sys, json, resource\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, oracle\n"
        "hip = oracle.MzDriver(%r)\n"
        "cd = np.array(%r, dtype=np.int64)\n"
        "sec, crc, ulen, st = hip.zip_read_all(%r, cd, nthreads=1, own_crc=False)\n"
        "print(json.dumps(dict(sec=sec, crc=[int(x) for x in crc], ulen=[int(x) for x in ulen], st=[int(x) for x in st],\n"
        "                      rss_kib=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)))\n" % (ROOT, DROP, [int(x) for x in cd], path))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert got["st"] == [0, 0] and got["ulen"] == [int(x) for x in ulen_r] and got["crc"] == [int(x) for x in crc_r]
    prog0 = prog.replace(repr([int(x) for x in cd]), repr([int(cd[1])]))
    r0 = subprocess.run([sys.executable, "-c", prog0], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r0.returncode == 0, r0.stderr[-2000:]
    base = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][-1])
    print("3 GiB .xz entry (archive %.1f MiB) through the drop-in: %.1f s = %.3f GiB/s, peak RSS %.0f MiB (the same process reading a small entry: %.0f MiB); "
          "the all-reference reader: %.1f s" % (os.path.getsize(path) / 2**20, got["sec"], total / got["sec"] / (1 << 30), got["rss_kib"] / 1024, base["rss_kib"] / 1024, sec_r))
    assert got["rss_kib"] - base["rss_kib"] < 512 * 1024                  # dictionary + a 64 MiB window + 16 MiB of input, twice (staging): not the entry


def test_written_entry_larger_than_any_segment_bounded_memory(tmp_path):
    """WRITE side of the same property (mz_strm_zlib.c:203-264 stages any entry through 32 767 bytes): a 3 GiB ZIP64 entry
    is written through the unmodified mz_zip_writer on the drop-in mz_stream_zlib WRITE stream -- 8 MiB segments, one K4
    launch each, nothing else kept -- in a process whose peak RSS stays far below the entry, and the ALL-REFERENCE reader
    extracts it with its own CRC verification."""
    import json
    import os
    import subprocess
    import sys

    import oracle

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing")
    path = str(tmp_path / "wbig.zip")
    total = 3 * (1 << 30) + 4321
    prog = (
        "import sys, json, resource, time\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, oracle\n"
        "from tests import synth\n"
        "hip = oracle.MzDriver(%r)\n"
        "piece = (synth.corpus()[:300000] + bytes(700000))\n"
        "t0 = time.time()\n"
        "hip.zip_write_repeat(%r, piece, TOTAL, method=8, level=1)\n"
        "print(json.dumps(dict(sec=time.time() - t0, rss_kib=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)))\n"
        % (ROOT, DROP, path))
    # the same process shape writing a 70 KB entry first: what the HIP runtime, numpy and the libraries cost by themselves
    r0 = subprocess.run([sys.executable, "-c", prog.replace("TOTAL", "70000")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r0.returncode == 0, r0.stderr[-2000:]
    base = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][-1])
    r = subprocess.run([sys.executable, "-c", prog.replace("TOTAL", str(total))], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ref = oracle.ref()
    table = ref.zip_index(path)
    assert len(table) == 2 and int(table[0, 4]) == total
    _, crc_r, ulen_r, st_r = ref.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=False)
    assert (st_r == 0).all() and int(ulen_r[0]) == total           # st 0 = the reference's own CRC verification passed
    print("3 GiB entry written through the drop-in: %.1f s, %.2f GiB/s, archive %.1f MiB, peak RSS %.0f MiB (a 70 KB entry: %.0f MiB)"
          % (got["sec"], total / 2**30 / got["sec"], os.path.getsize(path) / 2**20, got["rss_kib"] / 1024, base["rss_kib"] / 1024))
    assert got["rss_kib"] - base["rss_kib"] < 512 * 1024


def test_large_entry_device_resident(gpu):
    """mzhip_inflate_large: one large entry where it lies in HBM, a wave per DEFLATE block window after window, against the
    batch kernel's one wave for the same entry (mzhip_inflate_batch): the four result words agree for a whole stream, for
    a buffer that is too small, a truncated and a corrupted stream; bytes against zlib; both are timed."""
    import time

    import torch

    L = gpu.mz.lib()
    L.mzhip_inflate_large.restype = C.c_int32
    L.mzhip_inflate_large.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
    L.mzhip_inflate_batch.restype = C.c_int32
    L.mzhip_inflate_batch.argtypes = [C.c_void_p] * 6 + [C.c_uint32] + [C.c_void_p] * 5
    text, _ = synth.bench_corpus()
    dev = torch.device("cuda", 0)

    def both(z, cap):
        d_in = torch.from_numpy(np.frombuffer(z + bytes(64), dtype=np.uint8).copy()).to(dev)
        outs = []
        for which in ("large", "batch"):
            d_out = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            t0 = time.time()
            if which == "large":
                ol, iu, ck, st = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32()
                assert L.mzhip_inflate_large(d_in.data_ptr(), len(z), d_out.data_ptr(), cap, C.byref(ol), C.byref(iu), C.byref(ck), C.byref(st), None) == 0
                torch.cuda.synchronize()
                r = (st.value, ol.value, iu.value, ck.value)
            else:
                m64 = torch.tensor([0, 0], dtype=torch.int64, device=dev)
                m32 = torch.tensor([len(z), cap, 0, 0, 0, 0], dtype=torch.int32, device=dev)
                p = m32.data_ptr()
                assert L.mzhip_inflate_batch(d_in.data_ptr(), m64.data_ptr(), p, d_out.data_ptr(), m64.data_ptr() + 8, p + 4, 1, p + 8, p + 12, p + 16, p + 20, None) == 0
                torch.cuda.synchronize()
                v = m32.cpu().numpy()
                r = (int(v[5]), int(np.uint32(v[2])), int(np.uint32(v[3])), int(np.uint32(v[4])))
            outs.append((r, time.time() - t0, d_out[:r[1]].cpu().numpy().tobytes()))
        return outs

    d = (text + text[::-1][:100000]) * 60 + bytes(5000000) + text * 10               # ~43 MB
    for lvl in (6, 1):
        z = synth.deflate_raw(d, lvl)
        both(z[:200000], 1 << 20)                                                      # (warm both paths up)
        (ra, ta, oa), (rb, tb, ob) = both(z, len(d))
        assert ra == rb == (0, len(d), len(z), zlib.crc32(d)) and oa == d, (lvl, ra, rb)
        print("level %d: %d -> %d bytes; a wave per block %.1f ms (%.2f GB/s), one wave %.1f ms (%.2f GB/s)"
              % (lvl, len(z), len(d), ta * 1e3, len(d) / ta / 1e9, tb * 1e3, len(d) / tb / 1e9))
        assert ta * 5 < tb
        (ra, _, oa), (rb, _, ob) = both(z, len(d) - 1000)
        assert ra[0] == rb[0] == -200 and oa == d[:ra[1]], (ra, rb)
        (ra, _, oa), (rb, _, ob) = both(z[:len(z) // 2], len(d))
        assert ra[0] == rb[0] == -5 and oa == d[:ra[1]] and ob == d[:rb[1]], (ra, rb)
        zz = bytearray(z)
        zz[len(z) // 3] ^= 0x10
        (ra, _, oa), (rb, _, ob) = both(bytes(zz), 2 * len(d))
        assert ra[0] == rb[0], (ra, rb)
        if ra[0] == 0:
            assert ra == rb and oa == ob
