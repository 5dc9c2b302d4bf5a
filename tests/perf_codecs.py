"""GPU probe (not a pytest): throughput of K3 (LZMA decode), the .xz kernel, K4 (DEFLATE encode) and the SHA batch
kernel at config-like shapes.  Usage: python tests/perf_codecs.py [lzma|xz|deflate|sha|lzmaenc|crc|both|all]"""
import ctypes as C
import sys
import time
import zlib

import numpy as np
import torch

import os  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import gpu_util, synth  # noqa: E402
from tests.test_oracle import _zip_lzma  # noqa: E402

mz = gpu_util.mz
L = mz.lib()
L.mzhip_lzma_batch.restype = C.c_int32
L.mzhip_lzma_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
L.mzhip_deflate_batch.restype = C.c_int32
L.mzhip_deflate_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best


if which in ("lzma", "both", "all"):
    n_unique, n_total, size = 16, int(sys.argv[2]) if len(sys.argv) > 2 else 4608, 1 << 20  # 2 x 2304 resident waves
    rnd = np.random.RandomState(3)
    words = synth.corpus().split()
    datas = []
    for u in range(n_unique):
        blob = b" ".join(words[i] for i in rnd.randint(0, len(words), size=240000))
        datas.append(blob[:size])
    import os
    lc = int(os.environ.get("LZMA_LC", "3"))  # profiles/ab_k3.sh: lc = 0 streams fit a literal model of 0x300 probabilities
    if lc == 3:
        pays = [_zip_lzma(d) for d in datas]
    else:
        import lzma as _lz

        def _zl(d):
            raw = _lz.compress(d, format=_lz.FORMAT_ALONE, filters=[dict(id=_lz.FILTER_LZMA1, preset=6, lc=lc, lp=0, pb=2)])
            return bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]
        pays = [_zl(d) for d in datas]
    idx = np.arange(n_total) % n_unique
    b = gpu_util.make_batch([pays[i] for i in idx], [size] * n_total)
    out_len, in_used, crc, status = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(4))
    mo = torch.full((n_total,), size, dtype=torch.int64, device=dev)

    def run():
        assert L.mzhip_lzma_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                  b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                  mo.data_ptr(), n_total, out_len.data_ptr(), in_used.data_ptr(), crc.data_ptr(),
                                  status.data_ptr(), None) == 0
    ms = timed(run, 2)
    want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]
    ok = bool((status.cpu().numpy() == 0).all() and (mz.u32(crc) == want).all())
    ratio = sum(len(p) for p in pays) / (n_unique * size)
    print("LZMA decode: %d x %d B, ratio %.3f: %.1f ms  %.2f GiB/s out  ok=%s" % (
        n_total, size, ratio, ms, n_total * size / 2**30 / (ms / 1e3), ok), flush=True)

if which in ("deflate", "deflate_only", "both", "all"):
    n_unique, n_total, size = 512, 20000, 65536
    datas = synth.slices(n_unique, size, 1234)
    idx = np.arange(n_total) % n_unique
    b = gpu_util.make_batch([datas[i] for i in idx], [size + size // 8 + 64] * n_total)
    out_len, crc, status = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(3))

    def run2():
        assert L.mzhip_deflate_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                     b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(), None,
                                     n_total, out_len.data_ptr(), crc.data_ptr(), status.data_ptr(), None) == 0
    ms = timed(run2)
    want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]
    ol = out_len.cpu().numpy()
    h = b["d_out"].cpu().numpy()
    ok = bool((status.cpu().numpy() == 0).all() and (mz.u32(crc) == want).all())
    for i in range(0, n_total, 997):
        ok = ok and zlib.decompress(gpu_util.entry_bytes(b, h, i, int(ol[i])), -15) == datas[idx[i]]
    print("DEFLATE encode: %d x %d B: %.1f ms  %.2f GiB/s in  ratio %.3f  ok=%s" % (
        n_total, size, ms, n_total * size / 2**30 / (ms / 1e3), ol.sum() / (n_total * size), ok), flush=True)

if which in ("deflate", "both", "all", "deflate_levels"):
    # the classes behind COMPRESS_LEVEL: 1 = fast, 4 = four candidates + lazy rule, 6 = + cost parse; zlib's own ratios beside them
    L.mzhip_deflate_batch_level.restype = C.c_int32
    L.mzhip_deflate_batch_level.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_int32, C.c_int32] + [C.c_void_p] * 4
    n_unique, n_total, size = 512, 8192, 65536
    datas = synth.slices(n_unique, size, 1234)
    idx = np.arange(n_total) % n_unique
    b = gpu_util.make_batch([datas[i] for i in idx], [size + size // 8 + 64] * n_total)
    out_len, crc, status = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(3))
    def k4_sections():
        if not hasattr(L, "mzhip_prof_read"): return
        buf = (C.c_ulonglong * 32)()
        L.mzhip_prof_read(buf, 1)
        names = {16: "block set-up", 17: "hash, candidates, bucket update", 18: "match measurement", 19: "lazy rule + greedy selection",
                 20: "tokens out, histograms", 21: "CRC of the input", 22: "codes, block costs, header", 23: "pass 2 (bits out)",
                 24: "cost parse: price list", 25: "cost parse: dynamic programme", 26: "cost parse: choices picked up", 27: "cost parse: block fetch"}
        tot = float(sum(buf[i] for i in names)) or 1.0
        for i in sorted(names, key=lambda k: -buf[k]):
            if buf[i]: print("  %-36s %5.1f %%" % (names[i], 100.0 * buf[i] / tot))
    for level in (1, 4, 6, 9):
        k4_sections() if False else (hasattr(L, "mzhip_prof_read") and L.mzhip_prof_read((C.c_ulonglong * 32)(), 1))
        def runl():
            assert L.mzhip_deflate_batch_level(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                               b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(), None,
                                               n_total, level, 15, out_len.data_ptr(), crc.data_ptr(), status.data_ptr(), None) == 0
        ms = timed(runl)
        ol = out_len.cpu().numpy()
        h = b["d_out"].cpu().numpy()
        ok = bool((status.cpu().numpy() == 0).all())
        for i in range(0, n_total, 211):
            ok = ok and zlib.decompress(gpu_util.entry_bytes(b, h, i, int(ol[i])), -15) == datas[idx[i]]
        zr = sum(len(zlib.compress(d, level)) - 6 for d in datas) / (n_unique * size)
        print("DEFLATE encode level %d: %d x %d B: %.1f ms  %.2f GiB/s in  ratio %.4f (zlib-%d: %.4f)  ok=%s" % (
            level, n_total, size, ms, n_total * size / 2**30 / (ms / 1e3), ol.sum() / (n_total * size), level, zr, ok), flush=True)
        k4_sections()

if which in ("xz", "all"):
    import lzma as pylzma

    L.mzhip_xz_batch.restype = C.c_int32
    L.mzhip_xz_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
    n_unique, n_total, size = 16, 2048, 1 << 20
    rnd = np.random.RandomState(3)
    words = synth.corpus().split()
    datas = [b" ".join(words[i] for i in rnd.randint(0, len(words), size=240000))[:size] for _ in range(n_unique)]
    pays = [pylzma.compress(d, format=pylzma.FORMAT_XZ, preset=6) for d in datas]      # CRC64 check, like the reference writer
    idx = np.arange(n_total) % n_unique
    b = gpu_util.make_batch([pays[i] for i in idx], [size] * n_total)
    out_len, in_used, crc, status = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(4))

    def run3():
        assert L.mzhip_xz_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(), None, n_total,
                                out_len.data_ptr(), in_used.data_ptr(), crc.data_ptr(), status.data_ptr(), None) == 0
    ms = timed(run3, 2)
    want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]
    ok = bool((status.cpu().numpy() == 0).all() and (mz.u32(crc) == want).all())
    print(".xz decode (CRC64 verified): %d x %d B, ratio %.3f: %.1f ms  %.2f GiB/s out  ok=%s" % (
        n_total, size, sum(len(p) for p in pays) / (n_unique * size), ms, n_total * size / 2**30 / (ms / 1e3), ok), flush=True)

if which in ("sha", "all"):
    import hashlib

    L.mzhip_sha_batch.restype = C.c_int32
    L.mzhip_sha_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_uint32] + [C.c_void_p] * 2
    n_unique, n_total, size = 512, 65536, 65536
    datas = synth.slices(n_unique, size, 1234)
    idx = np.arange(n_total) % n_unique
    blob = torch.from_numpy(np.frombuffer(b"".join(datas), dtype=np.uint8).copy()).to(dev)
    off = torch.from_numpy((idx.astype(np.int64) * size)).to(dev)
    ln = torch.full((n_total,), size, dtype=torch.int32, device=dev)
    dg = torch.zeros(n_total * 32, dtype=torch.uint8, device=dev)
    for alg, fn in ((23, hashlib.sha256), (20, hashlib.sha1)):
        def run4():
            assert L.mzhip_sha_batch(blob.data_ptr(), off.data_ptr(), ln.data_ptr(), n_total, alg, dg.data_ptr(), None) == 0
        ms = timed(run4)
        h = dg.cpu().numpy().reshape(n_total, 32)
        sz = fn().digest_size
        ok = all(h[i, :sz].tobytes() == fn(datas[idx[i]]).digest() for i in range(0, n_total, 257))
        print("SHA (alg %d) batch: %d x %d B: %.1f ms  %.2f GiB/s  ok=%s" % (alg, n_total, size, ms,
                                                                            n_total * size / 2**30 / (ms / 1e3), ok), flush=True)

if which in ("lzmaenc", "all"):
    import lzma as pylzma

    L.mzhip_lzma_encode_batch.restype = C.c_int32
    L.mzhip_lzma_encode_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4
    L.mzhip_lzma_encode_batch_preset.restype = C.c_int32
    L.mzhip_lzma_encode_batch_preset.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_int32] + [C.c_void_p] * 4
    import os
    preset = int(os.environ.get("LZMA_PRESET", "1"))  # 1 = the fast class (what mzhip_lzma_encode_batch runs); 4 / 6 / 9: four candidates, 4 / 8 / 16 links
    for n_unique, n_total, size, tag in ((512, 18432, 65536, "64 KiB entries"), (16, 2304, 1 << 20, "1 MiB entries")):
        if preset != 1 and size == 65536:
            continue
        if size == 65536:
            datas = synth.slices(n_unique, size, 1234)
        else:
            rnd = np.random.RandomState(3)
            words = synth.corpus().split()
            datas = [b" ".join(words[i] for i in rnd.randint(0, len(words), size=240000))[:size] for _ in range(n_unique)]
        idx = np.arange(n_total) % n_unique
        b = gpu_util.make_batch([datas[i] for i in idx], [size + size // 8 + 1024] * n_total)
        out_len, crc, status = (torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(3))

        def run5():
            assert L.mzhip_lzma_encode_batch_preset(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(), size,
                                                    b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(), None,
                                                    n_total, preset, out_len.data_ptr(), crc.data_ptr(), status.data_ptr(), None) == 0
        ms = timed(run5, 2)
        want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]
        ol = out_len.cpu().numpy()
        h = b["d_out"].cpu().numpy()
        ok = bool((status.cpu().numpy() == 0).all() and (mz.u32(crc) == want).all())
        for i in range(0, n_total, 1777):
            z = gpu_util.entry_bytes(b, h, i, int(ol[i]))
            ok = ok and pylzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=pylzma.FORMAT_ALONE) == datas[idx[i]]
        print("LZMA encode (%s, preset %d): %d x %d B: %.1f ms  %.2f GiB/s in  ratio %.3f  ok=%s" % (
            tag, preset, n_total, size, ms, n_total * size / 2**30 / (ms / 1e3), ol.sum() / (n_total * size), ok), flush=True)

if which in ("crc", "all"):
    n_total, size = 65536, 65536           # STORE entries: CRC-32 only (mz_crypt_crc32_update over the raw stream)
    blob = torch.randint(0, 256, (n_total * size,), dtype=torch.uint8, device=dev)      # 4 GiB, HBM-resident
    off = torch.arange(n_total, dtype=torch.int64, device=dev) * size
    ln = torch.full((n_total,), size, dtype=torch.int32, device=dev)

    def run6():
        return mz.crc32_batch(blob, off, ln)
    ms = timed(run6)
    crc = mz.u32(run6())
    h = blob[:size * 3].cpu().numpy()
    ok = all(int(crc[i]) == zlib.crc32(h[i * size:(i + 1) * size].tobytes()) for i in range(3))
    print("CRC-32 batch: %d x %d B: %.1f ms  %.1f GiB/s  (%.2f TB/s)  ok=%s" % (n_total, size, ms, n_total * size / 2**30 / (ms / 1e3),
                                                                         n_total * size / 1e12 / (ms / 1e3), ok), flush=True)

# measurement builds (make PROF=1): per-section cycle sums of K4
if which in ("deflate", "both", "all") and hasattr(L, "mzhip_prof_read"):
    buf = (C.c_ulonglong * 32)()
    L.mzhip_prof_read(buf, 1)
    names = {16: "block set-up", 17: "hash, candidates, bucket update", 18: "match measurement", 19: "lazy rule + greedy selection",
             20: "tokens out, histograms", 21: "CRC of the input", 22: "codes, block costs, header", 23: "pass 2 (bits out)",
             24: "cost parse: price list", 25: "cost parse: dynamic programme", 26: "cost parse: choices picked up", 27: "cost parse: block fetch"}
    tot = float(sum(buf[i] for i in names)) or 1.0
    for i in sorted(names, key=lambda k: -buf[k]):
        print("  %-36s %5.1f %%" % (names[i], 100.0 * buf[i] / tot))
