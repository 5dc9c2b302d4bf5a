"""GPU-box probe (not a pytest): SURVEY 8(d) side measurements.
  1. CPU reference (oracle/_ref = the compiled reference + zlib 1.2.11 + liblzma 5.2.5) on the configs the bench
     line does not cover: config 4 (LZMA decode, 1 MiB entries) and config 5 (DEFLATE encode level 1), all host threads.
  2. The config-2 decode timed three ways: kernel only / H2D + kernel + D2H of {crc,len,status} / H2D + kernel +
     D2H of every decoded byte (pinned host memory) -- the PCIe-inclusive rates DESIGN.md quotes.
Usage: python tests/perf_modes.py [cpu|modes|all]"""
import os
import struct
import sys
import tempfile
import threading
import time
import zlib

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from tests import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
T = os.cpu_count() or 1


def markov_entries(n_unique, size):
    rnd = np.random.RandomState(3)
    words = synth.corpus().split()
    return [b" ".join(words[i] for i in rnd.randint(0, len(words), size=240000))[:size] for _ in range(n_unique)]


if which in ("cpu", "all"):
    ref = oracle.ref()
    with tempfile.TemporaryDirectory() as tmp:
        # config 4 shape: 1 MiB text-like entries, method 14, reference writer (preset 6, EOS marker)
        datas = markov_entries(8, 1 << 20)
        blob = np.frombuffer(b"".join(datas), dtype=np.uint8)
        n = max(2 * T, 64)
        small = os.path.join(tmp, "s.zip")
        t0 = time.time()
        ref.zip_write(small, blob, np.arange(8, dtype=np.int64) << 20, np.full(8, 1 << 20, dtype=np.int32), method=14, level=6)
        t_w = time.time() - t0
        # the reference encoder is ~2 MB/s: replicate its 8 entries (payload bytes as it wrote them) into an n-entry archive
        st = ref.zip_index(small)
        raw = open(small, "rb").read()
        path = os.path.join(tmp, "l.zip")
        with open(path, "wb") as f:
            cd = []
            for i in range(n):
                m, flag, crc, cs, us, _, _, po = (int(v) for v in st[i % 8])
                name = b"e/%06d" % i
                hdr = struct.pack("<IHHHHHIIIHH", 0x04034B50, 63, flag & ~8, m, 0, 0x21, crc & 0xFFFFFFFF, cs, us, len(name), 0)
                cd.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0x033F, 63, flag & ~8, m, 0, 0x21, crc & 0xFFFFFFFF, cs,
                                      us, len(name), 0, 0, 0, 0, 0, f.tell()) + name)
                f.write(hdr + name + raw[po:po + cs])
            cd_off = f.tell()
            f.write(b"".join(cd))
            f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, n, n, f.tell() - cd_off, cd_off, 0))
        tab = ref.zip_index(path)
        for th in (1, T):
            cd = tab[:, 6].copy() if th == T else tab[:8, 6].copy()
            sec, crc, ulen, st = ref.zip_read_all(path, cd, nthreads=th, own_crc=False)
            assert (st == 0).all()
            print("CPU reference LZMA decode: %d x 1 MiB, ratio %.3f, %d thread(s): %.3f GiB/s" % (
                len(cd), tab[:, 3].sum() / tab[:, 4].sum(), th, ulen.sum() / 2**30 / sec), flush=True)
        print("CPU reference LZMA encode (1 thread): %.4f GiB/s" % (8 / 1024 / t_w), flush=True)
        # config 5 shape: 64 KiB entries, DEFLATE level 1 through mz_zip_writer; one writer (= one archive) per thread
        ds = synth.slices(512, 65536, 1234)
        blob = np.frombuffer(b"".join(ds), dtype=np.uint8)
        per = 2000
        offs = (np.arange(per) % 512).astype(np.int64) * 65536
        lens = np.full(per, 65536, dtype=np.int32)
        for th in (1, T):
            ts = [threading.Thread(target=ref.zip_write, args=(os.path.join(tmp, "w%d.zip" % i), blob, offs, lens, 8, 1))
                  for i in range(th)]
            t0 = time.time()
            [t.start() for t in ts]
            [t.join() for t in ts]
            sec = time.time() - t0
            print("CPU reference DEFLATE encode L1: %d x 64 KiB, %d thread(s): %.3f GiB/s in" % (
                per * th, th, per * th * 65536 / 2**30 / sec), flush=True)

if which in ("modes", "all"):
    import torch
    from tests import gpu_util

    mz = gpu_util.mz
    mz.require_gpu()
    n_unique, n_total, size = 512, 20000, 65536
    datas = synth.slices(n_unique, size, 1234)
    pays = [synth.deflate_raw(d) for d in datas]
    idx = np.arange(n_total) % n_unique
    b = gpu_util.make_batch([pays[i] for i in idx], [size] * n_total)
    h_in = b["d_in"].cpu().pin_memory()
    h_out = torch.empty(b["d_out"].numel(), dtype=torch.uint8).pin_memory()
    d_in, d_out = b["d_in"], b["d_out"]
    want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]

    def best(fn, reps=3):
        t = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        return min(t), r

    def mode_i():
        return gpu_util.run_inflate(b)

    def mode_ii():
        d_in.copy_(h_in, non_blocking=True)
        return gpu_util.run_inflate(b)          # returns host copies of out_len / in_used / crc / status

    def mode_iii():
        d_in.copy_(h_in, non_blocking=True)
        r = gpu_util.run_inflate(b)
        h_out.copy_(d_out, non_blocking=True)
        return r

    tot = n_total * size / 2**30
    for name, fn in (("(i) kernel only (+12 B/entry D2H)", mode_i), ("(ii) H2D compressed + kernel + D2H crc/len/status", mode_ii),
                     ("(iii) H2D compressed + kernel + D2H all decoded bytes", mode_iii)):
        sec, r = best(fn)
        ok = bool((r[3] == 0).all() and (r[2] == want).all())
        print("%-58s %7.1f ms  %6.1f GiB/s out  ok=%s" % (name, sec * 1e3, tot / sec, ok), flush=True)
