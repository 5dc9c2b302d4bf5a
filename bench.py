#!/usr/bin/env python3
"""bench.py -- the benchmarks of BASELINE.json on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5] [--scaling strong|weak]

config 2 (default, the headline): DEFLATE level-6 100 000 x 64 KiB entries, inflate + fused CRC-32 (k_inflate_batch)
config 3: DEFLATE level-6 1 000 000 x 8 KiB entries, same kernel (the small-entry shape of the 8-GPU sharding config)
config 4: LZMA (ZIP method 14, preset 6, EOS marker) 10 000 x 1 MiB entries, range decode + fused CRC-32 (k_lzma_batch)
config 5: DEFLATE compress (level 1) of 100 000 x 64 KiB buffers + CRC-32 of the input (k_deflate_batch)

step    : ONE pass of the hot path over the whole batch = one batch launch over every entry of this rank's shard (the
          CRC-32 of every entry is computed inside that launch, every step) and (N > 1) the RCCL gather of the per-entry
          {crc, status} words -- the only collective on the path -- and the comparison of those words with the central
          directory's values (every step, inside the clock).  Inputs and outputs are resident in HBM.
launch  : `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) re-executes itself under
          `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); when the box has fewer than N
          GPUs it prints {"error": ...} and exits non-zero.  world == --gpus is asserted in every case.
scaling : strong by default, as north_star states it: ONE entry table (100 000 entries for config 2) is cut into N
          contiguous slices balanced by compressed + uncompressed bytes (archive.shard_bounds), rank r decodes slice r.
          --scaling weak gives every rank its own full-size table.
data    : synthetic, SURVEY 8(d): entries are slices of C = appnote.txt || appnote.iz.txt || alice29.txt of the reference
          tree (oracle/_ref/corpus.bin, built by oracle/make_corpus.py; CPython's pydoc prose when it did not travel);
          config 4 uses the seeded order-2 word-Markov expansion of C.  Every entry of config 2 is its OWN stream
          (100 000 unique slices, ~240 core-seconds of level-6 compression spread over all host cores; config 3: 524 288
          unique slices of the corpus + 4 MiB of its 4 KiB pieces in random order, config 4: 1 024 unique 1 MiB streams), compressed with exactly the reference writer's parameters
          (mz_strm_zlib.c:87: raw, 32 KiB window, memLevel 8 -- the cpu_baseline leg checks that the reference writer
          emits the same bytes).  Only when the host is too slow (--gen-seconds runs out) the streams made so far are
          tiled to the entry count, and `data` says how many were tiled; every entry has its own copy of its compressed
          bytes and its own output region in HBM either way.

Besides the contract fields the JSON line carries
  roofline     : the dominant kernel against the 8 TB/s HBM roofline; achieved = algorithmic bytes (compressed bytes
                 read once + decompressed bytes written once) / mean launch duration, measured with HIP
                 events on the launch stream inside the timed region (max over ranks).  `traffic` is NOT measured by this
                 run (counters cannot be collected from inside a timed run): it is the bytes of the L2's memory-side requests
                 (size x count) of the committed rocprofv3 --pmc passes named in `traffic_source`, `traffic_raw` what
                 FETCH_SIZE + WRITE_SIZE say of the same passes (gfx950 tallies 128-byte reads at 64), both null when the
                 passes were taken on another workload shape;
  cpu_baseline : the UNMODIFIED reference path (oracle/_ref) on the host cores of the same box, on a bounded sample of
                 the same workload.  Rank 0, N = 1 only;
  other_configs: (the default run: config 2, N = 1) BASELINE.json configs[2], [3], [4] -- `--config 3`, `4`, `5` run as child
                 processes behind the headline, each condensed to value, ms_per_step, crc32_match_rate, unique streams,
                 roofline {achieved, frac, traffic, traffic_source} and cpu_baseline;
  legs         : (config 2, N = 1) SURVEY 8(d) i-iii: kernel only / H2D of the compressed bytes + kernel + D2H of
                 {crc, len, status} from pinned host memory / the reference's unmodified mz_zip_reader loop on the
                 drop-in library (prime + vtbl shims) into host buffers, GiB/s of decompressed bytes each; and ONE large entry
                 (512 MiB of the corpus at level 1) through the same reader loop, per entry (`one_large_entry_vtbl`: window
                 mode, a wave per DEFLATE block) and primed (`one_large_entry_primed`), with the reference's one thread on the
                 same archive as `cpu_baseline.one_large_entry`.
"""
import argparse
import ctypes as C
import importlib
import json
import multiprocessing as mp
import os
import random
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    2: dict(entries=100000, size=65536, codec="inflate", kernel="k_inflate_batch", unique=100000,
            metric="decompressed GiB/s (whole node) + CRC32 match rate, 100k x 64KiB DEFLATE entries",
            workload="BASELINE.json configs[1]: DEFLATE level-6 %d x %d B entries, inflate + fused CRC32 (mzhip_inflate_batch), device-resident"),
    3: dict(entries=1000000, size=8192, codec="inflate", kernel="k_inflate_batch", unique=524288, extend_mib=4, gen_scale=1.4,
            metric="decompressed GiB/s (whole node) + CRC32 match rate, 1M x 8KiB DEFLATE entries",
            workload="BASELINE.json configs[2]: DEFLATE level-6 %d x %d B small entries, inflate + fused CRC32 (mzhip_inflate_batch), device-resident"),
    4: dict(entries=10000, size=1 << 20, codec="lzma", kernel="k_lzma_slot_batch (+ k_lzma_batch over the streams it gives back)", unique=1024,
            metric="decompressed GiB/s (whole node) + CRC32 match rate, 10k x 1MiB LZMA entries",
            workload="BASELINE.json configs[3]: LZMA (method 14, preset 6) %d x %d B entries, range decode + fused CRC32 (mzhip_lzma_batch), device-resident"),
    5: dict(entries=100000, size=65536, codec="deflate", kernel="k_deflate_batch", unique=100000,
            metric="compressed-input GiB/s (whole node) + round-trip match rate, 100k x 64KiB DEFLATE level-1 compress",
            workload="BASELINE.json configs[4]: DEFLATE compress level 1 of %d x %d B buffers + CRC32 of the input (mzhip_deflate_batch), device-resident"),
}


def measured_traffic(cfg, n, size):
    """(bytes the L2 moved to and from memory per launch, the raw FETCH_SIZE + WRITE_SIZE sum, where the numbers come from): the
    committed PMC passes (profiles/r*/hbm_traffic*.json, newest round first), only when they were taken on this very workload
    shape; counters cannot be collected from inside a timed run, so these are STATIC numbers and `traffic_source` says so.
    Round 5 on: traffic = the L2's memory-side requests summed as size x count (TCC_EA0_RDREQ_{32B,64B,128B}, WRREQ): on gfx950
    every read request of K1 is 128 bytes and FETCH_SIZE tallies those at 64, so the derived counters under-report by the
    read half (profiles/r4/pmc_calibration.json); `traffic_raw` keeps what they say."""
    name = "hbm_traffic.json" if cfg == 2 else "hbm_traffic_cfg%d.json" % cfg
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        from req_harvest import kernel_src_hash
        mine = kernel_src_hash()
    except Exception:  # noqa: BLE001
        mine = None
    for rnd in ("r6", "r5", "r4", "r3", "r2"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, name)) as f:
                t = json.load(f)
            if n == t.get("entries", 100000) and size == t.get("entry_bytes", 65536):
                if "read_bytes_per_launch" in t:
                    theirs = t.get("kernel_src_hash")
                    same = ("taken on EXACTLY the device sources this run was built from (hash %s)" % mine if theirs and theirs == mine else
                            "taken on OTHER device sources than this run's (%s against %s)" % (theirs, mine) if theirs else "device sources of the measurement not recorded")
                    return (t["read_bytes_per_launch"] + t["write_bytes_per_launch"], t.get("traffic_raw"),
                            "static: profiles/%s/%s (rocprofv3 --pmc passes of the L2's memory-side requests by size, bytes = size x count, commit %s; %s; "
                            "regenerate with `bash profiles/gpu.sh evidence <name>`), not measured by this run" % (rnd, name, t.get("commit", "unknown"), same))
                raw = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                return (raw + t["fetch_bytes_per_launch"], raw,
                        "static: profiles/%s/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of commit %s; traffic = 2 x FETCH + WRITE, the "
                        "gfx950 correction of the guide: 128-byte read requests are tallied at 64), not measured by this run"
                        % (rnd, name, t.get("commit", "unknown")))
        except (OSError, ValueError, KeyError):
            pass
    return None, None, None


def usable_cores():
    """host cores this process may really use: the affinity mask, cut by the container's CPU quota (cgroup cpu.max).
    A box that shows 256 CPUs under a 16-CPU quota runs 256 threads slower than 16 (they burn the quota and are all
    throttled for the rest of the period: measured, profiles/r3/threads_quota.log)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def corpus():
    from tests import synth

    return synth.bench_corpus()


_C = None
_LEVEL = 6
_TABLE_CACHE = None


def _compress_one(off_size):
    off, size = off_size
    d = _C[off:off + size]
    co = zlib.compressobj(_LEVEL, zlib.DEFLATED, -15, 8)  # == deflateInit2(level, Z_DEFLATED, -15, 8, default)
    return co.compress(d) + co.flush(), zlib.crc32(d)


def _pool_init(c, level):
    global _C, _LEVEL
    _C, _LEVEL = c, level


def shared_unique_deflate(c, n_unique, size, seed, gen_seconds, world, local_rank, tag):
    """make_unique_deflate() once per NODE when several ranks build the SAME table (strong scaling: same seed on every rank):
    the rank with LOCAL_RANK 0 compresses with every host core and leaves the table in a file of the temp directory, the
    others wait for it and load it -- instead of N ranks compressing the same 6 GiB at 1/N of the cores each, N times over
    (keep the --gpus 2 / 4 / 8 lines cheap for the driver)."""
    global _TABLE_CACHE
    path = os.path.join(tempfile.gettempdir(), "mzhip_bench_%s_%d_%d_%d_%08x_%d.npz" % (tag, n_unique, size, seed, zlib.crc32(bytes(c[:1 << 20])) & 0xFFFFFFFF, len(c)))
    if local_rank == 0:
        _TABLE_CACHE = path  # removed by this rank when every rank has reported (main)
        if os.path.exists(path):
            os.remove(path)  # (a run that died: the others wait for THIS run's file... they may have loaded the old one, which holds the same streams)
        offs, pays, crcs = make_unique_deflate(c, n_unique, size, seed, gen_seconds, 1)
        lens = np.array([len(p) for p in pays], dtype=np.int64)
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, offs=np.array([o for o, _ in offs], dtype=np.int64), lens=lens, crcs=crcs, blob=np.frombuffer(b"".join(pays), dtype=np.uint8))
        os.replace(tmp, path)
        return offs, pays, crcs
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 4 * gen_seconds + 600:
            fail("rank with LOCAL_RANK %d: the table of LOCAL_RANK 0 (%s) did not appear" % (local_rank, path))
        time.sleep(0.5)
    z = np.load(path)
    ends = np.cumsum(z["lens"])
    blob = z["blob"].tobytes()
    pays = [blob[int(e - l):int(e)] for e, l in zip(ends, z["lens"])]
    return [(int(o), size) for o in z["offs"]], pays, z["crcs"]


def make_unique_deflate(c, n_unique, size, seed, gen_seconds, world):
    """n_unique level-6 streams of distinct slices of the corpus, on every host core this rank may use.  Stops early
    (at least 256 streams) only when gen_seconds runs out: the caller then tiles and says so."""
    rnd = random.Random(seed)
    seen, offs = set(), []
    span = len(c) - size
    while len(offs) < n_unique:  # distinct offsets while the corpus has them
        o = rnd.randrange(span)
        if o in seen and len(seen) < span:
            continue
        seen.add(o)
        offs.append((o, size))
    procs = max(1, usable_cores() // max(world, 1))  # ranks share the host cores
    t0 = time.time()
    out = []
    # (close + join, not the context manager's terminate: under rocprofv3 a SIGTERM to the forked workers runs the tool's
    # signal handler in each of them, and that has left a --pmc pass hanging until its timeout twice, profiles/r4/README.md)
    pool = mp.Pool(procs, initializer=_pool_init, initargs=(c, 6))
    try:
        for r in pool.imap(_compress_one, offs, chunksize=64):
            out.append(r)
            if time.time() - t0 > gen_seconds and len(out) >= 256:
                pool.terminate()
                break
        else:
            pool.close()
    finally:
        pool.join()
    offs = offs[:len(out)]
    return offs, [p for p, _ in out], np.array([k for _, k in out], dtype=np.uint32)


def _lzma_one(d):
    import lzma

    # what the reference writer emits for method 14 (mz_strm_lzma.c:94-104,250-265, mz_zip.c:1984): the 4-byte ZIP-LZMA
    # header (version 5.2, properties size 5), the 5 properties bytes, the raw LZMA1 stream of preset 6 with the end marker
    raw = lzma.compress(d, format=lzma.FORMAT_ALONE, filters=[dict(id=lzma.FILTER_LZMA1, preset=6)])
    assert raw[5:13] == b"\xff" * 8  # .lzma alone = props(5) + size(8, unknown) + stream + end marker
    return bytes([5, 2, 5, 0]) + raw[:5] + raw[13:], zlib.crc32(d)


def _markov_lzma_one(args):
    """one config-4 entry, made and compressed in the same worker (1 MiB crosses the process boundary once)"""
    from tests import synth

    size, seed = args
    d = synth.markov_entries(1, size, seed, _C)[0]
    p, k = _lzma_one(d)
    return d, p, k


def make_markov_lzma(c, n_unique, size, seed, gen_seconds, world):
    """n_unique order-2 word-Markov expansions of the corpus, one seed each (pure Python: ~0.3 s per MiB), each compressed by
    liblzma at preset 6 (~0.6 s per MiB), on every host core this rank may use.  Stops early (at least 64 streams) only
    when gen_seconds runs out: the caller then tiles and says so."""
    procs = max(1, min(usable_cores() // max(world, 1), n_unique))
    t0 = time.time()
    datas, pays, crcs = [], [], []
    pool = mp.Pool(procs, initializer=_pool_init, initargs=(c, 6))
    try:
        for d, p, k in pool.imap(_markov_lzma_one, [(size, seed * 100003 + i) for i in range(n_unique)]):
            datas.append(d)
            pays.append(p)
            crcs.append(k)
            if time.time() - t0 > gen_seconds and len(datas) >= 64:
                pool.terminate()
                break
        else:
            pool.close()
    finally:
        pool.join()
    return datas, pays, np.array(crcs, dtype=np.uint32)


def device_blob(torch, dev, pays, pick, align=16):
    """every entry gets its own copy of its bytes: gather 16-byte granules on the host, one H2D copy"""
    plen = np.array([len(p) for p in pays], dtype=np.int64)
    in_len = plen[pick]
    n = len(pick)
    pad = (in_len + align - 1) // align * align
    in_off = np.zeros(n, dtype=np.int64)
    np.cumsum(pad[:-1], out=in_off[1:])
    total_in = int(in_off[-1] + pad[-1]) if n else align
    upad = (plen + align - 1) // align * align
    uoff = np.zeros(len(pays), dtype=np.int64)
    np.cumsum(upad[:-1], out=uoff[1:])
    ublob = np.zeros(int(uoff[-1] + upad[-1]), dtype=np.uint8)
    for i, p in enumerate(pays):
        ublob[uoff[i]:uoff[i] + len(p)] = np.frombuffer(p, dtype=np.uint8)
    gran = (pad // 16).astype(np.int64)
    gstart = np.concatenate(([0], np.cumsum(gran)[:-1]))
    u16 = ublob.view(np.dtype((np.void, 16)))
    h_in = np.empty(total_in // 16, dtype=u16.dtype)
    CH = 8192
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        g0, g1 = int(gstart[lo]), int(gstart[hi - 1] + gran[hi - 1])
        src = np.repeat(uoff[pick[lo:hi]] // 16 - gstart[lo:hi], gran[lo:hi]) + np.arange(g0, g1)
        h_in[g0:g1] = u16[src]
    h_u8 = h_in.view(np.uint8)
    d_in = torch.from_numpy(h_u8).to(dev)
    return d_in, h_u8, in_off, in_len


# ---------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_inflate(c, offs, size, pays, want_crc, cores, keep_path=None):
    """The unmodified reference reader on the host cores (oracle/_ref): mz_zip_entry_read -> mz_stream_zlib_read ->
    zlib inflate + mz_crypt_crc32_update + CRC verify, one reader handle per thread."""
    import oracle

    if not oracle.have_ref():
        return None
    ref = oracle.ref()
    n = min(len(offs), 2048 if size >= 65536 else 16384)
    blob = np.frombuffer(c, dtype=np.uint8)
    o = np.array([x[0] for x in offs[:n]], dtype=np.int64)
    ln = np.full(n, size, dtype=np.int32)
    tmp = tempfile.mkdtemp(prefix="mzhip_bench_")
    path = keep_path or os.path.join(tmp, "sample.zip")
    ref.zip_write(path, blob, o, ln, method=8, level=6)  # the reference writer itself (mz_zip_rw.c:1546)
    table = ref.zip_index(path)
    raw = open(path, "rb").read()
    same = all(raw[int(table[i, 7]):int(table[i, 7] + table[i, 3])] == pays[i] for i in range(0, n, max(1, n // 64)))
    cd = table[:, 6].copy()
    rates = {}
    for mapped in (False, True):  # the reader's own open_file (split stream: re-opens the file per entry) / readers on one shared mapping
        passes, total_s = 0, 0.0
        while total_s < 3.0 and passes < 400:
            sec, crc, ulen, st = ref.zip_read_all(path, cd, nthreads=cores, own_crc=False, mapped=mapped)
            assert (st == 0).all() and (ulen == size).all() and (crc == want_crc[:n]).all()
            passes += 1
            total_s += sec
        rates[mapped] = (n * size / 2**30 * passes / total_s, passes)
    k = min(n, 256)
    sec1, _, _, _ = ref.zip_read_all(path, cd[:k], nthreads=1, own_crc=False)
    if not keep_path:
        os.remove(path)
        os.rmdir(tmp)
    best = max(rates.values())[0]
    return dict(value=round(best, 4), unit="GiB/s", cores=cores, kind="reference",
                sample="%d x %d B DEFLATE-6 entries written by the reference writer (payload bytes %s the bench's streams), "
                       "mz_zip_entry_read (zlib 1.2.11 inflate + crc32 + CRC verify) with %d threads, one reader handle each: "
                       "%.3f GiB/s with mz_zip_reader_open_file (%d passes; its split stream re-opens the file twice per entry), "
                       "%.3f GiB/s with the readers on mz_stream_mem over one shared mapping (%d passes); value = the better; "
                       "1-thread: %.3f GiB/s.  DEVIATION from BASELINE.md 3 (whole archive, median of 3): this is a SAMPLE of the "
                       "archive read repeatedly for ~3 s per mode (it stays in the page cache and the last-level cache), best of "
                       "two ways to open it -- both choices favour the CPU.  The codec under it is zlib 1.2.11, the slowest inflate the reference "
                       "builds on: its CMakeLists prefers zlib-ng (absent here, no network), whose inflate is about 2 x this" % (n, size, "identical to" if same else "DIFFER from", cores, rates[False][0],
                                                 rates[False][1], rates[True][0], rates[True][1], k * size / 2**30 / sec1))


def cpu_baseline_lzma(datas, cores):
    """The reference LZMA reader (mz_stream_lzma_read -> liblzma 5.2.5).  Its encoder runs at ~2 MB/s, so 8 entries are
    written by the reference writer and their payloads replicated into a 2 x cores-entry archive."""
    import oracle

    if not oracle.have_ref():
        return None
    ref = oracle.ref()
    size = len(datas[0])
    with tempfile.TemporaryDirectory() as tmp:
        k = min(8, len(datas))
        blob = np.frombuffer(b"".join(datas[:k]), dtype=np.uint8)
        small = os.path.join(tmp, "s.zip")
        ref.zip_write(small, blob, np.arange(k, dtype=np.int64) * size, np.full(k, size, dtype=np.int32), method=14, level=6)
        st = ref.zip_index(small)
        raw = open(small, "rb").read()
        n = max(2 * cores, 64)
        path = os.path.join(tmp, "l.zip")
        with open(path, "wb") as f:
            cd = []
            for i in range(n):
                m, flag, crc, cs, us, _, _, po = (int(v) for v in st[i % k])
                name = b"e/%06d" % i
                f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 63, flag & ~8, m, 0, 0x21, crc & 0xFFFFFFFF, cs, us, len(name), 0))
                cd.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0x033F, 63, flag & ~8, m, 0, 0x21, crc & 0xFFFFFFFF, cs,
                                      us, len(name), 0, 0, 0, 0, 0, f.tell() - 30) + name)
                f.write(name + raw[po:po + cs])
            cd_off = f.tell()
            f.write(b"".join(cd))
            f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, n, n, f.tell() - cd_off, cd_off, 0))
        tab = ref.zip_index(path)
        sec, crc, ulen, stt = ref.zip_read_all(path, tab[:, 6].copy(), nthreads=cores, own_crc=False)
        assert (stt == 0).all()
        sec1, _, ulen1, _ = ref.zip_read_all(path, tab[:4, 6].copy(), nthreads=1, own_crc=False)
        return dict(value=round(float(ulen.sum()) / 2**30 / sec, 4), unit="GiB/s", cores=cores, kind="reference",
                    sample="%d x %d B method-14 entries (8 written by the reference writer, preset 6, ratio %.3f, replicated), "
                           "one pass of mz_zip_entry_read (liblzma 5.2.5 + crc32 + CRC verify) with %d threads; 1-thread: %.3f GiB/s"
                           % (n, size, tab[:, 3].sum() / tab[:, 4].sum(), cores, float(ulen1.sum()) / 2**30 / sec1))


def cpu_baseline_deflate(c, offs, size, cores):
    """The reference writer at level 1 (mz_zip_writer_add_buffer -> mz_stream_zlib_write -> zlib deflate + crc32),
    one archive per thread."""
    import threading

    import oracle

    if not oracle.have_ref():
        return None
    ref = oracle.ref()
    per = 1000
    blob = np.frombuffer(c, dtype=np.uint8)
    o = np.array([offs[i % len(offs)][0] for i in range(per)], dtype=np.int64)
    ln = np.full(per, size, dtype=np.int32)
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for th in (1, cores):
            ts = [threading.Thread(target=ref.zip_write, args=(os.path.join(tmp, "w%d.zip" % i), blob, o, ln, 8, 1)) for i in range(th)]
            t0 = time.time()
            [t.start() for t in ts]
            [t.join() for t in ts]
            res[th] = per * th * size / 2**30 / (time.time() - t0)
            ratio = os.path.getsize(os.path.join(tmp, "w0.zip")) / (per * size)
    return dict(value=round(res[cores], 4), unit="GiB/s", cores=cores, kind="reference",
                sample="%d x %d B buffers per thread through mz_zip_writer_add_buffer at level 1 (zlib 1.2.11 deflate + crc32, "
                       "archive ratio %.3f), %d threads = %d archives; 1-thread: %.3f GiB/s" % (per, size, ratio, cores, cores, res[1]))


def write_stream_zip(path, pays, crcs, size, method=8):
    """A plain ZIP archive (no data descriptors) around streams that exist already: what the reference writer lays down for
    them (mz_zip.c:1236-1438 local header, :1440-1600 central record); with more than 65 535 entries the ZIP64 end record and
    its locator in front of the end record (mz_zip.c:1139-1190), as the reference writes them.  Offsets stay below 4 GiB."""
    with open(path, "wb", buffering=1 << 22) as f:
        cd = []
        pos = 0
        for i, p in enumerate(pays):
            name = b"e/%06d" % i
            crc = int(crcs[i]) & 0xFFFFFFFF
            f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, method, 0, 0x21, crc, len(p), size, len(name), 0))
            cd.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0x0314, 20, 0, method, 0, 0x21, crc, len(p), size,
                                  len(name), 0, 0, 0, 0, 0, pos) + name)
            f.write(name)
            f.write(p)
            pos += 30 + len(name) + len(p)
        assert pos < (1 << 32)
        cd_off = pos
        cdb = b"".join(cd)
        f.write(cdb)
        n = len(pays)
        if n > 0xFFFF:
            e64 = cd_off + len(cdb)
            f.write(struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, n, n, len(cdb), cd_off))
            f.write(struct.pack("<IIQI", 0x07064B50, 0, e64, 1))
        f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, min(n, 0xFFFF), min(n, 0xFFFF), len(cdb), cd_off, 0))


# ---------------------------------------------------------------------------------------------- config-2 legs
def legs_config2(torch, mz, dev, h_in, in_off, in_len, size, want_crc_np, kernel_gib, sample_zip):
    """SURVEY 8(d) (i)-(iii).  (ii) and (iii) run on bounded samples; all figures are decompressed GiB/s.
      kernel               the timed region of this bench (inputs and outputs resident in HBM)
      h2d_kernel_d2hcrc    pinned host memory -> device, decode, {crc, len, status} back: chunks of the entry table, copies on
                           one stream and launches on another, so that H2D(i + 1) runs under kernel(i)
      vtbl_end_to_end      mzhip_prime_mem_begin (index, then pipelined H2D / launches / D2H of every decoded byte on a worker
                           thread) with the reference's unmodified reader loop on the drop-in library running under it, ONE
                           reader thread
      vtbl_end_to_end_T    the same with T = all host cores reader threads, one mz_zip_reader each
                           (integration/extract_threads.c: the shape of the cpu_baseline leg)
      vtbl_unprimed_default  the application is only re-linked: the unmodified reader loop on the drop-in with NO prime call
                           and nothing in the environment, one thread -- the first read() images the archive through the
                           reader's own stream and primes it (shim_autoprime.c, on by default since round 5); beside it
                           vtbl_unprimed_per_entry, the same loop with MZHIP_AUTOPRIME=0 (a launch and a PCIe round trip per
                           entry, what rounds 1 - 4 did by default), on a quarter of the sample"""
    out = {"kernel": round(kernel_gib, 2)}
    # chunks of one launch round each: a launch keeps 4096 waves resident (16 per CU) and a wave decodes one entry, so a
    # chunk of 5000 entries would pay a second, 22 % full round
    per = 4096
    n = min(len(in_len), 5 * per)
    end = int(in_off[n - 1] + (in_len[n - 1] + 15) // 16 * 16)
    hp = torch.from_numpy(h_in[:end]).pin_memory()
    d_in = torch.empty(end, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(in_off[:n]).to(dev)
    d_len = torch.from_numpy(in_len[:n].astype(np.int32)).to(dev)
    d_out = torch.empty(n * size, dtype=torch.uint8, device=dev)
    d_oo = torch.arange(n, dtype=torch.int64, device=dev) * size
    d_oc = torch.full((n,), size, dtype=torch.int32, device=dev)
    nchunk = (n + per - 1) // per
    cuts = [min(n, per * i) for i in range(nchunk + 1)]
    h_parts = [torch.empty((3, cuts[i + 1] - cuts[i]), dtype=torch.int32).pin_memory() for i in range(nchunk)]  # contiguous targets
    # one stream copies, in order, so that chunk i is complete as early as the link allows; one computes (chunk i's launch
    # waits for chunk i's copy only); the results go back on a third
    s_copy, s_comp, s_back = (torch.cuda.Stream(device=dev) for _ in range(3))
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = []
        for ci in range(nchunk):
            lo, hi = cuts[ci], cuts[ci + 1]
            b0, b1 = int(in_off[lo]), (int(in_off[hi]) if hi < n else end)
            with torch.cuda.stream(s_copy):
                d_in[b0:b1].copy_(hp[b0:b1], non_blocking=True)
                arrived = s_copy.record_event()
            with torch.cuda.stream(s_comp):
                s_comp.wait_event(arrived)
                out_len, in_used, crc, status = mz.inflate_batch(d_in, d_off[lo:hi], d_len[lo:hi], d_out, d_oo[lo:hi], d_oc[lo:hi])
                res = torch.stack((crc, out_len, status))
                decoded = s_comp.record_event()
            with torch.cuda.stream(s_back):
                s_back.wait_event(decoded)
                h_parts[ci].copy_(res, non_blocking=True)
            keep.append(res)  # (allocated on s_comp, read on s_back: alive until the synchronize below)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    h_res = torch.cat(h_parts, dim=1)
    ok = bool((h_res[2].numpy() == 0).all() and (h_res[0].numpy().view(np.uint32) == want_crc_np[:n]).all())
    out["h2d_kernel_d2hcrc"] = round(n * size / 2**30 / best, 2) if ok else None
    out["h2d_kernel_d2hcrc_sample"] = "%d entries in %d chunks of one launch round: a copy stream, a compute stream, a stream for the results; pinned host memory next to the device, best of 3" % (n, nchunk)
    # (iii) the reference's unmodified reader loop on the drop-in library, after mzhip_prime_file
    drop = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    out["vtbl_end_to_end"] = out["vtbl_end_to_end_T"] = None
    if sample_zip and os.path.exists(sample_zip) and os.path.exists(drop):
        D = C.CDLL(drop)
        L = mz.lib()
        if hasattr(D, "mzdrop_extract_all"):
            D.mzdrop_extract_all.restype = C.c_double
            D.mzdrop_extract_all.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                             C.POINTER(C.c_double), C.POINTER(C.c_int32)]
            cores = max(1, usable_cores() // 2)  # the prime worker feeds the copy engines and the HIP runtime has threads of its own: past half the quota more readers slow the pipeline down (profiles/r3/threads_trace.log)
            for key, T in (("vtbl_end_to_end", 1), ("vtbl_end_to_end_T", cores)):
                best, desc = None, ""
                for _ in range(3):
                    L.mzhip_prime_clear()
                    ne, nb, tp, fe = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_int32(0)
                    sec = D.mzdrop_extract_all(sample_zip.encode(), T, 2, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe))
                    if sec > 0 and fe.value == 0 and ne.value > 0 and (best is None or sec < best):
                        best = sec
                        desc = ("%d entries / %.0f MiB: one read-only mapping of the archive, mzhip_prime_mem_begin over it (index, then "
                                "pipelined H2D, launches, D2H of every byte on a worker thread) with %d reader thread(s) running under "
                                "it front to back (the calling thread spent %.0f ms in begin + wait), one mz_zip_reader each on "
                                "mz_stream_mem over the mapping, mz_zip_entry_read in 65 535-byte calls + CRC verification "
                                "(mz_zip.c:2116-2128) on libmzhipdrop.so; best of 3"
                                % (ne.value, nb.value / 2**20, T, tp.value * 1e3))
                        nbytes = nb.value
                L.mzhip_prime_clear()
                if best:
                    out[key] = round(nbytes / 2**30 / best, 3)
                    out[key + "_sample"] = desc
            # the re-linked application: no prime call, nothing in the environment (prime mode 0 of the same driver)
            had = os.environ.pop("MZHIP_AUTOPRIME", None)
            try:
                L.mzhip_autoprime_count.restype = C.c_uint64
                best = None
                for _ in range(3):
                    L.mzhip_prime_clear()
                    a0 = L.mzhip_autoprime_count()
                    ne, nb, tp, fe = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_int32(0)
                    sec = D.mzdrop_extract_all(sample_zip.encode(), 1, 0, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe))
                    if sec > 0 and fe.value == 0 and ne.value > 0 and L.mzhip_autoprime_count() == a0 + 1 and (best is None or sec < best):
                        best, nbytes, nent = sec, nb.value, ne.value
                L.mzhip_prime_clear()
                if best:
                    out["vtbl_unprimed_default"] = round(nbytes / 2**30 / best, 3)
                    out["vtbl_unprimed_default_sample"] = ("%d entries / %.0f MiB: the unmodified mz_zip_reader loop on libmzhipdrop.so, ONE thread, no "
                                                           "mzhip_prime_* call, no environment variable: the first read() images the archive through the reader's own "
                                                           "stream and primes it (shim_autoprime.c), mz_zip_entry_read in 65 535-byte calls + CRC verification; "
                                                           "map + index + imaging + decode + every read inside the clock; best of 3" % (nent, nbytes / 2**20))
            finally:
                if had is not None:
                    os.environ["MZHIP_AUTOPRIME"] = had
    return out


def full_archive_legs(mz, path, n, size, want_crc_np, cores, with_reference):
    """The WHOLE config-2 archive (every entry of the bench's table around the bench's own streams, ZIP64 end records: what
    BASELINE.json configs[1] describes, ~2 GiB, 6.1 GiB decoded) through the unmodified reader loop, nothing but re-linked:
      vtbl_unprimed_cfg2_archive     mz_zip_reader_open_file (split / buffered / OS streams), goto_first / goto_next, ONE thread,
                                     no mzhip_prime_* call, no environment: the archive is far over the auto-prime's whole-image
                                     limit, so it is rolled over window by window ahead of the reader (shim_autoprime.c, round 6);
                                     peak page-locked bytes of the windows beside it
      vtbl_unprimed_cfg2_archive_T   the same with T reader threads, a contiguous share of the entries each
      vtbl_unprimed_cfg2_mapped      one thread, the reader on mz_stream_mem over a mapping (mzdrop_extract_all without a prime; the
                                     windows are imaged through the readers' memory streams); _mapped_T: T such readers
    and, with_reference, the CPU baseline of record as BASELINE.md 3 defines it: the reference's reader over the same file,
    whole archive, cores threads, median of 3 (+ one thread on a slice)."""
    out, cb = {}, None
    drop = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not os.path.exists(drop):
        return out, cb
    D = C.CDLL(drop)
    L = mz.lib()
    if not hasattr(D, "mzdrop_extract_file"):
        return out, cb
    D.mzdrop_extract_file.restype = C.c_double
    D.mzdrop_extract_file.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    D.mzdrop_extract_all.restype = C.c_double
    D.mzdrop_extract_all.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int32)]
    L.mzhip_autoprime_stats.argtypes = [C.POINTER(C.c_uint64)] * 4
    L.mzhip_autoprime_count.restype = C.c_uint64
    fsize = os.path.getsize(path)
    what = "%d entries, archive %.2f GiB, %.2f GiB decoded" % (n, fsize / 2**30, n * size / 2**30)
    had = os.environ.pop("MZHIP_AUTOPRIME", None)
    try:
        for key, T, mapped in (("vtbl_unprimed_cfg2_archive", 1, False), ("vtbl_unprimed_cfg2_archive_T", max(2, min(8, cores // 2)), False),
                               ("vtbl_unprimed_cfg2_mapped", 1, True), ("vtbl_unprimed_cfg2_mapped_T", max(2, min(8, cores // 2)), True)):
            if mapped and fsize > 0x7FFFFFFF:
                continue
            best, info = None, ""
            for _ in range(3):
                L.mzhip_prime_clear()
                a0 = L.mzhip_autoprime_count()
                w0 = [C.c_uint64() for _ in range(4)]
                L.mzhip_autoprime_stats(*[C.byref(x) for x in w0])
                ne, nb, fe, tp = C.c_int64(0), C.c_int64(0), C.c_int32(0), C.c_double(0)
                if mapped:
                    sec = D.mzdrop_extract_all(path.encode(), T, 0, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe))
                else:
                    sec = D.mzdrop_extract_file(path.encode(), T, C.byref(ne), C.byref(nb), C.byref(fe))
                w1 = [C.c_uint64() for _ in range(4)]
                L.mzhip_autoprime_stats(*[C.byref(x) for x in w1])
                # (rolled over by the library itself: windows were primed during the call -- the archive's index is made once per process)
                ok = sec > 0 and fe.value == 0 and ne.value == n and nb.value == n * size and w1[0].value > w0[0].value and L.mzhip_autoprime_count() >= max(a0, 1)
                if ok and (best is None or sec < best):
                    best = sec
                    info = "%d windows primed, %d evicted, at most %.0f MiB of decoded bytes held at once" % (
                        w1[0].value - w0[0].value, w1[1].value - w0[1].value, w1[3].value / 2**20)
            L.mzhip_prime_clear()
            out[key] = round(n * size / 2**30 / best, 3) if best else None
            out[key + "_sample"] = ("%s: the unmodified reader loop on libmzhipdrop.so (%s), %d reader thread(s), no mzhip_prime_* call, no "
                                    "environment variable, mz_zip_entry_read in 65 535-byte calls + CRC verification of every entry; open + "
                                    "central directory + every window's imaging and decode + every read inside the clock; best of 3; %s"
                                    % (what, "mz_stream_mem over one mapping" if mapped else "mz_zip_reader_open_file", T, info))
    finally:
        if had is not None:
            os.environ["MZHIP_AUTOPRIME"] = had
    if with_reference:
        import oracle

        if oracle.have_ref():
            ref = oracle.ref()
            table = ref.zip_index(path)
            cd = table[:, 6].copy()
            res = {}
            for mapped in ((False, True) if fsize <= 0x7FFFFFFF else (False,)):
                secs = []
                for _ in range(3):
                    sec, crc, ulen, st = ref.zip_read_all(path, cd, nthreads=cores, own_crc=False, mapped=mapped)
                    if not ((st == 0).all() and (ulen == size).all() and (crc == want_crc_np[:n]).all()):
                        secs = None
                        break
                    secs.append(sec)
                if secs:
                    res[mapped] = sorted(secs)[1]
            k = min(n, 4096)
            sec1, _, _, _ = ref.zip_read_all(path, cd[:k], nthreads=1, own_crc=False)
            if res:
                med = min(res.values())
                cb = dict(value=round(n * size / 2**30 / med, 4), unit="GiB/s", cores=cores, kind="reference",
                          sample="BASELINE.md 3: the WHOLE archive (%s; the bench's own level-6 streams, byte-identical to the reference writer's on "
                                 "the sample), the reference's reader (mz_zip_entry_read: zlib 1.2.11 inflate + crc32 + CRC verify) with %d threads, "
                                 "one reader handle and a contiguous share each, page cache warm, median of 3: %s; value = the better way to open "
                                 "it; one thread on %d entries: %.3f GiB/s" % (
                                     what, cores, ", ".join("%.3f GiB/s with %s" % (n * size / 2**30 / v, "mz_stream_mem over one mapping" if m else
                                                                                   "mz_zip_reader_open_file") for m, v in sorted(res.items())),
                                     k, k * size / 2**30 / sec1))
    return out, cb


def large_entry_leg(c, mib, with_reference):
    """ONE large DEFLATE entry (mib MiB of the corpus, zlib level 1, ZIP64) through the unmodified mz_zip_reader on the drop-in:
    per entry (the READ stream's window mode: a wave per DEFLATE block) and under the prime (the same
    kernels where the entry lies in HBM).  -> (legs dict, cpu_baseline dict or None): decompressed GiB/s, one reader thread."""
    import zipfile

    drop = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    if not os.path.exists(drop):
        return {}, None
    D = C.CDLL(drop)
    if not hasattr(D, "mzdrop_extract_all"):
        return {}, None
    D.mzdrop_extract_all.restype = C.c_double
    D.mzdrop_extract_all.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int32)]
    tmp = tempfile.mkdtemp(prefix="mzhip_large_")
    path = os.path.join(tmp, "large.zip")
    total, piece = mib << 20, bytes(c) * 2
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as zf:
        with zf.open("large.bin", "w", force_zip64=True) as f:
            left = total
            while left > 0:
                k = min(left, len(piece))
                f.write(piece[:k])
                left -= k
    legs, cb = {}, None
    what = "one %d MiB entry (the corpus repeated, zlib level 1, ratio %.2f)" % (mib, os.path.getsize(path) / total)
    for key, prime, how in (("one_large_entry_vtbl", 0, "per entry: mz_stream_zlib READ in window mode, every window by a wave per DEFLATE block"),
                            ("one_large_entry_primed", 2, "under mzhip_prime_mem_begin: the entry decoded where it lies in HBM (mzhip_inflate_large), every byte copied back")):
        best = None
        for _ in range(3):
            ne, nb, tp, fe = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_int32(0)
            sec = D.mzdrop_extract_all(path.encode(), 1, prime, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe))
            if sec > 0 and fe.value == 0 and nb.value == total and (best is None or sec < best):
                best = sec
        legs[key] = round(total / 2**30 / best, 3) if best else None
        legs[key + "_sample"] = "%s through the unmodified mz_zip_reader on libmzhipdrop.so, one reader thread, CRC verified, %s; best of 3" % (what, how)
    ref = None
    if with_reference:
        import oracle

        if oracle.have_ref():
            ref = oracle.ref()
            table = ref.zip_index(path)
            sec, _, ulen, st = ref.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=False)
            if (st == 0).all() and int(ulen[0]) == total:
                cb = dict(value=round(total / 2**30 / sec, 3), unit="GiB/s", cores=1, kind="reference",
                          sample=what + ": mz_zip_entry_read (zlib 1.2.11 inflate + crc32 + CRC verify) on one thread -- one entry is one inflate() state")
    # ... and WRITTEN: the unmodified mz_zip writer on the drop-in, 65 535 bytes per mz_zip_entry_write (mz_driver.c), level 1
    if hasattr(D, "drv_zip_write_repeat"):
        D.drv_zip_write_repeat.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int64]
        pc = np.frombuffer(piece, dtype=np.uint8)
        wpath = os.path.join(tmp, "written.zip")
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            err = D.drv_zip_write_repeat(wpath.encode(), 8, 1, pc.ctypes.data, pc.size, total)
            sec = time.perf_counter() - t0
            if err == 0 and (best is None or sec < best):
                best = sec
        ok = best is not None
        if ok and ref is not None:  # the all-reference reader inflates what was written and verifies its CRC
            table = ref.zip_index(wpath)
            _, _, ulen, st = ref.zip_read_all(wpath, table[:, 6].copy(), nthreads=1, own_crc=False)
            ok = bool((st == 0).all() and int(ulen[0]) == total)
        legs["one_large_entry_vtbl_write"] = round(total / 2**30 / best, 3) if ok else None
        legs["one_large_entry_vtbl_write_sample"] = ("one %d MiB entry written through the unmodified mz_zip writer on libmzhipdrop.so in 65 535-byte calls, level 1 "
                                                     "(mz_stream_zlib WRITE: 8 MiB segments, one coded while the next is collected; mz_crypt_crc32_update per call), "
                                                     "archive %.3f of the input%s; best of 2" % (mib, os.path.getsize(wpath) / total if os.path.exists(wpath) else 0.0,
                                                                                                ", read back and CRC-verified by the all-reference reader" if ref is not None else ""))
        if ref is not None:
            rpath, rtotal = os.path.join(tmp, "written_ref.zip"), total // 8
            t0 = time.perf_counter()
            ref.zip_write_repeat(rpath, pc, rtotal, method=8, level=1)
            legs["one_large_entry_reference_write"] = round(rtotal / 2**30 / (time.perf_counter() - t0), 3)
            legs["one_large_entry_reference_write_sample"] = "the all-reference writer (zlib 1.2.11 deflate level 1 + crc32), one thread, %d MiB of the same entry" % (rtotal >> 20)
            os.remove(rpath)
        if os.path.exists(wpath):
            os.remove(wpath)
    os.remove(path)
    os.rmdir(tmp)
    return legs, cb


def other_configs(args):
    """`python bench.py --config 3|4|5` (one GPU) as child processes of the default run, condensed: the driver sees the four
    configurations of BASELINE.json that run on a GPU in ONE line.  A step of config 4 takes a second, so the children run
    at most 3 timed steps after 1 warm-up step."""
    import subprocess

    out = {}
    for k in (3, 4, 5):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(k), "--gpus", "1", "--steps", str(min(args.steps, 3)),
               "--warmup", "1", "--no-legs", "--gen-seconds", str(args.gen_seconds)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001 -- a child that failed must not take the headline with it
            out[str(k)] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            continue
        if "error" in j:
            out[str(k)] = j
            continue
        rf = j["roofline"]
        out[str(k)] = {
            "metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
            "warmup": j["warmup"], "crc32_match_rate": j["crc32_match_rate"], "bytes_spot_check": j["bytes_spot_check"],
            "data": j["data"], "unique_streams": j["config"].get("unique_streams"), "workload": j["config"]["workload"],
            "roofline": {x: rf.get(x) for x in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_raw", "traffic_source", "kernel",
                                            "kernel_ms", "algorithmic_bytes_per_launch")},
            "cpu_baseline": j.get("cpu_baseline"), "wall_s": round(time.time() - t0, 1)}
    return out


def fail(msg):
    """one JSON line and a non-zero exit: a run that cannot be the run that was asked for must not print a metric"""
    print(json.dumps({"error": msg}), flush=True)
    sys.exit(2)


def _visible_gpus():
    """GPUs this process could use, without creating a HIP context in the launcher process"""
    import subprocess

    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def self_launch(n):
    """`python bench.py --gpus N` with no launcher: run N ranks of this script under torch.distributed.run (what the
    driver's own command line does), one per GPU; exit non-zero with an {"error"} line when the box cannot."""
    import socket
    import subprocess

    have = _visible_gpus()
    if have < n and os.environ.get("MZHIP_BENCH_SHARE_GPU") != "1":
        fail("--gpus %d but only %d GPU(s) are visible on this node; not measuring fewer GPUs than asked for" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    # the reference's header parser turns every DOS date into a time_t with mktime() (mz_zip.c: mz_zip_dosdate_to_time_t),
    # and glibc's mktime stats /etc/localtime under a process-wide lock when TZ is unset: 16 reader threads then spend 95 %
    # of their time queueing there (125 ms -> 27 ms for 16 readers of a 1 GiB archive; profiles/r3/threads_mapped_tzunset.log is the unset case).  Both the reference baseline and the
    # drop-in legs run with a fixed zone; it changes no byte of what they read.
    os.environ.setdefault("TZ", "UTC")
    time.tzset()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--entries", type=int, default=0, help="entries of the table (strong) / per GPU (weak); 0 = the config's")
    ap.add_argument("--entry-size", type=int, default=0)
    ap.add_argument("--unique", type=int, default=0, help="max unique compressed slices to generate")
    ap.add_argument("--gen-seconds", type=float, default=45.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="config 2, N = 1: do not run configs 3, 4, 5 behind the headline")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.gpus < 1:
        fail("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # no launcher around us: become one (one rank per GPU)

    import torch

    mz = importlib.import_module("minizip-ng_amd")
    archive = importlib.import_module("minizip-ng_amd.archive")
    from tests import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (a launcher with another world size, or --gpus 1 under a launcher)
        fail("WORLD_SIZE (%d) != --gpus (%d): start one rank per GPU" % (world, args.gpus))
    mz.require_gpu()  # no CPU fallback: fail loudly if the HIP path is unavailable
    share_gpu = os.environ.get("MZHIP_BENCH_SHARE_GPU") == "1"
    if not share_gpu and torch.cuda.device_count() < world:
        fail("--gpus %d but this rank sees %d device(s)" % (world, torch.cuda.device_count()))
    size = args.entry_size or cfg["size"]
    n_table = args.entries or cfg["entries"]
    n_unique = args.unique or cfg["unique"]
    strong = args.scaling == "strong"
    c, cdesc = corpus()
    if cfg.get("extend_mib") and not args.unique:
        # config 3 wants half a million UNIQUE 8 KiB streams and the corpus has 463 000 distinct 8 KiB slices: slices are taken
        # from the corpus followed by 4 KiB pieces of it in a seeded random order (an 8 KiB slice of that part is natural text
        # with one or two seams: level 6 makes 0.367 of it where a plain slice gives 0.347)
        ernd = random.Random(777)
        cb = bytes(c)
        ext = b"".join(cb[p:p + 4096] for p in (ernd.randrange(len(cb) - 4096) for _ in range(cfg["extend_mib"] * 256)))
        c = cb + ext
        cdesc += " followed by %d MiB of 4 KiB pieces of it in a seeded random order" % cfg["extend_mib"]
    # the worker pool that compresses the synthetic slices forks: do it before this process creates its HIP context
    # and the RCCL threads.  Strong scaling: every rank derives the SAME table (same seeds); weak: its own.
    seed = 1234 if strong else 1234 + rank
    offs = datas = None
    if cfg["codec"] == "lzma":
        datas, pays, crcs = make_markov_lzma(c, n_unique, size, seed, max(args.gen_seconds, 90.0), world)
    else:
        if strong and world > 1 and not os.environ.get("MZHIP_BENCH_NO_TABLE_CACHE"):
            offs, pays, crcs = shared_unique_deflate(c, n_unique, size, seed, args.gen_seconds * cfg.get("gen_scale", 1.0), world, local, "c%d" % args.config)
        else:
            offs, pays, crcs = make_unique_deflate(c, n_unique, size, seed, args.gen_seconds * cfg.get("gen_scale", 1.0), world)
    # MZHIP_BENCH_SHARE_GPU=1 (tests/test_gpu_bench_ranks.py, a box with ONE GPU): every rank uses device 0 and the
    # collectives go through gloo on host copies -- the N > 1 logic (sharding, gather, reductions) on real kernels
    # where RCCL cannot run (it refuses two ranks on one device).  Never set by the driver: the product path is RCCL.
    share = world > 1 and share_gpu
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # this process next to its GPU (two sockets: the far one costs a third of the PCIe rate and half of the reader threads'
    # memcpy rate, profiles/r3/threads_numa.log): the CPUs of the device's NUMA node, as many as the container's CPU quota
    # when this is the only rank (threads that wander over 256 CPUs strand quota slices), the whole node otherwise.
    # Applies to everything measured from here on, the reference's CPU baseline included.
    near = mz.lib().mzhip_bind_thread_near_device(local, usable_cores() if world == 1 else 0)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_reduce(t, op):
        if share:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def all_gather_into(dst, src):
        if share:
            h = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_gather_into_tensor(h, src.cpu())
            dst.copy_(h)
        else:
            dist.all_gather_into_tensor(dst, src)

    U = len(pays)
    if strong and world > 1:
        # one table for all ranks: the slice generator stops early on a slow host (--gen-seconds), so agree on the
        # number of unique streams every rank really has (same seeds: the first U are the same everywhere)
        u = torch.tensor([U], dtype=torch.int64, device=dev)
        all_reduce(u, dist.ReduceOp.MIN)
        U = int(u.item())
        pays, crcs = pays[:U], crcs[:U]
        if offs is not None:
            offs = offs[:U]
    rnd = np.random.RandomState(99 if strong else 99 + rank)
    if U >= n_table:
        pick_all = np.arange(n_table)  # every entry is its own stream
    else:
        # fewer streams than entries (config 3's table is 524 288, or the generator ran out of time, or --unique asked for fewer):
        # every stream is used once, the rest of the entries are repeats drawn at random, in a random order
        pick_all = np.concatenate((np.arange(U), rnd.randint(0, U, size=n_table - U)))
        rnd.shuffle(pick_all)
    tiled = n_table - len(np.unique(pick_all))
    plen = np.array([len(p) for p in pays], dtype=np.int64)
    lo, hi = 0, n_table
    bounds = None
    if strong and world > 1:
        table = np.zeros((n_table, 8), dtype=np.int64)
        table[:, archive.COL_CSIZE] = plen[pick_all]
        table[:, archive.COL_USIZE] = size
        bounds = archive.shard_bounds(table, world)  # contiguous slices balanced by compressed + uncompressed bytes
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    pick = pick_all[lo:hi]
    n = len(pick)
    max_shard = int(np.diff(bounds).max()) if bounds is not None else n

    L = mz.lib()
    want_crc_np = crcs[pick]
    want_crc = torch.from_numpy(want_crc_np.view(np.int32).copy()).to(dev)  # the central directory's CRCs
    if cfg["codec"] in ("inflate", "lzma"):
        d_in, h_in, in_off, in_len = device_blob(torch, dev, pays, pick)
        for e in (0, n // 3, n // 2, n - 1):  # the device input really is the compressed stream
            assert d_in[in_off[e]:in_off[e] + in_len[e]].cpu().numpy().tobytes() == pays[pick[e]]
        d_in_off = torch.from_numpy(in_off).to(dev)
        d_in_len = torch.from_numpy(in_len.astype(np.int32)).to(dev)
        d_out = torch.empty(n * size, dtype=torch.uint8, device=dev)
        d_out_off = torch.arange(n, dtype=torch.int64, device=dev) * size
        d_out_cap = torch.full((n,), size, dtype=torch.int32, device=dev)
        algo_bytes = int(in_len.sum()) + n * size
        ratio = float(in_len.sum()) / (n * size)
        if cfg["codec"] == "lzma":
            L.mzhip_lzma_batch.restype = C.c_int32
            L.mzhip_lzma_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
            r_len, r_used, r_crc, r_st = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
            d_max = torch.full((n,), size, dtype=torch.int64, device=dev)  # TOTAL_OUT_MAX, as mz_zip.c:1833-1846 sets it
    else:
        srcs = [c[o:o + size] for o, _ in offs]
        d_src, h_src, in_off, in_len = device_blob(torch, dev, srcs, pick)
        cap = size + size // 8 + 64
        d_in_off = torch.from_numpy(in_off).to(dev)
        d_in_len = torch.from_numpy(in_len.astype(np.int32)).to(dev)
        d_out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
        d_out_off = torch.arange(n, dtype=torch.int64, device=dev) * cap
        d_out_cap = torch.full((n,), cap, dtype=torch.int32, device=dev)
        L.mzhip_deflate_batch.restype = C.c_int32
        L.mzhip_deflate_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4
        r_len, r_crc, r_st = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
        algo_bytes = None  # known once the compressed sizes are

    gathered = torch.empty(world * max_shard * 2, dtype=torch.int32, device=dev) if world > 1 else None
    mine = torch.zeros(max_shard * 2, dtype=torch.int32, device=dev) if world > 1 else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    stats = {}
    stream = torch.cuda.current_stream().cuda_stream

    def launch():
        if cfg["codec"] == "inflate":
            return mz.inflate_batch(d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap)
        if cfg["codec"] == "lzma":
            rc = L.mzhip_lzma_batch(d_in.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), d_out.data_ptr(),
                                    d_out_off.data_ptr(), d_out_cap.data_ptr(), d_max.data_ptr(), n, r_len.data_ptr(),
                                    r_used.data_ptr(), r_crc.data_ptr(), r_st.data_ptr(), C.c_void_p(stream))
            assert rc == 0, rc
            return r_len, r_used, r_crc, r_st
        rc = L.mzhip_deflate_batch(d_src.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), d_out.data_ptr(),
                                   d_out_off.data_ptr(), d_out_cap.data_ptr(), None, n, r_len.data_ptr(), r_crc.data_ptr(),
                                   r_st.data_ptr(), C.c_void_p(stream))
        assert rc == 0, rc
        return r_len, None, r_crc, r_st

    def step(i_timed=None):
        """one pass of the hot path over the rank's shard (+ the CRC gather when N > 1) and the per-entry comparison with
        the central directory's CRCs, lengths and consumed bytes -- on every step, inside the clock."""
        if i_timed is not None:
            ev[i_timed][0].record()
        out_len, in_used, crc, status = launch()
        if i_timed is not None:
            ev[i_timed][1].record()
        if world > 1:
            if i_timed is not None:
                gev[i_timed][0].record()
            mine[:2 * n] = torch.stack((crc, status)).reshape(-1)
            all_gather_into(gathered, mine)
            if i_timed is not None:
                gev[i_timed][1].record()
        if True:  # every step, timed or not (the check stays inside the clock; a handful of elementwise launches, ~30 us)
            if cfg["codec"] == "deflate":
                ok = (crc == want_crc) & (status == 0) & (out_len > 0)
            else:
                ok = (crc == want_crc) & (status == 0) & (out_len == size) & (in_used == d_in_len)
            stats["match"] = ok.sum()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    match = int(stats["match"].item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gev])) if world > 1 else 0.0  # the {crc, status} all-gather behind the launch

    # byte-exact spot check of the output buffer (outside the timed region)
    bytes_ok = True
    if cfg["codec"] == "deflate":
        ol = r_len.cpu().numpy().astype(np.int64)
        algo_bytes = int(ol.sum()) + n * size
        ratio = float(ol.sum()) / (n * size)
        for e in range(0, n, max(1, n // 64)):
            z = d_out[e * cap:e * cap + int(ol[e])].cpu().numpy().tobytes()
            bytes_ok = bytes_ok and zlib.decompress(z, -15) == srcs[pick[e]]  # round trip through zlib's inflate
    else:
        for e in range(0, n, max(1, n // 64)):
            got = d_out[e * size:(e + 1) * size].cpu().numpy().tobytes()
            want = datas[pick[e]] if datas is not None else c[offs[pick[e]][0]:offs[pick[e]][0] + size]
            bytes_ok = bytes_ok and got == want

    total_entries = n
    algo_all = float(algo_bytes)
    rank_ms = [kernel_ms]
    rank_entries = [n]
    if world > 1:
        pr = torch.zeros(2 * world, dtype=torch.float64, device=dev)
        pr[2 * rank], pr[2 * rank + 1] = kernel_ms, float(n)
        all_reduce(pr, dist.ReduceOp.SUM)
        rank_ms = [round(float(x), 3) for x in pr[0::2].tolist()]
        rank_entries = [int(x) for x in pr[1::2].tolist()]
        t = torch.tensor([elapsed, kernel_ms, gather_ms], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        gather_ms = float(t[2].item())
        s = torch.tensor([float(match), float(n), float(algo_bytes), float(bytes_ok)], dtype=torch.float64, device=dev)
        all_reduce(s, dist.ReduceOp.SUM)
        elapsed, kernel_ms = float(t[0].item()), float(t[1].item())
        match, total_entries, algo_all = int(s[0].item()), int(s[1].item()), float(s[2].item())
        bytes_ok = int(s[3].item()) == world

    if rank == 0:
        traffic, traffic_raw, traffic_src = measured_traffic(args.config, n, size)
        value = total_entries * size * args.steps / elapsed / 2**30
        achieved = algo_all / world / (kernel_ms / 1e3) / 1e9  # per GPU: the slowest rank's launch over an average shard
        line = {
            "metric": cfg["metric"], "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
            "data": "synthetic: %d x %d B %s of %s (%s, ratio %.3f); %d unique streams%s%s, each entry with its own "
                    "bytes in HBM" % (n_table, size, "order-2 word-Markov expansions" if datas is not None else "slices", cdesc,
                                      {"inflate": "zlib level 6 raw", "lzma": "LZMA preset 6 + end marker",
                                       "deflate": "compressed here at level 1"}[cfg["codec"]], ratio, n_table - tiled,
                                      (", %d entries are tiled repeats" % tiled) if tiled else ", none tiled",
                                      "" if strong else " per GPU"),
            "crc32_match_rate": match / total_entries, "bytes_spot_check": bool(bytes_ok),
            "config": {"workload": cfg["workload"] % (n_table, size), "config": args.config, "unique_streams": int(n_table - tiled),
                       "entries_total": total_entries, "entries_rank0": n, "entry_bytes": size,
                       "sharding": ("one entry table, contiguous slices balanced by c+u bytes (archive.shard_bounds), "
                                    if strong else "independent table per rank, ") +
                                   "RCCL all_gather of per-entry {crc,status} only" if world > 1 else "single GPU",
                       "host_placement": ("process bound to %d CPUs of the GPU's NUMA node: %s" % (near, sorted(os.sched_getaffinity(0))[:1] + sorted(os.sched_getaffinity(0))[-1:])
                                          if near > 0 else "not bound (one NUMA node, or sysfs does not name the device's)")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_raw": traffic_raw, "traffic_source": traffic_src,
                         "kernel": cfg["kernel"], "kernel_ms": round(kernel_ms, 3), "kernel_ms_per_rank": rank_ms,
                         "algorithmic_bytes_per_launch": int(algo_all / world), "per_gpu": True},
        }
        if cfg["codec"] == "inflate":
            geo = (C.c_uint32(), C.c_uint32(), C.c_uint32())
            L.mzhip_inflate_launch_geometry(n, *(C.byref(g) for g in geo))
            launched = geo[0].value * geo[1].value  # what this shard's launch starts (every resident slot once it has that many entries)
            L.mzhip_inflate_launch_geometry(0x7FFFFFFF, *(C.byref(g) for g in geo))
            resident = geo[0].value * geo[1].value
            # one wave decodes one entry and the waves of a launch are persistent: a shard of E equal entries takes
            # ceil(E / resident waves) rounds, so a shard that is not a multiple of the resident waves pays for a last,
            # partly empty round (at N = 8, 12 500 entries over 4096 waves = 3.05 -> 4 rounds)
            rounds = [e / max(resident, 1) for e in rank_entries]
            line["config"]["launch"] = {"workgroups": launched // max(geo[1].value, 1), "waves_per_wg": geo[1].value, "lds_bytes_per_wg": geo[2].value,
                                        "resident_waves": resident, "entries_per_rank": rank_entries,
                                        "rounds_per_rank": [round(r, 3) for r in rounds],
                                        "predicted_quantisation_efficiency": round(min(r / max(1.0, float(np.ceil(r))) for r in rounds), 3)}
        else:
            line["config"]["launch"] = {"entries_per_rank": rank_entries}
        if world > 1:
            line["config"]["launch"]["gather_ms"] = round(gather_ms, 3)  # slowest rank, mean of the timed steps (waits for the slowest rank's launch)
        sample_zip = None
        if world == 1 and not args.no_cpu_baseline:
            cores = usable_cores()  # threads the CPU baseline really gets (affinity and cgroup quota, not the CPU count)
            if cfg["codec"] == "inflate":
                sample_zip = os.path.join(tempfile.mkdtemp(prefix="mzhip_bench_"), "sample.zip")
                cb = cpu_baseline_inflate(c, offs, size, pays, crcs, cores, keep_path=sample_zip)
            elif cfg["codec"] == "lzma":
                cb = cpu_baseline_lzma(datas, cores)
            else:
                cb = cpu_baseline_deflate(c, offs, size, cores)
            if cb is not None:
                line["cpu_baseline"] = cb
        if world == 1 and args.config == 2 and not args.no_legs:
            # the archive of leg (iii): 16 384 of the bench's own streams (1 GiB decoded), so that the prime pipeline has
            # something to pipeline; the reference writer's 2 048-entry sample of the cpu_baseline leg when that is all there is
            legs_zip = sample_zip
            k_leg = min(len(pays), 16384, (1 << 32) // max(size, 1) - 1)
            if k_leg > 2048:
                legs_zip = os.path.join(tempfile.mkdtemp(prefix="mzhip_legs_"), "legs.zip")
                write_stream_zip(legs_zip, [pays[i] for i in pick[:k_leg]], want_crc_np[:k_leg], size)
            line["legs"] = legs_config2(torch, mz, dev, h_in, in_off, in_len, size, want_crc_np, value, legs_zip)
            if len(pays) >= n and n >= 65536 and n * 1.0 * size < 60e9 and sum(len(pays[i]) for i in pick) + 100 * n < (1 << 32):
                # the headline archive itself: every entry of the table, one file
                full_zip = os.path.join(tempfile.mkdtemp(prefix="mzhip_full_"), "cfg2.zip")
                write_stream_zip(full_zip, [pays[i] for i in pick], want_crc_np, size)
                fl, fcb = full_archive_legs(mz, full_zip, n, size, want_crc_np, usable_cores(), not args.no_cpu_baseline)
                line["legs"].update(fl)
                if fcb and "cpu_baseline" in line:
                    # the baseline of record is the whole archive; the sampled figure stays beside it
                    line["cpu_baseline"] = dict(fcb, sampled=line["cpu_baseline"])
                os.remove(full_zip)
                os.rmdir(os.path.dirname(full_zip))
            if not (args.entries or args.entry_size or args.unique):
                ll, lcb = large_entry_leg(c, 512, not args.no_cpu_baseline)
                line["legs"].update(ll)
                if lcb and "cpu_baseline" in line:
                    line["cpu_baseline"]["one_large_entry"] = lcb
            if legs_zip != sample_zip and legs_zip:
                os.remove(legs_zip)
                os.rmdir(os.path.dirname(legs_zip))
        if sample_zip and os.path.exists(sample_zip):
            os.remove(sample_zip)
            os.rmdir(os.path.dirname(sample_zip))
        if world == 1 and args.config == 2 and not args.no_other_configs and not (args.entries or args.entry_size or args.unique):
            # BASELINE.json configs[2..4] behind the headline, each as its own process (its own data, its own HIP context; this
            # process's tensors stay allocated: 9 GB of 288), each with its own roofline and cpu_baseline
            del d_in, d_out
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()  # (every rank has loaded the shared table long ago: it can go)
        if _TABLE_CACHE and os.path.exists(_TABLE_CACHE):
            os.remove(_TABLE_CACHE)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
