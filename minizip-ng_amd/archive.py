"""Archive-level batch path: index a ZIP, shard its entries, decode a whole shard in a few launches.

Host side of the "thousands of members in one launch" path (SURVEY 8b "Batching"): the central directory is
indexed once by the C indexer (mzhip_zip_index_mem), the archive bytes are placed in HBM as they are (the entry
payloads are used in place -- no repacking), and every entry of the shard is decoded by mzhip_inflate_batch /
mzhip_lzma_batch / mzhip_xz_batch / mzhip_crc32_batch according to its method.  The per-entry CRC is compared with
the central-directory CRC exactly where the reference compares it (mz_zip.c:2116-2128); with verify_hash the first
Hash extrafield (0x1a51, doc/mz_extrafield.md) of each entry is checked against a device-computed SHA digest the way
mz_zip_reader_entry_open / _close do it on the CPU in a crypto build (mz_zip_rw.c:409-451,465-466).

Sharding: entries are independent, so ranks take contiguous slices of the entry table balanced by
compressed+uncompressed bytes; the only collective is the gather of the per-entry {crc, out_len, status} words.
"""
import ctypes as C
import importlib
import mmap
import os

import numpy as np

_mz = importlib.import_module("minizip-ng_amd")

COL_METHOD, COL_FLAG, COL_CRC, COL_CSIZE, COL_USIZE, COL_LOCAL, COL_CDPOS, COL_PAYLOAD = range(8)
MZ_FORMAT_ERROR = -103
MZ_CRC_ERROR = -105       # mz.h:34
MZ_SUPPORT_ERROR = -109   # mz.h:38
MZ_ZIP_EXTENSION_HASH = 0x1A51   # mz.h:113
MZ_HASH_SHA1, MZ_HASH_SHA256 = 20, 23   # mz.h:127,131


def hash_fields(buf, table):
    """First Hash extrafield of every entry's central-directory record (mz_zip_reader_entry_get_first_hash,
    mz_zip_rw.c:560-600): -> (algorithm u16[n] (0 = none), digest_size u16[n], digest u8[n, 64])."""
    a = np.frombuffer(buf, dtype=np.uint8)
    n = len(table)
    alg = np.zeros(n, dtype=np.uint16)
    dsz = np.zeros(n, dtype=np.uint16)
    dig = np.zeros((n, 64), dtype=np.uint8)
    for i in range(n):
        p = int(table[i, COL_CDPOS])
        fn, ex = int(a[p + 28]) | int(a[p + 29]) << 8, int(a[p + 30]) | int(a[p + 31]) << 8
        q, end = p + 46 + fn, p + 46 + fn + ex
        while q + 4 <= end:
            fid, fsz = int(a[q]) | int(a[q + 1]) << 8, int(a[q + 2]) | int(a[q + 3]) << 8
            if fid == MZ_ZIP_EXTENSION_HASH and fsz >= 4 and q + 4 + fsz <= end:
                alg[i] = int(a[q + 4]) | int(a[q + 5]) << 8
                k = min(int(a[q + 6]) | int(a[q + 7]) << 8, fsz - 4, 64)
                dsz[i] = k
                dig[i, :k] = a[q + 8:q + 8 + k]
                break
            q += 4 + fsz
    return alg, dsz, dig


def index_bytes(buf):
    """Entry table [n, 8] int64 (columns COL_*) of the archive held in `buf` (bytes / mmap / uint8 array)."""
    a = np.frombuffer(buf, dtype=np.uint8)
    L = _mz.lib()
    L.mzhip_zip_index_mem.restype = C.c_int64
    L.mzhip_zip_index_mem.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64]
    n = L.mzhip_zip_index_mem(a.ctypes.data, a.size, None, 0)
    if n < 0:
        raise _mz.MzHipError("mzhip_zip_index_mem: %d" % n)
    t = np.zeros((max(n, 1), 8), dtype=np.int64)
    n2 = L.mzhip_zip_index_mem(a.ctypes.data, a.size, t.ctypes.data, n)
    assert n2 == n
    return t[:n]


def index_file(path):
    with open(path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            raise _mz.MzHipError("empty file")
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
            return index_bytes(m)


def shard_bounds(table, world):
    """Contiguous slices [b[r], b[r+1]) of the entry table, balanced by compressed + uncompressed bytes."""
    w = (table[:, COL_CSIZE] + table[:, COL_USIZE] + 64).astype(np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    targets = cum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(cum, targets, side="left")
    return np.concatenate(([0], cuts, [len(table)])).astype(np.int64)


def gather_results(results, world, group=None):
    """results: int64 tensor [n_local, 3] = (crc, out_len, status) of this rank's slice, any device.
    All-gathers the (ragged) slices; returns the full [n, 3] table on every rank.  The ONLY collective."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return results
    n_local = torch.tensor([results.shape[0]], dtype=torch.int64, device=results.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    m = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((m, 3), dtype=torch.int64, device=results.device)
    pad[: results.shape[0]] = results
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(c.item())] for p, c in zip(parts, counts)], dim=0)


LARGE_ENTRY = 4 << 20   # compressed bytes from which a DEFLATE entry is decoded by a wave per block (as mzhip_prime_* does)


class DeviceArchive:
    """A ZIP archive resident in HBM, decoded shard-wise by the batch kernels."""

    def __init__(self, path, device="cuda:0"):
        import torch

        _mz.require_gpu()
        self.path = path
        self.device = torch.device(device)
        self.table = index_file(path)
        self.h_file = np.fromfile(path, dtype=np.uint8)
        self.d_file = torch.from_numpy(self.h_file).to(self.device)

    def decode(self, lo=0, hi=None, keep_output=True, verify_hash=False):
        """Decode entries [lo, hi).  Returns dict(crc u32[n], out_len i64[n], status i32[n], ok bool[n],
        out (uint8 CUDA tensor) , out_off i64[n]).  status: 0, MZ_* / zlib-numbered errors, MZ_CRC_ERROR when
        the CRC differs from the central directory, MZ_SUPPORT_ERROR for methods other than 0 / 8 / 14 / 95.
        verify_hash: entries carrying a Hash extrafield are also checked against a device-computed SHA-1 / SHA-256
        (mismatch -> MZ_CRC_ERROR, other algorithms -> MZ_SUPPORT_ERROR, as mz_zip_reader_entry_open / _close)."""
        import torch

        t = self.table[lo:hi]
        n = len(t)
        dev = self.device
        usize = t[:, COL_USIZE]
        out_off = np.zeros(n, dtype=np.int64)
        if n:
            np.cumsum(((usize + 15) // 16 * 16)[:-1], out=out_off[1:])
        total = int(out_off[-1] + (usize[-1] + 15) // 16 * 16) if n else 0
        d_out = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
        crc = np.zeros(n, dtype=np.uint32)
        out_len = np.zeros(n, dtype=np.int64)
        status = np.full(n, MZ_SUPPORT_ERROR, dtype=np.int32)
        if (t[:, COL_PAYLOAD] < 0).any():
            raise _mz.MzHipError("entry without a usable local header")
        # sizes come from the archive: nothing negative, nothing that reaches past the file image (the indexer already
        # refuses both; a table handed in from elsewhere is checked again here before it becomes device offsets)
        flen = int(self.h_file.size)
        if n and ((t[:, COL_CSIZE] < 0).any() or (usize < 0).any() or (t[:, COL_PAYLOAD] + t[:, COL_CSIZE] > flen).any()):
            raise _mz.MzHipError("entry sizes outside the archive")
        L = _mz.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def dev_i64(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)

        def dev_i32(a):
            return torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)

        with torch.cuda.device(dev):
            # DEFLATE entries of LARGE_ENTRY compressed bytes and more: one wave per entry is 0.1 - 0.2 GB/s, so each of them
            # is decoded by a wave per DEFLATE block (mzhip_inflate_large, csrc/inflate_parallel.inc), one call per entry
            large = np.nonzero((t[:, COL_METHOD] == 8) & ((t[:, COL_FLAG] & 1) == 0) & (t[:, COL_CSIZE] >= LARGE_ENTRY))[0]
            if len(large) and ((t[large, COL_CSIZE] >= 2**32) | (usize[large] >= 2**32)).any():
                raise _mz.MzHipError("entries >= 4 GiB are outside the batch path")
            L.mzhip_inflate_large.restype = C.c_int32
            L.mzhip_inflate_large.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
            for e in large:
                ol, iu, ck, st1 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32()
                rc = L.mzhip_inflate_large(self.d_file.data_ptr() + int(t[e, COL_PAYLOAD]), int(t[e, COL_CSIZE]),
                                           d_out.data_ptr() + int(out_off[e]), int(usize[e]), C.byref(ol), C.byref(iu), C.byref(ck),
                                           C.byref(st1), stream)
                if rc != 0:
                    raise _mz.MzHipError("mzhip_inflate_large failed: %d %s" % (rc, L.mzhip_last_error().decode()))
                crc[e], out_len[e] = ck.value, ol.value
                status[e] = MZ_CRC_ERROR if (st1.value == 0 and iu.value == t[e, COL_CSIZE] and ck.value != np.uint32(t[e, COL_CRC])) else st1.value
            for method in (8, 14, 95, 0):
                sel = np.nonzero((t[:, COL_METHOD] == method) & ((t[:, COL_FLAG] & 1) == 0) &
                                 ~((t[:, COL_METHOD] == 8) & (t[:, COL_CSIZE] >= LARGE_ENTRY)))[0]
                if len(sel) == 0:
                    continue
                k = len(sel)
                if (t[sel, COL_CSIZE] >= 2**31).any() or (usize[sel] >= 2**31).any():
                    raise _mz.MzHipError("entries >= 2 GiB are outside the batch path")
                d_in_off, d_in_len = dev_i64(t[sel, COL_PAYLOAD]), dev_i32(t[sel, COL_CSIZE])
                d_out_off, d_cap = dev_i64(out_off[sel]), dev_i32(usize[sel])
                r_len, r_used, r_crc, r_st = (torch.zeros(k, dtype=torch.int32, device=dev) for _ in range(4))
                if method == 8:
                    rc = L.mzhip_inflate_batch(self.d_file.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(),
                                               d_out.data_ptr(), d_out_off.data_ptr(), d_cap.data_ptr(), k,
                                               r_len.data_ptr(), r_used.data_ptr(), r_crc.data_ptr(), r_st.data_ptr(),
                                               stream)
                elif method in (14, 95):
                    fn = L.mzhip_lzma_batch if method == 14 else L.mzhip_xz_batch
                    fn.restype = C.c_int32
                    fn.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
                    # TOTAL_OUT_MAX = uncompressed size when the EOS flag is set (mz_zip.c:1833-1846), else none
                    d_max = dev_i64(np.where(t[sel, COL_FLAG] & 2, usize[sel], -1))
                    rc = fn(self.d_file.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(),
                            d_out.data_ptr(), d_out_off.data_ptr(), d_cap.data_ptr(), d_max.data_ptr(),
                            k, r_len.data_ptr(), r_used.data_ptr(), r_crc.data_ptr(), r_st.data_ptr(), stream)
                else:   # STORE: the payload IS the data (mz_stream_raw, mz_zip.c:1769); CRC in place, then copy
                    rc = L.mzhip_crc32_batch(self.d_file.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), k, None,
                                             r_crc.data_ptr(), stream)
                    r_len = d_in_len.clone()
                    r_used = d_in_len.clone()
                    # a stored entry's two sizes must agree (the slot was sized from the uncompressed one): anything
                    # else is a format error and nothing is copied for it
                    same = t[sel, COL_CSIZE] == usize[sel]
                    r_st = dev_i32(np.where(same, 0, MZ_FORMAT_ERROR))
                    if keep_output:
                        for j, e in enumerate(sel):   # host-driven D2D copies: STORE is the CPU-plumbing config
                            if not same[j]:
                                continue
                            c0, cl = int(t[e, COL_PAYLOAD]), int(t[e, COL_CSIZE])
                            d_out[out_off[e]:out_off[e] + cl] = self.d_file[c0:c0 + cl]
                if rc != 0:
                    raise _mz.MzHipError("batch launch failed: %d %s" % (rc, L.mzhip_last_error().decode()))
                torch.cuda.synchronize()
                crc[sel] = _mz.u32(r_crc)
                out_len[sel] = r_len.cpu().numpy()
                st = r_st.cpu().numpy().astype(np.int32)
                used = r_used.cpu().numpy().astype(np.int64)
                # mz_zip_entry_read_close: CRC is verified iff the whole entry was consumed (mz_zip.c:2116-2128)
                bad_crc = (st == 0) & (used == t[sel, COL_CSIZE]) & (crc[sel] != t[sel, COL_CRC].astype(np.uint32))
                st[bad_crc] = MZ_CRC_ERROR
                status[sel] = st
        if verify_hash and n:
            alg, dsz, dig = hash_fields(self.h_file, t)
            status[(alg != 0) & (alg != MZ_HASH_SHA1) & (alg != MZ_HASH_SHA256) & (status == 0)] = MZ_SUPPORT_ERROR
            L.mzhip_sha_batch.restype = C.c_int32
            L.mzhip_sha_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_uint32] + [C.c_void_p] * 2
            with torch.cuda.device(dev):
                for a_id in (MZ_HASH_SHA1, MZ_HASH_SHA256):
                    sel = np.nonzero((alg == a_id) & (status == 0))[0]
                    if len(sel) == 0:
                        continue
                    if not keep_output and (t[sel, COL_METHOD] == 0).any():
                        raise _mz.MzHipError("verify_hash of STORE entries needs keep_output")
                    d_dig = torch.zeros(len(sel) * 32, dtype=torch.uint8, device=dev)
                    d_o, d_l = dev_i64(out_off[sel]), dev_i32(out_len[sel])   # keep both alive across the launch
                    rc = L.mzhip_sha_batch(d_out.data_ptr(), d_o.data_ptr(), d_l.data_ptr(), len(sel), a_id,
                                           d_dig.data_ptr(), stream)
                    if rc != 0:
                        raise _mz.MzHipError("mzhip_sha_batch failed: %d" % rc)
                    torch.cuda.synchronize()
                    got = d_dig.cpu().numpy().reshape(len(sel), 32)
                    for j, e in enumerate(sel):   # memcmp over the extrafield's digest size (mz_zip_rw.c:446-447)
                        k = min(int(dsz[e]), 32)
                        if not (got[j, :k] == dig[e, :k]).all():
                            status[e] = MZ_CRC_ERROR
        ok = (status == 0) & (out_len == usize)
        return dict(crc=crc, out_len=out_len, status=status, ok=ok, out=d_out if keep_output else None,
                    out_off=out_off)
