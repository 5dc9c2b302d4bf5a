"""ctypes bindings for the oracle (TEST INFRASTRUCTURE -- never imported by the product path).

Two checkers live here:

* ``liboracle``  -- this repo's CPU restatement (oracle/crc32.c, inflate.c, lzma_dec.c).
* ``libmzref``   -- the unmodified reference compiled from /root/reference together with
  integration/mz_driver.c (zlib 1.2.11 + liblzma 5.2.5 behind mz_strm_zlib.c / mz_strm_lzma.c /
  mz_crypt.c).  Built by oracle/Makefile in the build container; travels to the GPU box as
  a prebuilt file under oracle/_ref/.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libmzref.so")

ORC_OK, ORC_DATA_ERROR, ORC_BUF_ERROR, ORC_OUT_FULL = 0, -3, -5, -200

_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the restatement (always) and oracle/_ref (only where /root/reference exists)."""
    if force or not os.path.exists(_ORACLE_SO) or os.path.exists("/root/reference/mz_zip.c"):
        subprocess.run(["make", "-s", "-C", _HERE], check=True, capture_output=True)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


def _as_u8(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build()
        L = C.CDLL(_ORACLE_SO)
        L.orc_crc32_update.restype = C.c_uint32
        L.orc_crc32_update.argtypes = [C.c_uint32, _u8p, C.c_size_t]
        L.orc_crc32_combine.restype = C.c_uint32
        L.orc_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_inflate_raw.restype = C.c_int32
        L.orc_inflate_raw.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_lzma_zip_decode.restype = C.c_int32
        L.orc_lzma_zip_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int64, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def crc32(data, value=0):
    a = _as_u8(data)
    return int(lib().orc_crc32_update(value, _ptr(a), a.size))


def crc32_combine(c1, c2, len2):
    return int(lib().orc_crc32_combine(c1, c2, len2))


def inflate_raw(data, out_cap):
    """-> (status, in_used, out_bytes)"""
    a = _as_u8(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    iu, ol = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_inflate_raw(_ptr(a), a.size, _ptr(out), out_cap, C.byref(iu), C.byref(ol))
    return int(st), int(iu.value), out[: ol.value].tobytes()


def xz_decode(data, out_cap, max_out=-1):
    """-> (status, in_used, out_bytes) for a method-95 payload (one .xz stream)"""
    L = lib()
    L.orc_xz_decode.restype = C.c_int32
    L.orc_xz_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int64, C.POINTER(C.c_size_t),
                                C.POINTER(C.c_size_t)]
    a = _as_u8(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    iu, ol = C.c_size_t(), C.c_size_t()
    st = L.orc_xz_decode(_ptr(a), a.size, _ptr(out), out_cap, max_out, C.byref(iu), C.byref(ol))
    return st, iu.value, out[:ol.value].tobytes()


def crc64(data):
    L = lib()
    L.orc_crc64.restype = C.c_uint64
    L.orc_crc64.argtypes = [_u8p, C.c_size_t]
    a = _as_u8(data)
    return L.orc_crc64(_ptr(a), a.size)


def sha256(data):
    L = lib()
    L.orc_sha256.restype = None
    L.orc_sha256.argtypes = [_u8p, C.c_size_t, _u8p]
    a = _as_u8(data)
    out = np.zeros(32, dtype=np.uint8)
    L.orc_sha256(_ptr(a), a.size, _ptr(out))
    return out.tobytes()


def lzma_zip_decode(data, out_cap, max_out=-1):
    a = _as_u8(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    iu, ol = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_lzma_zip_decode(_ptr(a), a.size, _ptr(out), out_cap, max_out, C.byref(iu), C.byref(ol))
    return int(st), int(iu.value), out[: ol.value].tobytes()


# --------------------------------------------------------------------------- reference


class MzDriver:
    """Binding of the drv_* entry points of integration/mz_driver.c.  The same class binds
    oracle/_ref/libmzref.so (reference codecs) and integration/_build/libmzhipdrop.so
    (reference zip layer + HIP codecs): identical signatures, comparable results."""

    def __init__(self, path):
        L = C.CDLL(path)
        self.path = path
        L.drv_crc32_update.restype = C.c_uint32
        L.drv_crc32_update.argtypes = [C.c_uint32, _u8p, C.c_int32]
        L.drv_stream_decode.restype = C.c_int32
        L.drv_stream_decode.argtypes = [C.c_int32, _u8p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _u8p, C.c_int32,
                                        C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int64)]
        L.drv_stream_encode.restype = C.c_int32
        L.drv_stream_encode.argtypes = [C.c_int32, C.c_int32, C.c_int32, _u8p, C.c_int32, C.c_int32, _u8p, C.c_int32,
                                        C.POINTER(C.c_int64)]
        L.drv_zip_write.restype = C.c_int32
        L.drv_zip_write.argtypes = [C.c_char_p, C.c_int32, C.c_int32, _u8p, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int32), C.c_int32]
        L.drv_zip_index.restype = C.c_int64
        L.drv_zip_index.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64]
        L.drv_zip_read_all.restype = C.c_double
        L.drv_zip_read_all.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), _u8p,
                                       C.POINTER(C.c_int64)]
        self.L = L

    def crc32(self, data, value=0):
        a = _as_u8(data)
        return int(self.L.drv_crc32_update(value, _ptr(a), a.size))

    def stream_decode(self, method, data, out_cap, chunk=65535, max_in=0, max_out=-1, window_bits=0):
        """-> dict(rets, out, total_in, total_out, close, error, base_pos, open)"""
        a = _as_u8(data)
        out = np.zeros(max(out_cap, 1), dtype=np.uint8)
        rets = (C.c_int32 * 4096)()
        info = (C.c_int64 * 6)()
        n = self.L.drv_stream_decode(method, _ptr(a), a.size, max_in, max_out, window_bits, _ptr(out), out_cap, chunk,
                                     rets, 4096, info)
        rl = [int(rets[i]) for i in range(min(max(n, 0), 4096))]
        produced = sum(r for r in rl if r > 0)
        return dict(rets=rl, out=out[:produced].tobytes(), total_in=int(info[0]), total_out=int(info[1]),
                    close=int(info[2]), error=int(info[3]), base_pos=int(info[4]), open=int(info[5]))

    def stream_delete_unclosed(self, method, data, out_cap, chunk=65535, nreads=4):
        """read() a few chunks, then delete() the codec stream without close(): -> the bytes read"""
        a = _as_u8(data)
        out = np.zeros(max(out_cap, 1), dtype=np.uint8)
        n = self.L.drv_stream_delete_unclosed(method, _ptr(a), a.size, _ptr(out), out_cap, chunk, nreads)
        return out[:max(n, 0)].tobytes()

    def stream_encode(self, method, data, level=6, chunk=65535, window_bits=0):
        a = _as_u8(data)
        cap = a.size + a.size // 2 + 4096
        out = np.zeros(cap, dtype=np.uint8)
        info = (C.c_int64 * 6)()
        n = self.L.drv_stream_encode(method, level, window_bits, _ptr(a), a.size, chunk, _ptr(out), cap, info)
        if n < 0:
            raise RuntimeError("drv_stream_encode failed: %d" % n)
        return out[:n].tobytes(), dict(total_in=int(info[0]), total_out=int(info[1]), close=int(info[2]),
                                       error=int(info[3]), open=int(info[5]))

    def zip_reader_walk(self, path, max_entries=1 << 20):
        """every entry through mz_zip_reader_entry_open / _read / _close (the layer that verifies Hash extra fields in a
        crypto build): -> (status[n] i32, ulen[n] i64)"""
        st = np.zeros(max_entries, dtype=np.int32)
        ul = np.zeros(max_entries, dtype=np.int64)
        self.L.drv_zip_reader_walk.restype = C.c_int64
        self.L.drv_zip_reader_walk.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int64]
        n = self.L.drv_zip_reader_walk(path.encode(), st.ctypes.data_as(C.POINTER(C.c_int32)), ul.ctypes.data_as(C.POINTER(C.c_int64)),
                                       max_entries)
        if n < 0:
            raise RuntimeError("drv_zip_reader_walk failed: %d" % n)
        return st[:n].copy(), ul[:n].copy()

    def zip_write_repeat(self, path, piece, total, method=8, level=6):
        """one entry of `total` bytes (the piece repeated) + a small one, written in 65 535-byte calls"""
        a = _as_u8(piece)
        self.L.drv_zip_write_repeat.argtypes = [C.c_char_p, C.c_int32, C.c_int32, _u8p, C.c_int32, C.c_int64]
        err = self.L.drv_zip_write_repeat(path.encode(), method, level, _ptr(a), a.size, total)
        if err != 0:
            raise RuntimeError("drv_zip_write_repeat failed: %d" % err)

    def zip_write(self, path, blob, offs, lens, method=8, level=6):
        b = _as_u8(blob)
        o = np.ascontiguousarray(offs, dtype=np.int64)
        l = np.ascontiguousarray(lens, dtype=np.int32)
        err = self.L.drv_zip_write(path.encode(), method, level, _ptr(b), o.ctypes.data_as(C.POINTER(C.c_int64)),
                                   l.ctypes.data_as(C.POINTER(C.c_int32)), len(l))
        if err != 0:
            raise RuntimeError("drv_zip_write failed: %d" % err)

    def zip_index(self, path, max_entries=1 << 21):
        """-> int64 array [n, 8]: method, flag, crc, csize, usize, local_off, cd_pos, payload_off"""
        t = np.zeros((max_entries, 8), dtype=np.int64)
        n = self.L.drv_zip_index(path.encode(), t.ctypes.data_as(C.POINTER(C.c_int64)), max_entries)
        if n < 0:
            raise RuntimeError("drv_zip_index failed: %d" % n)
        return t[:n].copy()

    def zip_read_all(self, path, cd_pos, nthreads=1, chunk=65535, own_crc=True, out=None, out_off=None, mapped=False):
        """-> (seconds, crc[n] u32, ulen[n] i64, status[n] i32).  mapped: every reader sits on mz_stream_mem over one shared
        read-only mapping of the archive instead of mz_zip_reader_open_file (whose split stream re-opens the file twice
        per entry: the difference between 3 and 30 GiB/s with 256 threads on one path)"""
        cd = np.ascontiguousarray(cd_pos, dtype=np.int64)
        n = len(cd)
        crc = np.zeros(n, dtype=np.uint32)
        ulen = np.zeros(n, dtype=np.int64)
        st = np.zeros(n, dtype=np.int32)
        po = _ptr(out) if out is not None else None
        oo = None
        if out_off is not None:
            out_off = np.ascontiguousarray(out_off, dtype=np.int64)
            oo = out_off.ctypes.data_as(C.POINTER(C.c_int64))
        sec = self.L.drv_zip_read_all(path.encode(), cd.ctypes.data_as(C.POINTER(C.c_int64)), n, nthreads, chunk,
                                      (1 if own_crc else 0) | (2 if mapped else 0), crc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      ulen.ctypes.data_as(C.POINTER(C.c_int64)),
                                      st.ctypes.data_as(C.POINTER(C.c_int32)), po, oo)
        return float(sec), crc, ulen, st


_ref = None


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    """The compiled reference.  Raises if oracle/_ref/libmzref.so was never built."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_SO):
            build()
        if not os.path.exists(_REF_SO):
            raise RuntimeError("oracle/_ref/libmzref.so missing: run `make -C oracle` where /root/reference exists")
        _ref = MzDriver(_REF_SO)
    return _ref
