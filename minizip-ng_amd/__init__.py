"""minizip-ng_amd -- MI355X-native codec backend for minizip-ng (host-side Python binding).

The product is ``_build/libmzhip.so`` (hand-written HIP for gfx950 behind a C ABI, see
include/mzhip.h).  This module is only the thin ctypes binding used by the tests and by
bench.py; PyTorch appears solely as the owner of device memory and streams.

There is NO CPU fallback: importing works anywhere (so the C-ABI symbol checks can run
without a GPU), but every compute entry point raises if the library or a HIP device is missing.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MZHIP_LIB") or os.path.join(_HERE, "_build", "libmzhip.so")  # env: tuning builds only


_u64p, _u32p, _i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mzhip.h")


class InflateState(C.Structure):
    """include/mzhip.h mzhip_inflate_state"""
    _fields_ = [("hdr_bit", C.c_uint32), ("bit", C.c_uint32), ("out_pos", C.c_uint32), ("flags", C.c_uint32)]


class InflateHostArgs(C.Structure):
    """include/mzhip.h mzhip_inflate_host_args (tests/test_abi.py checks the layout against the header's text)"""
    _fields_ = [("size", C.c_uint32), ("in_len", C.c_uint32), ("buf_cap", C.c_uint32), ("seg_first", C.c_uint32),
                ("seg_stride", C.c_uint32), ("seg_cap", C.c_uint32), ("in_", C.c_void_p), ("buf", C.c_void_p),
                ("state_in", C.c_void_p), ("state_out", C.c_void_p), ("out_len", C.c_void_p), ("in_used", C.c_void_p),
                ("crc", C.c_void_p), ("adler", C.c_void_p), ("seg_crc", C.c_void_p), ("nseg", C.c_void_p)]


class DeflateHostArgs(C.Structure):
    """include/mzhip.h mzhip_deflate_host_args"""
    _fields_ = [("size", C.c_uint32), ("in_len", C.c_uint32), ("final", C.c_uint32), ("out_cap", C.c_uint32),
                ("level", C.c_int32), ("window_log2", C.c_int32), ("in_", C.c_void_p), ("out", C.c_void_p),
                ("out_len", C.c_void_p), ("crc", C.c_void_p), ("adler", C.c_void_p)]


_SCALARS = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "size_t": C.c_size_t,
            "int": C.c_int, "uint16_t": C.c_uint16, "uint8_t": C.c_uint8}


def header_prototypes(path=None):
    """{name: (restype, [argtypes])} of every MZHIP_API function include/mzhip.h declares -- the ONE source of the ctypes
    signatures (round 4 lost a GPU suite to a hand-written argtypes list that lagged the header by two arguments).  Every
    pointer becomes c_void_p (bytes, integers, byref() and ctypes arrays all pass), every scalar its fixed-width type."""
    import re

    text = open(path or HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"MZHIP_API\s+([^;{}()]+?)\b(mzhip_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())

        def ctype(decl, is_ret=False):
            decl = decl.strip()
            if "*" in decl:
                return C.c_char_p if is_ret and "char" in decl else C.c_void_p
            words = [w for w in decl.replace("const", " ").split() if w]
            base = words[0] if is_ret or len(words) == 1 else words[-2] if len(words) >= 2 else words[0]
            if base == "void":
                return None
            return _SCALARS[base]

        argt = [] if args in ("", "void") else [ctype(a) for a in args.split(",")]
        out[name] = (ctype(ret, True), argt)
    return out


# every function include/mzhip.h declares (tests/test_abi.py checks that the library exports them all)
BATCH_SYMBOLS = sorted(header_prototypes())


def bind(L, path=None):
    """restype / argtypes of every function the header declares, on a loaded library (libmzhip.so, the drop-in, the mock)"""
    for name, (ret, argt) in header_prototypes(path).items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype = ret
            fn.argtypes = argt
    return L


class MzHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP/C source for gfx950 into _build/libmzhip.so."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise MzHipError("libmzhip.so build failed")
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MzHipError("%s missing: run __graft_entry__.build() / make -C minizip-ng_amd/csrc" % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))  # every signature from include/mzhip.h
    return _lib


def _check(rc, what):
    if rc != 0:
        raise MzHipError("%s failed (%d): %s" % (what, rc, lib().mzhip_last_error().decode()))


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise MzHipError("no HIP device visible: the MI355X backend has no CPU fallback")
    n = lib().mzhip_device_count()
    if n <= 0:
        raise MzHipError("mzhip_device_count() = %d: %s" % (n, lib().mzhip_last_error().decode()))
    return n


def _stream_handle():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def inflate_batch(d_in, in_off, in_len, d_out, out_off, out_cap):
    """Launch K1+K2 over device-resident tensors on torch's current stream (asynchronous).

    d_in/d_out: uint8 CUDA tensors; in_off/out_off: int64; in_len/out_cap: int32 (values < 2^31).
    Returns CUDA tensors (out_len, in_used, crc [as int64-safe uint32 in int32 storage], status)."""
    import torch

    require_gpu()
    n = in_off.numel()
    dev = d_in.device
    out_len = torch.empty(n, dtype=torch.int32, device=dev)
    in_used = torch.empty(n, dtype=torch.int32, device=dev)
    crc = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    for t, dt in ((in_off, torch.int64), (out_off, torch.int64), (in_len, torch.int32), (out_cap, torch.int32)):
        assert t.dtype == dt and t.is_cuda and t.is_contiguous()
    assert d_in.dtype == torch.uint8 and d_out.dtype == torch.uint8
    with torch.cuda.device(dev):
        _check(lib().mzhip_inflate_batch(d_in.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), d_out.data_ptr(),
                                         out_off.data_ptr(), out_cap.data_ptr(), n, out_len.data_ptr(),
                                         in_used.data_ptr(), crc.data_ptr(), status.data_ptr(), _stream_handle()),
               "mzhip_inflate_batch")
    return out_len, in_used, crc, status


def crc32_batch(d_buf, off, length, init=None):
    import torch

    require_gpu()
    n = off.numel()
    crc = torch.empty(n, dtype=torch.int32, device=d_buf.device)
    with torch.cuda.device(d_buf.device):
        _check(lib().mzhip_crc32_batch(d_buf.data_ptr(), off.data_ptr(), length.data_ptr(), n,
                                       init.data_ptr() if init is not None else None, crc.data_ptr(),
                                       _stream_handle()), "mzhip_crc32_batch")
    return crc


def inflate_host(data, out_cap):
    """One entry through the host-buffer convenience entry point -> (status, in_used, out bytes, crc)."""
    require_gpu()
    out = C.create_string_buffer(max(out_cap, 1))
    src = C.create_string_buffer(bytes(data), max(len(data), 1))
    ol, iu, crc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    a = InflateHostArgs(size=C.sizeof(InflateHostArgs), in_len=len(data), buf_cap=out_cap, in_=C.addressof(src), buf=C.addressof(out),
                        out_len=C.addressof(ol), in_used=C.addressof(iu), crc=C.addressof(crc))
    st = lib().mzhip_inflate_host_a(C.byref(a))
    return int(st), int(iu.value), out.raw[: ol.value], int(crc.value)


def crc32_host(data, value=0):
    require_gpu()
    return int(lib().mzhip_crc32_host(value, bytes(data), len(data)))


def u32(t):
    """int32-stored uint32 tensor -> numpy uint32"""
    import numpy as np

    return t.detach().cpu().numpy().view(np.uint32)
