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

# symbols include/mzhip.h and include/mz_strm_hip.h declare (checked by tests/test_abi.py)
BATCH_SYMBOLS = [
    "mzhip_device_count", "mzhip_init", "mzhip_last_error", "mzhip_version", "mzhip_inflate_batch",
    "mzhip_crc32_batch", "mzhip_adler32_batch", "mzhip_lzma_batch", "mzhip_xz_batch", "mzhip_deflate_batch",
    "mzhip_sha_batch", "mzhip_inflate_host", "mzhip_inflate_host2", "mzhip_lzma_host", "mzhip_xz_host",
    "mzhip_deflate_host", "mzhip_deflate_host2", "mzhip_crc32_host", "mzhip_inflate_launch_geometry",
    "mzhip_zip_index_mem", "mzhip_prime_file", "mzhip_prime_mem", "mzhip_prime_mem_begin", "mzhip_prime_wait", "mzhip_device_local_cpus", "mzhip_bind_thread_near_device", "mzhip_prime_clear", "mzhip_prime_stats",
    "mzhip_prime_write", "mzhip_prime_write_clear", "mzhip_prime_write_stats", "mzhip_prime_file_multi",
    "mzhip_prime_mem_multi", "mzhip_shard_bounds", "mzhip_deflate_batch_level", "mzhip_deflate_host_level",
    "mzhip_lzma_encode_batch", "mzhip_lzma_encode_batch_preset", "mzhip_lzma_encode_host_preset", "mzhip_xz_encode_host_preset",
    "mzhip_inflate_resume_batch", "mzhip_inflate_resume_host", "mzhip_inflate_resume_host_seg", "mzhip_inflate_resume_host_seg2", "mzhip_inflate_parallel_host", "mzhip_inflate_large",
    "mzhip_set_stream_window", "mzhip_set_write_segment", "mzhip_set_stream_parallel", "mzhip_window_alloc", "mzhip_window_free", "mzhip_lzma_resume_host", "mzhip_lzma_model_bytes", "mzhip_lzma_encode_resume_host", "mzhip_xz_encode_block_host", "mzhip_xz_encode_finish_host",
]

_u64p, _u32p, _i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)


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
        L = C.CDLL(LIB_PATH)
        L.mzhip_last_error.restype = C.c_char_p
        L.mzhip_version.restype = C.c_char_p
        L.mzhip_inflate_batch.restype = C.c_int32
        L.mzhip_inflate_batch.argtypes = [C.c_void_p] * 11 + [C.c_void_p]
        L.mzhip_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mzhip_crc32_batch.restype = C.c_int32
        L.mzhip_crc32_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.mzhip_inflate_host.restype = C.c_int32
        L.mzhip_inflate_host.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p]
        L.mzhip_crc32_host.restype = C.c_uint32
        L.mzhip_crc32_host.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.mzhip_inflate_launch_geometry.restype = None
        L.mzhip_inflate_launch_geometry.argtypes = [C.c_uint32, _u32p, _u32p, _u32p]
        _lib = L
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
    ol, iu, crc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    st = lib().mzhip_inflate_host(bytes(data), len(data), out, out_cap, C.byref(ol), C.byref(iu), C.byref(crc))
    return int(st), int(iu.value), out.raw[: ol.value], int(crc.value)


def crc32_host(data, value=0):
    require_gpu()
    return int(lib().mzhip_crc32_host(value, bytes(data), len(data)))


def u32(t):
    """int32-stored uint32 tensor -> numpy uint32"""
    import numpy as np

    return t.detach().cpu().numpy().view(np.uint32)
