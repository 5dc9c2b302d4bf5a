"""CPU tests of the C-ABI boundary: libmzhip.so loads without a GPU and exports every symbol
include/mzhip.h and include/mz_strm_hip.h declare; the vtbl the shims publish has the reference's
12-slot layout (mz_strm.h:53-67).  No compute entry point is called here."""
import ctypes as C
import importlib
import os
import re

import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mz = importlib.import_module("minizip-ng_amd")


def declared_symbols():
    names = []
    for h in ("mzhip.h", "mz_strm_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"(?m)^\s*#[^\n]*", "", src)
        names += re.findall(r"MZHIP_API\s+[^;(]*?\b(\w+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(mz.LIB_PATH):
        mz.build()
    return mz.lib()


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 36, names
    for n in names:
        assert hasattr(lib, n), "libmzhip.so does not export %s" % n
    for n in mz.BATCH_SYMBOLS:
        assert n in names


def test_ctypes_signatures_come_from_the_header(lib):
    """Every MZHIP_API function gets its restype / argtypes from include/mzhip.h (minizip-ng_amd.header_prototypes): a change
    of a signature cannot leave a hand-written list behind (round 4: test_one_window_on_many_waves crashed the GPU suite that
    way).  The two args structs of the host entry points are restated in Python; their field lists are checked against the
    header's text, and the retired suffix-versioned entry points are gone."""
    protos = mz.header_prototypes()
    assert set(protos) == set(declared_symbols()) - {n for n in declared_symbols() if not n.startswith("mzhip_")}
    for name, (ret, argt) in protos.items():
        fn = getattr(lib, name)
        assert fn.restype == ret and list(fn.argtypes) == argt, name
    assert len(protos["mzhip_inflate_parallel_host"][1]) == 16 and len(protos["mzhip_inflate_batch"][1]) == 12
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mzhip.h")).read(), flags=re.S)
    for cls, cname in ((mz.InflateHostArgs, "mzhip_inflate_host_args"), (mz.DeflateHostArgs, "mzhip_deflate_host_args"),
                       (mz.InflateState, "mzhip_inflate_state")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ptr = "*" in decl
            names = [w.strip(" *") for w in decl.replace("*", " * ").split(",")]
            names[0] = names[0].split()[-1]
            fields += [(n.split()[-1].strip("*"), ptr) for n in names]
        got = [(n.rstrip("_"), t is C.c_void_p) for n, t in cls._fields_]
        assert got == fields, (cname, got, fields)
    for gone in ("mzhip_inflate_host2", "mzhip_inflate_resume_host", "mzhip_inflate_resume_host_seg", "mzhip_inflate_resume_host_seg2",
                 "mzhip_deflate_host2", "mzhip_deflate_host_level",
                 # the names that took positional arguments until round 4 are exported by NO build any more -- a binary
                 # compiled against the old header must fail to link, not pass its input buffer as the args struct
                 "mzhip_inflate_host", "mzhip_deflate_host"):
        assert not hasattr(lib, gone), gone
    assert hasattr(lib, "mzhip_inflate_host_a") and hasattr(lib, "mzhip_deflate_host_a")
    assert b"0.6" in lib.mzhip_version()


def test_dropin_symbol_set_matches_reference_headers(lib):
    """exactly the 13 functions mz_strm_zlib.h:20-35 / mz_strm_lzma.h:20-35 declare, plus the CRC"""
    per_stream = ["open", "is_open", "read", "write", "tell", "seek", "close", "error", "get_prop_int64",
                  "set_prop_int64", "create", "delete", "get_interface"]
    for codec in ("zlib", "lzma"):
        for f in per_stream:
            assert hasattr(lib, "mz_stream_%s_%s" % (codec, f))
    assert hasattr(lib, "mz_crypt_crc32_update")


def test_vtbl_layout_and_instance_header(lib):
    """get_interface() returns 12 function pointers in the reference's order; create() returns an
    instance whose first word is that vtbl and whose second word (base) is NULL (mz_strm.h:69-72)."""
    for codec in ("zlib", "lzma"):
        gi = getattr(lib, "mz_stream_%s_get_interface" % codec)
        gi.restype = C.c_void_p
        vt = C.cast(gi(), C.POINTER(C.c_void_p * 12)).contents
        order = ["open", "is_open", "read", "write", "tell", "seek", "close", "error", "create", "delete",
                 "get_prop_int64", "set_prop_int64"]
        for slot, name in enumerate(order):
            fn = getattr(lib, "mz_stream_%s_%s" % (codec, name))
            assert vt[slot] == C.cast(fn, C.c_void_p).value, (codec, name)
        create = getattr(lib, "mz_stream_%s_create" % codec)
        create.restype = C.c_void_p
        inst = create()
        words = C.cast(inst, C.POINTER(C.c_void_p * 2)).contents
        assert words[0] == gi() and not words[1]
        # props answer like the reference before open (mz_strm_zlib.c:312-355, mz_strm_lzma.c:380-428)
        getp = getattr(lib, "mz_stream_%s_get_prop_int64" % codec)
        setp = getattr(lib, "mz_stream_%s_set_prop_int64" % codec)
        getp.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        setp.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        v = C.c_int64(-7)
        assert getp(inst, 1, C.byref(v)) == 0 and v.value == 0          # TOTAL_IN
        assert getp(inst, 5, C.byref(v)) == 0 and v.value == (0 if codec == "zlib" else 4)  # HEADER_SIZE
        assert getp(inst, 8, C.byref(v)) == -107                         # DISK_NUMBER -> MZ_EXIST_ERROR
        assert setp(inst, 2, 1234) == 0 and getp(inst, 2, C.byref(v)) == 0 and v.value == 1234
        if codec == "zlib":
            assert getp(inst, 11, C.byref(v)) == 0 and v.value == -15   # raw deflate, 32 KiB window
            assert setp(inst, 4, 10) == -107                            # TOTAL_OUT_MAX unknown to zlib stream
        else:
            assert getp(inst, 4, C.byref(v)) == 0 and v.value == -1
            assert setp(inst, 4, -2) == -102                            # MZ_PARAM_ERROR
        isopen = getattr(lib, "mz_stream_%s_is_open" % codec)
        isopen.argtypes = [C.c_void_p]
        assert isopen(inst) == -111                                      # MZ_OPEN_ERROR before open
        for name, want in (("tell", -114), ("seek", -113)):
            f = getattr(lib, "mz_stream_%s_%s" % (codec, name))
            f.restype = C.c_int64 if name == "tell" else C.c_int32
            f.argtypes = [C.c_void_p] if name == "tell" else [C.c_void_p, C.c_int64, C.c_int32]
            assert (f(inst) if name == "tell" else f(inst, 0, 0)) == want
        delete = getattr(lib, "mz_stream_%s_delete" % codec)
        p = C.c_void_p(inst)
        delete(C.byref(p))
        assert not p.value


def test_no_gpu_means_loud_failure(lib):
    """Without a HIP device every compute entry point must refuse -- there is no CPU fallback."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mz.MzHipError):
        mz.require_gpu()
    with pytest.raises(mz.MzHipError):
        mz.inflate_host(b"\x03\x00", 16)


def test_crc32_small_calls_stay_on_the_host(lib):
    """mz_crypt_crc32_update below MZHIP_CRC_HOST_BELOW bytes is folded on the host with the product's own tables
    (the reference calls it once per byte from mz_strm_pkcrypt.c:79,86 with a ~-wrapped state): no device, no abort,
    same chaining contract (mz_crypt.c:81,90).  Runs with or without a GPU."""
    import os
    import time
    import zlib

    f = lib.mz_crypt_crc32_update
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_char_p, C.c_int32]
    data = os.urandom(100000)
    t0 = time.time()
    v = 0
    for i in range(len(data)):
        v = f(v, data[i:i + 1], 1)
    dt = time.time() - t0
    assert v == zlib.crc32(data)
    assert dt < 2.0, dt          # ~0.5 us per call is ctypes; a launch per byte would take minutes
    # the pkcrypt pattern: the state travels inverted between calls (mz_strm_pkcrypt.c:75-89)
    k = 0x12345678
    for b in data[:2000]:
        k = (~f(~k & 0xFFFFFFFF, bytes([b]), 1)) & 0xFFFFFFFF
    want = 0x12345678
    for b in data[:2000]:
        want = (~zlib.crc32(bytes([b]), ~want & 0xFFFFFFFF)) & 0xFFFFFFFF
    assert k == want
    for n in (0, 1, 2, 3, 5, 17, 255, 4095):
        assert f(123, data[7:7 + n], n) == zlib.crc32(data[7:7 + n], 123), n
    if oracle.have_ref():
        r = oracle.ref()
        for n in (1, 3, 4095):
            assert f(77, data[:n], n) == r.crc32(data[:n], 77)


def test_crc32_without_device_is_exact_and_latched(lib):
    """No usable device and a large buffer: the symbol has no error channel, so the value is still exact (host fold) and
    the failure is recorded (mzhip_last_error; the next codec-stream call of the thread returns MZ_STREAM_ERROR) instead
    of aborting the process."""
    import os
    import zlib

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = lib.mz_crypt_crc32_update
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_char_p, C.c_int32]
    data = os.urandom(70000)
    assert f(5, data, len(data)) == zlib.crc32(data, 5)
    lib.mzhip_last_error.restype = C.c_char_p
    assert lib.mzhip_last_error()          # the reason is on record; a codec stream cannot even be opened without a device
