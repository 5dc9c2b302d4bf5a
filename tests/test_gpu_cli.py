"""The reference's own ctest matrix, driven through the reference's own command-line tools on the HIP codecs.

integration/_build/minizip_hip and minigzip_hip are the reference's minizip.c / minigzip.c compiled unmodified
and linked against the drop-in (reference zip layer + libmzhip.so, neither libz nor liblzma behind the codec
symbols); oracle/_ref/minizip_ref and minigzip_ref are the same two files on the reference codecs.  The matrix
below restates CMakeLists.txt:807-942: for each method (-0 raw, -9 deflate, -m lzma, -n xz) and each flavour
(generic, span "-k 1024", zipcd "-z"): zip -> list -> unzip -> append -> unzip -> erase -> unzip, plus the
gz/ungz pair -- and then every archive written on one side is extracted on the other.
"""
import filecmp
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_ZIP = os.path.join(ROOT, "integration", "_build", "minizip_hip")
HIP_GZ = os.path.join(ROOT, "integration", "_build", "minigzip_hip")
REF_ZIP = os.path.join(ROOT, "oracle", "_ref", "minizip_ref")
REF_GZ = os.path.join(ROOT, "oracle", "_ref", "minigzip_ref")

METHODS = [("raw", "-0"), ("deflate", "-9"), ("lzma", "-m"), ("xz", "-n")]
FLAVOURS = [("generic", []), ("span", ["-k", "1024"]), ("zipcd", ["-z"])]
MEMBERS = ["test.c", "test.h", "empty.txt", "random.bin", "uniform.bin", "fuzz"]


def _need(*paths):
    for p in paths:
        if not os.path.exists(p):
            pytest.skip(f"{os.path.relpath(p, ROOT)} not built (needs the reference tree at build time)")


def _run(exe, args, cwd):
    r = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, f"{os.path.basename(exe)} {' '.join(args)} -> {r.returncode}\n{out[-2000:]}"
    assert "Error" not in out, f"{os.path.basename(exe)} {' '.join(args)}\n{out[-2000:]}"
    return out


@pytest.fixture(scope="module")
def src(tmp_path_factory):
    """Stand-ins for the reference's test/ directory (same names, same shapes: text, empty, 128 KiB
    incompressible, 96 KiB of one byte, a directory tree, a one-byte file)."""
    d = tmp_path_factory.mktemp("cli_src")
    rng = np.random.default_rng(20250509)
    words = [b"stream", b"header", b"int32_t", b"return", b"mz_zip", b"entry", b"static", b"{", b"}", b";\n", b" "]
    text = b"".join(words[i] for i in rng.integers(0, len(words), 9000))
    (d / "test.c").write_bytes(text)
    (d / "test.h").write_bytes(text[:4097])
    (d / "empty.txt").write_bytes(b"")
    (d / "random.bin").write_bytes(rng.integers(0, 256, 131072, dtype=np.uint8).tobytes())
    (d / "uniform.bin").write_bytes(b"\x55" * 98304)
    (d / "single.txt").write_bytes(b"1")
    (d / "fuzz" / "corpus").mkdir(parents=True)
    (d / "fuzz" / "seed.dict").write_bytes(b'"PK\\x03\\x04"\n' * 40)
    for i in range(4):
        n = int(rng.integers(1, 70000))
        (d / "fuzz" / "corpus" / f"case{i}.bin").write_bytes(
            (rng.integers(0, 4 + 60 * i, n, dtype=np.uint8)).tobytes())
    return d


def _tree(base):
    out = []
    for r, _, fs in os.walk(base):
        out += [os.path.relpath(os.path.join(r, f), base) for f in fs]
    return sorted(out)


def _same(src, dest, names):
    for n in names:
        a, b = os.path.join(src, n), os.path.join(dest, n)
        if os.path.isdir(a):  # minizip.c adds a directory argument's contents relative to that directory
            sub = _tree(a)
            assert sub, n
            _same(a, dest, sub)
        else:
            assert os.path.isfile(b), f"{n} missing from {dest}"
            assert filecmp.cmp(a, b, shallow=False), f"{n} differs"


def _matrix(exe, other, src, work, mname, marg, fname, fargs):
    z = str(work / f"{mname}-{fname}.zip")
    dest = str(work / f"{mname}-{fname}")
    _run(exe, [marg, "-o"] + fargs + [z] + MEMBERS, src)
    listing = _run(exe, ["-l"] + fargs + [z], src)
    for n in ("test.c", "random.bin", "uniform.bin"):
        assert n in listing
    if mname != "raw":
        assert mname in listing
    _run(exe, ["-x", "-o"] + fargs + ["-d", dest, z], src)
    _same(str(src), dest, MEMBERS)
    # a freshly written archive, read by the other side's codecs
    _run(other, ["-x", "-o"] + fargs + ["-d", dest + "-cross", z], src)
    _same(str(src), dest + "-cross", MEMBERS)
    _run(exe, [marg, "-a"] + fargs + [z, "single.txt"], src)
    # minizip.c's append under -z rewrites the compressed central directory with the new entry only (the
    # reference does the same on its own codecs), so that flavour keeps just single.txt from here on
    kept = ["single.txt"] if fname == "zipcd" else MEMBERS + ["single.txt"]
    for who, tag in ((exe, "-appended"), (other, "-appended-cross")):
        _run(who, ["-x", "-o"] + fargs + ["-d", dest + tag, z], src)
        _same(str(src), dest + tag, kept)
    if fname == "generic":  # CMakeLists.txt:875 passes no EXTRA_ARGS to the erase step: only the plain flavour is erasable
        _run(exe, ["-o", "-e", z, "test.c", "test.h"], src)
        for who, tag in ((exe, "-erased"), (other, "-erased-cross")):
            _run(who, ["-x", "-o", "-d", dest + tag, z], src)
            _same(str(src), dest + tag, ["empty.txt", "random.bin", "uniform.bin", "fuzz", "single.txt"])
            assert not os.path.exists(os.path.join(dest + tag, "test.c"))
            assert not os.path.exists(os.path.join(dest + tag, "test.h"))


@pytest.mark.parametrize("fname,fargs", FLAVOURS, ids=[f[0] for f in FLAVOURS])
@pytest.mark.parametrize("mname,marg", METHODS, ids=[m[0] for m in METHODS])
def test_cli_matrix_hip(src, tmp_path, mname, marg, fname, fargs):
    """zip/list/unzip/append/erase with the reference CLI on the HIP codecs; the reference codecs read the result."""
    _need(HIP_ZIP, REF_ZIP)
    if fname == "zipcd" and mname == "raw":
        pytest.skip("CMakeLists.txt:813-816: the raw method is left out of the -z flavour")
    _matrix(HIP_ZIP, REF_ZIP, src, tmp_path, mname, marg, fname, fargs)


@pytest.mark.parametrize("mname,marg", METHODS[1:], ids=[m[0] for m in METHODS[1:]])
def test_cli_reference_archive_read_by_hip(src, tmp_path, mname, marg):
    """An archive written by the reference codecs (level 9 / preset 9 streams) extracted by the HIP codecs."""
    _need(HIP_ZIP, REF_ZIP)
    z = str(tmp_path / f"ref-{mname}.zip")
    _run(REF_ZIP, [marg, "-o", z] + MEMBERS, src)
    dest = str(tmp_path / "out")
    _run(HIP_ZIP, ["-x", "-o", "-d", dest, z], src)
    _same(str(src), dest, MEMBERS)


def test_cli_gz_ungz(src, tmp_path):
    """CMakeLists.txt:932-942: minigzip random.bin, then minigzip -x; both directions against the reference."""
    _need(HIP_GZ, REF_GZ)
    for name in ("random.bin", "test.c", "uniform.bin"):
        for k, (packer, unpacker) in enumerate(((HIP_GZ, HIP_GZ), (HIP_GZ, REF_GZ), (REF_GZ, HIP_GZ))):
            w = tmp_path / f"{name}-{k}"
            w.mkdir()
            data = (src / name).read_bytes()
            (w / name).write_bytes(data)
            _run(packer, [name], str(w))
            gz = (w / (name + ".gz")).read_bytes()
            assert gz[:3] == b"\x1f\x8b\x08"
            os.remove(w / name)
            (w / "out").mkdir()
            _run(unpacker, ["-x", "-d", str(w / "out"), name + ".gz"], str(w))
            assert (w / "out" / name).read_bytes() == data


HIP_COMPAT = os.path.join(ROOT, "integration", "_build", "compat_hip")
REF_COMPAT = os.path.join(ROOT, "oracle", "_ref", "compat_ref")


def test_compat_layer(tmp_path):
    """test/test_compat.cc restated (integration/compat_check.c): the minizip 1.x API -- zipOpen64 ... zipClose,
    unzOpen ... unzClose -- on the HIP codecs, and each side's archive read by the other."""
    _need(HIP_COMPAT, REF_COMPAT)
    for k, (w, r) in enumerate(((HIP_COMPAT, HIP_COMPAT), (HIP_COMPAT, REF_COMPAT), (REF_COMPAT, HIP_COMPAT))):
        z = str(tmp_path / f"compat{k}.zip")
        for exe, mode in ((w, "write"), (r, "read")):
            p = subprocess.run([exe, mode, z], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
            assert p.returncode == 0, f"{os.path.basename(exe)} {mode}: {p.returncode} failed checks\n{p.stdout.decode()}"
