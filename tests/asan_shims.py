"""Not a pytest: the READ / WRITE shims' round-6 paths (window decoded ahead, segments coded ahead, compressed-size hints, window mode
through the zip layer, delete without close) on the host emulation under AddressSanitizer + UBSan.
    make -C tests/emul B=_build_par_asan SAN=address,undefined 'SHIM_DEFS=-DMZH_STREAM_WINDOW="(1536<<10)" -DMZH_STREAM_GULP="(256<<10)" \
        -DMZH_PAR_MIN_IN="(32<<10)" -DMZH_PAR_MIN_ROOM="(128<<10)" -DMZH_STREAM_EARLY="(64<<10)" -DMZH_WRITE_SEGMENT="(256<<10)"' _build_par_asan/libmockdrop.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 python tests/asan_shims.py"""
import os, sys, zlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from tests import synth
so=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul', '_build_par_asan', 'libmockdrop.so')
hip=oracle.MzDriver(so); ref=oracle.ref()
text,_=synth.bench_corpus()
big=text[:900000]*3
z=synth.deflate_raw(big,6)
for chunk in (65535, 300000, 777):
    a=ref.stream_decode(8,z,len(big)+10,chunk=chunk); b=hip.stream_decode(8,z,len(big)+10,chunk=chunk)
    print("read", chunk, a==b)
zz=bytearray(z); zz[len(zz)//2]^=0x10
a=ref.stream_decode(8,bytes(zz),2*len(big)+(1<<20)); b=hip.stream_decode(8,bytes(zz),2*len(big)+(1<<20)); a.pop("base_pos"); b.pop("base_pos"); print("flip", a==b)
print("cut", ref.stream_decode(8,z[:len(z)//2],len(big)+10)==hip.stream_decode(8,z[:len(z)//2],len(big)+10))
got=hip.stream_delete_unclosed(8, z, len(big), chunk=65535, nreads=6); print("delete unclosed", got==big[:len(got)])
for chunk in (65535, 1000):
    e=hip.stream_encode(8,big,level=1,chunk=chunk); print("write", chunk, zlib.decompress(e[0],-15)==big, e[1]["close"], e[1]["error"])
# archives: a few entries through the zip layer (csize hints, window mode, per-entry)
c=np.frombuffer(text*3,dtype=np.uint8)
with tempfile.TemporaryDirectory() as tmp:
    p=os.path.join(tmp,'a.zip')
    lens=np.array([900000, 70000, 300000, 5, 0],dtype=np.int32); offs=np.array([0,1000,2000,3000,4000],dtype=np.int64)
    ref.zip_write(p,c,offs,lens,method=8,level=6)
    cd=ref.zip_index(p)[:,6].copy()
    os.environ["MZHIP_AUTOPRIME"]="0"
    r1=ref.zip_read_all(p,cd,nthreads=1,own_crc=False); r2=hip.zip_read_all(p,cd,nthreads=1,own_crc=False)
    print("archive", (r1[1]==r2[1]).all(), (r1[3]==r2[3]).all(), r2[3])
print("asan run done")
