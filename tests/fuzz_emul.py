"""CPU-side probe (not a pytest): a long differential run of K1 in the 64-lane host emulation against the oracle -- valid streams of
every zlib level / strategy / window / memLevel, with mid-stream flushes, and bit flips, byte smashes and cuts of them, at
four input / output misalignments and tight capacities.  Usage: python tests/fuzz_emul.py [seed=1] [N=2500] [-D flags of the build ...]
(e.g. -DMZ_REC_CAP1=16u -DMZ_REC_CAP2=8u -DMZ_CHASE_SMAX=512u: the chase window with every cap in reach)"""
import sys, random, zlib, ctypes as C
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests import test_kernel_emul as t, synth
import oracle
L=t._build_variant("fuzz%d" % (int(sys.argv[1]) if len(sys.argv) > 1 else 1), sys.argv[3:])
c=synth.corpus()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
rnd=random.Random(seed)
def gen():
    k, n = rnd.randrange(7), rnd.randrange(1, 90000)
    if k == 0: d = c[rnd.randrange(len(c) - n):][:n]
    elif k == 1: d = bytes(rnd.randrange(256) for _ in range(min(n, 4000)))
    elif k == 2: d = bytes([rnd.randrange(4)]) * n
    elif k == 3: d = bytes(min(255, int(rnd.expovariate(1 / (8 + 200 * rnd.random())))) for _ in range(min(n, 8000)))
    elif k == 4: d = (c[rnd.randrange(1000):][:rnd.randrange(1, 300)]) * rnd.randrange(1, 60)
    elif k == 5: d = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 300))) + c[:n]
    else:
        d = bytearray()
        while len(d) < n:
            o = rnd.randrange(len(c)-500); d += c[o:o+rnd.randrange(3,500)]
            if rnd.random()<0.3: d += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1,40)))
        d = bytes(d)
    co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -rnd.randrange(9, 16), rnd.randrange(1, 10),
                          rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    z = co.compress(d[:len(d) // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if rnd.random() < 0.3 else b"") + co.compress(d[len(d) // 2:]) + co.flush()
    return d, z
bad=0; ok=0
for it in range(N):
    d,z=gen()
    k=it%4
    zz=bytearray(z)
    if k==1: zz[rnd.randrange(len(zz))]^=1<<rnd.randrange(8)
    elif k==2: del zz[rnd.randrange(1,len(zz)):]
    elif k==3: zz[rnd.randrange(len(zz))]=rnd.randrange(256)
    zz=bytes(zz)
    cap=rnd.choice((len(d)+8, len(d), max(1,len(d)-1), 200000))
    so,uo,oo=oracle.inflate_raw(zz,cap)
    st,used,out,crc=t._run(L.emul_inflate,zz,cap,mis=it%4,omis=(it//4)%4)
    if st!=so or (so==0 and ((used,out)!=(uo,oo) or crc!=oracle.crc32(oo))):
        bad+=1; print("MISMATCH",it,k,st,so,used,uo,len(out),len(oo))
    ok+= so==0
print("seed",seed,"n",N,"decoded",ok,"bad",bad)
