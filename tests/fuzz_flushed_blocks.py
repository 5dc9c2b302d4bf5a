#!/usr/bin/env python3
"""Differential run of mz_stream_zlib READ on streams of MANY SMALL BLOCKS (not pytest): 5 - 300 pieces of 0 - 3 000 bytes -- text, noise,
runs, copies of earlier output up to 40 000 bytes back -- with a flush (sync, full, block, partial) behind four of five, every level and
strategy (fixed, Huffman only, RLE, filtered), memory levels 1 - 9, raw / zlib / gzip framing and windows of 9 and 12 bits; whole, cut
anywhere, two single-bit flips; read() calls of 65 535, 7, 1 000 bytes and 1 MiB; with and without TOTAL_IN_MAX; one buffer and in windows
(two builds of the library) -- against the all-reference build: every read() return value, byte, TOTAL_IN / TOTAL_OUT, close(), error().
    python tests/fuzz_flushed_blocks.py [seed] [streams per library] [library ...]    (default: the two host-emulation builds)"""
import os, sys, zlib, random
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import oracle
from tests import synth
rnd=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1); N=int(sys.argv[2]) if len(sys.argv)>2 else 100
text=synth.bench_corpus()[0]
ref=oracle.ref()
KEYS=("rets","out","total_in","total_out","close","error","open")
tot=bad=0
LIBS=sys.argv[3:] or [os.path.join(ROOT,'tests','emul',b,'libmockdrop.so') for b in ('_build','_build_small')]
for libn in LIBS:
    hip=oracle.MzDriver(libn)
    for it in range(N):
        wb=rnd.choice((-15,-15,15,31,-9,-12))
        co=zlib.compressobj(rnd.randrange(0,10),zlib.DEFLATED,wb,rnd.randrange(1,10),rnd.choice((0,0,zlib.Z_FIXED,zlib.Z_HUFFMAN_ONLY,zlib.Z_RLE,zlib.Z_FILTERED)))
        z=b''; d=b''
        for _ in range(rnd.randrange(5,300)):
            k=rnd.randrange(4); n=rnd.choice((0,1,2,5,30,300,3000))
            if k==0: o=rnd.randrange(len(text)-n-1); p=text[o:o+n]
            elif k==1: p=bytes(rnd.getrandbits(8) for _ in range(n))
            elif k==2: p=bytes([rnd.randrange(256)])*n
            else: p=d[-rnd.randrange(1,40000):][:n] if d else b''
            d+=p; z+=co.compress(p)
            if rnd.random()<0.8: z+=co.flush(rnd.choice((zlib.Z_SYNC_FLUSH,zlib.Z_FULL_FLUSH,zlib.Z_BLOCK,zlib.Z_PARTIAL_FLUSH)))
        z+=co.flush()
        vs=[('whole',z),('cut',z[:rnd.randrange(0,len(z)+1)])]
        for _ in range(2):  # (no flips under a window below 15 bits: there inflate() refuses a distance by the history it holds PLUS what the current call
            # has produced -- the caller's buffer size decides, INTEGRATION.md "One deviation" -- and a corrupted distance is where that shows)
            zz=bytearray(z); zz[rnd.randrange(len(zz))]^=1<<rnd.randrange(8)
            if wb in (-15,15,31): vs.append(('flip',bytes(zz)))
        for name,data in vs:
            chunk=rnd.choice((65535,7,1000,1<<20)); mi=rnd.choice((0,len(data)))
            a=hip.stream_decode(8,data,len(d)+70000,chunk=chunk,window_bits=wb,max_in=mi); b=ref.stream_decode(8,data,len(d)+70000,chunk=chunk,window_bits=wb,max_in=mi)
            tot+=1
            diff={k:(a[k],b[k]) for k in KEYS if k!='out' and a[k]!=b[k]}
            if diff or a['out']!=b['out']:
                bad+=1
                if bad<12:
                    if os.environ.get('MZ_FUZZ_DUMP'): open(os.path.join(os.environ['MZ_FUZZ_DUMP'],'flush_bad_%d.bin'%bad),'wb').write(data)
                    print(libn,it,name,'wb',wb,'chunk',chunk,'len',len(data),{k:((len(v[0]),len(v[1]),v[0][-2:],v[1][-2:]) if k=='rets' else v) for k,v in diff.items()},'out eq' if a['out']==b['out'] else 'OUT DIFF %d %d'%(len(a['out']),len(b['out'])))
print('flush fuzz cases',tot,'bad',bad)
