#!/usr/bin/env python3
"""Differential run of the gzip / zlib wrappers of mz_stream_zlib READ (not pytest): hand-built gzip members -- every FLG combination,
FEXTRA / FNAME / FCOMMENT of 0 .. 40 000 bytes, FHCRC, trailing bytes -- and zlib streams around bodies of 0 .. 300 000 bytes; whole, cut
anywhere, cut inside the header, one bit flipped in the first 40 or the last 16 bytes; window bits 31 / 15 / 47 (auto); read() calls of 65 535,
7 bytes and 1 MiB; one buffer and in windows (two builds of the library) -- against the all-reference build: every read() return value,
byte, TOTAL_IN / TOTAL_OUT, close(), error(), is_open().  Round 5: 1 500 cases found one (a gzip trailer cut inside ISIZE behind a wrong
CRC field: -3 where it stands, not -5).
    python tests/fuzz_wrappers.py [seed] [members per library] [library ...]    (default: the two host-emulation builds)"""
import os, sys, zlib, random, struct
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import oracle
from tests import synth
rnd=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
N=int(sys.argv[2]) if len(sys.argv)>2 else 200
text=synth.bench_corpus()[0]
ref=oracle.ref()
KEYS=("rets","out","total_in","total_out","close","error","open")
tot=bad=0
LIBS=sys.argv[3:] or [os.path.join(ROOT,'tests','emul',b,'libmockdrop.so') for b in ('_build','_build_small')]
for libn in LIBS:
    hip=oracle.MzDriver(libn)
    for it in range(N):
        n=rnd.choice((0,1,10,1000,70000,300000)); o=rnd.randrange(len(text)-n-1); d=text[o:o+n]
        co=zlib.compressobj(rnd.randrange(0,10),zlib.DEFLATED,-15); body=co.compress(d)+co.flush()
        kind=rnd.choice(('gzip','gzip','zlib'))
        if kind=='gzip':
            flg=rnd.randrange(32) if rnd.random()<0.8 else rnd.randrange(256)
            h=bytearray(b'\x1f\x8b\x08'+bytes([flg])+struct.pack('<I',rnd.getrandbits(32))+bytes([rnd.choice((0,2,4)),rnd.randrange(256)]))
            if flg&4:
                xl=rnd.choice((0,1,5,300,40000)); h+=struct.pack('<H',xl)+bytes(rnd.getrandbits(8) for _ in range(xl))
            if flg&8: h+=bytes(rnd.randrange(1,256) for _ in range(rnd.choice((0,3,200,40000))))+b'\0'
            if flg&16: h+=bytes(rnd.randrange(1,256) for _ in range(rnd.choice((0,3,200))))+b'\0'
            if flg&2: h+=struct.pack('<H',zlib.crc32(bytes(h))&0xffff)
            z=bytes(h)+body+struct.pack('<II',zlib.crc32(d),len(d)&0xffffffff)+bytes(rnd.choice((0,0,7)))
            wbs=(31,47,31)
        else:
            z=zlib.compress(d,rnd.randrange(0,10))+bytes(rnd.choice((0,0,5))); wbs=(15,47,15)
        variants=[('whole',z),('cut',z[:rnd.randrange(0,len(z)+1)]),('cuthdr',z[:rnd.randrange(0,min(len(z),60)+1)])]
        for _ in range(2):
            zz=bytearray(z); at=rnd.randrange(min(len(zz),40)) if rnd.random()<0.5 else len(zz)-1-rnd.randrange(min(len(zz),16)); zz[at]^=1<<rnd.randrange(8); variants.append(('flip@%d'%at,bytes(zz)))
        for name,data in variants:
            wb=rnd.choice(wbs); chunk=rnd.choice((65535,7,1<<20))
            a=hip.stream_decode(8,data,len(d)+70000,chunk=chunk,window_bits=wb); b=ref.stream_decode(8,data,len(d)+70000,chunk=chunk,window_bits=wb)
            tot+=1
            diff={k:(a[k],b[k]) for k in KEYS if k!='out' and a[k]!=b[k]}
            if diff or a['out']!=b['out']:
                bad+=1
                if os.environ.get('MZ_FUZZ_DUMP'): open(os.path.join(os.environ['MZ_FUZZ_DUMP'],'wrap_bad_%d.bin'%bad),'wb').write(data)
                if bad<15: print(libn,it,kind,name,'wb',wb,'chunk',chunk,'len',len(data),{k:((v[0][-2:],v[1][-2:]) if k=='rets' else v) for k,v in diff.items()},'out eq' if a['out']==b['out'] else 'OUT DIFF')
print('wrapper fuzz cases',tot,'bad',bad)
