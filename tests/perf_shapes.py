import sys, zlib, time, numpy as np, torch
sys.path.insert(0,'.')
from tests import gpu_util, synth
mz=gpu_util.mz
def run(name, datas, reps=3):
    pays=[synth.deflate_raw(d) for d in datas]
    n=len(pays)
    batch=gpu_util.make_batch(pays,[len(d) for d in datas])
    want=np.array([zlib.crc32(d) for d in datas],dtype=np.uint32)
    for r in range(reps):
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); out_len,in_used,crc,status=mz.inflate_batch(batch["d_in"],batch["in_off"],batch["in_len"],batch["d_out"],batch["out_off"],batch["out_cap"]); e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1); tot=sum(len(d) for d in datas)
    ok=bool((status.cpu().numpy()==0).all() and (mz.u32(crc)==want).all())
    print("%-28s %8.3f ms %8.2f GiB/s ok=%s ratio %.4f"%(name,ms,tot/2**30/(ms/1e3),ok,sum(map(len,pays))/tot))
rnd=np.random.RandomState(1)
c=synth.corpus()
run("zeros 4096 x 1 MiB",[bytes(1<<20)]*4096)
run("runs 4096 x 256 KiB",[b"".join(bytes([int(rnd.randint(256))])*int(rnd.randint(1,3000)) for _ in range(180))[:262144].ljust(262144,b'x') for _ in range(64)]*64)
run("period-3 4096 x 256 KiB",[b"abc"*87382]*4096)
run("random 8192 x 64 KiB",[rnd.bytes(65536) for _ in range(128)]*64)
run("text 8192 x 64 KiB",[c[o:o+65536] for o in rnd.randint(0,len(c)-65536,size=256)]*32)
run("text 4096 x 1 MiB",[(c*3)[o:o+(1<<20)] for o in rnd.randint(0,len(c),size=16)]*256)
