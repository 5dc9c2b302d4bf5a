import torch, time
dev=torch.device("cuda",0)
n=1<<30
d=torch.empty(n,dtype=torch.uint8,device=dev)
h=torch.empty(n,dtype=torch.uint8).pin_memory()
def t(f,reps=3):
    best=1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter(); f(); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    return best
print("D2H 1 GiB one copy: %.1f GB/s"%(n/1e9/t(lambda: h.copy_(d,non_blocking=True))))
print("H2D 1 GiB one copy: %.1f GB/s"%(n/1e9/t(lambda: d.copy_(h,non_blocking=True))))
ss=[torch.cuda.Stream() for _ in range(3)]
def chunks(dst,src,c):
    for i,o in enumerate(range(0,n,c)):
        with torch.cuda.stream(ss[i%3]): dst[o:o+c].copy_(src[o:o+c],non_blocking=True)
for c in (8<<20,48<<20,256<<20):
    print("D2H chunks %d MiB on 3 streams: %.1f GB/s"%(c>>20,n/1e9/t(lambda: chunks(h,d,c))))
h2=torch.empty(n//3,dtype=torch.uint8).pin_memory(); d2=torch.empty(n//3,dtype=torch.uint8,device=dev)
def both():
    with torch.cuda.stream(ss[0]): h.copy_(d,non_blocking=True)
    with torch.cuda.stream(ss[1]): d2.copy_(h2,non_blocking=True)
print("D2H 1 GiB + H2D 1/3 GiB together: %.1f ms"%(t(both)*1e3))
import os
# first-touch cost of a fresh pinned buffer
t0=time.perf_counter(); hh=torch.empty(n,dtype=torch.uint8).pin_memory(); print("pin 1 GiB: %.1f ms"%((time.perf_counter()-t0)*1e3))
