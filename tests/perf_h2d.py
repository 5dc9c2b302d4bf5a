"""Quick GPU probe (not a pytest): where the time of bench.py's `h2d_kernel_d2hcrc` leg goes -- the H2D copy alone, the launches
alone, both in chunks on three streams, with the event times of every chunk.  Usage: python tests/perf_h2d.py [entries=20480]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib, importlib.util
import torch
from tests import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); sys.modules["bench"] = bench; spec.loader.exec_module(bench)
mz = importlib.import_module("minizip-ng_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
size = 65536
dev = torch.device("cuda:0")
c, _ = synth.bench_corpus()
offs, pays, crcs = bench.make_unique_deflate(c, n, size, 77, 60, 6)
in_len = np.array([len(p) for p in pays], dtype=np.int64)
in_off = np.concatenate(([0], np.cumsum((in_len + 15) // 16 * 16)[:-1])).astype(np.int64)
end = int(in_off[-1] + (in_len[-1] + 15) // 16 * 16)
h_in = np.zeros(end, dtype=np.uint8)
for i, p in enumerate(pays): h_in[in_off[i]:in_off[i] + len(p)] = np.frombuffer(p, dtype=np.uint8)
hp = torch.from_numpy(h_in).pin_memory()
d_in = torch.empty(end, dtype=torch.uint8, device=dev)
d_off = torch.from_numpy(in_off).to(dev); d_len = torch.from_numpy(in_len.astype(np.int32)).to(dev)
d_out = torch.empty(n * size, dtype=torch.uint8, device=dev)
d_oo = torch.arange(n, dtype=torch.int64, device=dev) * size
d_oc = torch.full((n,), size, dtype=torch.int32, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
print("%d entries, %.1f MB compressed, %.1f MiB decoded" % (n, end / 1e6, n * size / 2**20))

def run(nchunk, h2d=True, kern=True, nstreams=3, trace=False):
    cuts = [n * i // nchunk for i in range(nchunk + 1)]
    best, tl = None, None
    for rep in range(4):
        ev = []
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t0 = time.perf_counter()
        for ci in range(nchunk):
            lo, hi = cuts[ci], cuts[ci + 1]
            b0, b1 = int(in_off[lo]), (int(in_off[hi]) if hi < n else end)
            with torch.cuda.stream(streams[ci % nstreams]):
                ea, eb, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                ea.record()
                if h2d: d_in[b0:b1].copy_(hp[b0:b1], non_blocking=True)
                eb.record()
                if kern: r = mz.inflate_batch(d_in, d_off[lo:hi], d_len[lo:hi], d_out, d_oo[lo:hi], d_oc[lo:hi])
                ec.record()
                ev.append((ea, eb, ec))
        t_q = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best = dt
            tl = [(e0.elapsed_time(a), e0.elapsed_time(b), e0.elapsed_time(c_)) for a, b, c_ in ev] + [t_q * 1e3]
    print("chunks %2d streams %d h2d %d kernel %d: %.2f ms  %.1f GiB/s decoded" % (nchunk, nstreams, h2d, kern, best * 1e3, n * size / 2**30 / best))
    if trace:
        for ci, (a, b, c_) in enumerate(tl[:-1]): print("   chunk %d: h2d %.2f .. %.2f ms, kernel .. %.2f ms" % (ci, a, b, c_))
        print("   host done queueing at %.2f ms" % tl[-1])

run(1, True, False); run(1, False, True); run(1, True, True)
run(5, True, False); run(5, False, True)
run(5, True, True, trace=True)
run(5, True, True, nstreams=2)
run(4, True, True); run(8, True, True); run(10, True, True, trace=True); run(20, True, True)
