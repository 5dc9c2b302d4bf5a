"""Diagnostic (not a pytest): reproduce bench.py's batch at full size and report which entries fail and why."""
import importlib
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

mz = importlib.import_module("minizip-ng_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
size = 65536
dev = torch.device("cuda:0")
c, offs, pays, crcs = bench.make_unique(2048, size, 1234, 20)
U = len(pays)
rnd = np.random.RandomState(99)
pick = rnd.randint(0, U, size=n)
plen = np.array([len(p) for p in pays], dtype=np.int64)
in_len = plen[pick]
in_off = np.zeros(n, dtype=np.int64)
np.cumsum(((in_len + 15) // 16 * 16)[:-1], out=in_off[1:])
total_in = int(in_off[-1] + (in_len[-1] + 15) // 16 * 16)
uoff = np.zeros(U, dtype=np.int64)
np.cumsum(((plen + 15) // 16 * 16)[:-1], out=uoff[1:])
ublob = np.zeros(int(uoff[-1] + (plen[-1] + 15) // 16 * 16), dtype=np.uint8)
for i, p in enumerate(pays):
    ublob[uoff[i]:uoff[i] + len(p)] = np.frombuffer(p, dtype=np.uint8)
gran = ((in_len + 15) // 16).astype(np.int64)
gstart = np.concatenate(([0], np.cumsum(gran)[:-1]))
u16 = ublob.view(np.dtype((np.void, 16)))
h_in = np.empty(total_in // 16, dtype=u16.dtype)
for lo in range(0, n, 8192):
    hi = min(n, lo + 8192)
    g0, g1 = int(gstart[lo]), int(gstart[hi - 1] + gran[hi - 1])
    src = np.repeat(uoff[pick[lo:hi]] // 16 - gstart[lo:hi], gran[lo:hi]) + np.arange(g0, g1)
    h_in[g0:g1] = u16[src]
d_in = torch.from_numpy(h_in.view(np.uint8)).to(dev)
torch.cuda.synchronize()
# verify the device input of a sample of entries
bad_in = 0
for e in list(range(0, n, n // 50)) + [n - 1]:
    z = d_in[in_off[e]:in_off[e] + in_len[e]].cpu().numpy().tobytes()
    bad_in += z != pays[pick[e]]
print("total_in", total_in, "bad device inputs in sample:", bad_in, flush=True)
d_in_off = torch.from_numpy(in_off).to(dev)
d_in_len = torch.from_numpy(in_len.astype(np.int32)).to(dev)
d_out = torch.empty(n * size, dtype=torch.uint8, device=dev)
d_out_off = torch.arange(n, dtype=torch.int64, device=dev) * size
d_out_cap = torch.full((n,), size, dtype=torch.int32, device=dev)
want = crcs[pick]
for rep in range(2):
    out_len, in_used, crc, status = mz.inflate_batch(d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    k = mz.u32(crc)
    ol = out_len.cpu().numpy()
    iu = in_used.cpu().numpy()
    okm = (st == 0) & (k == want) & (ol == size) & (iu == in_len)
    print("rep", rep, "ok", int(okm.sum()), "of", n, "status hist", {int(v): int((st == v).sum()) for v in np.unique(st)},
          flush=True)
    badi = np.nonzero(~okm)[0]
    if len(badi):
        print(" first bad", badi[:12], "last bad", badi[-5:], flush=True)
        print(" bad in [0,32768):", int((badi < 32768).sum()), "[32768,65536):", int(((badi >= 32768) & (badi < 65536)).sum()),
              ">=65536:", int((badi >= 65536).sum()), flush=True)
        for e in badi[:4]:
            print("  e", e, "status", st[e], "out_len", ol[e], "in_used", iu[e], "in_len", in_len[e], "crc", hex(k[e]), hex(want[e]),
                  "in_off", in_off[e], flush=True)
            got = d_out[e * size:(e + 1) * size].cpu().numpy().tobytes()
            o = offs[pick[e]][0]
            exp = c[o:o + size]
            firstdiff = next((i for i in range(size) if got[i] != exp[i]), -1)
            print("   first differing output byte:", firstdiff, flush=True)
