"""Writes the streams tests/study/lzma_lockstep.c replays: ZIP method-14 payloads of the bench's config-4 text (the order-2
Markov expansion of the corpus, oracle/make_corpus.py) at preset 6, 256 KiB each.  python tests/study/lzma_lockstep.py <dir> [n]"""
import lzma
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import synth  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
os.makedirs(out, exist_ok=True)
datas = synth.markov_entries(n, 256 << 10, seed=4)       # what bench.py's config 4 decodes (1 MiB there)
ratio = 0.0
for i, d in enumerate(datas):
    raw = lzma.compress(d, format=lzma.FORMAT_ALONE, filters=[dict(id=lzma.FILTER_LZMA1, preset=6)])
    open(os.path.join(out, "s%02d.bin" % i), "wb").write(bytes([5, 2, 5, 0]) + raw[:5] + raw[13:])
    ratio += (len(raw) - 4) / len(d) / n
print("wrote %d streams to %s, ratio %.3f" % (n, out, ratio))
