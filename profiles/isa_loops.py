#!/usr/bin/env python3
"""Static view of a kernel's loops, from the compiler's own assembly (no GPU needed): for every loop of the chosen kernel its
nesting depth, instruction count and how many of those are branches, waits, LDS and global-memory operations, moves and
selects -- the numbers the "VALU diet" of K1's walk loops is steered by between GPU runs (285 instructions per
decode step).  The -D flags are those of the default `make`; extra ones may be appended.

    python profiles/isa_loops.py [kernel=k_inflate_batch] [min_instructions=150] [-DMZ_TOKEN_SELECT=1 ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from resource_usage import make_flags, SRC  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    kernel = args[0] if args else "k_inflate_batch"
    floor = int(args[1]) if len(args) > 1 else 150
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, "-I" + os.path.join(ROOT, "include")] + make_flags() + extra + [SRC]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read().splitlines()
    # the kernel's body: from its label to s_endpgm
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\d+%s\w*:" % re.escape(kernel), l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end + 1]
    # loops: a header comment "=> This (Inner) Loop Header: Depth=N" follows the loop's label; every other block of the loop
    # (and of the loops inside it) is annotated "in Loop: Header=BBx_y" / "Parent Loop BBx_y": the loop runs from its
    # label to the end of the last block that names it
    label_at = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"Loop Header: Depth=(\d+)", l)
        if not m:
            continue
        lab_i = max(j for j in label_at if j <= i)
        lab = re.match(r"^\.(LBB\d+_\d+):", body[lab_i]).group(1)[1:]  # "BB0_898"
        named = [j for j, t in enumerate(body) if j > lab_i and re.search(r"(Header=|Parent Loop )%s\b" % re.escape(lab), t)]
        if not named:
            continue
        last_block = max(j for j in label_at if j <= max(named))
        nxt = [j for j in label_at if j > last_block]
        loops.append(("." + "L" + lab, int(m.group(1)), lab_i, (nxt[0] - 1) if nxt else len(body) - 1))

    def is_op(t):
        t = t.strip()
        return t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":")

    print("%s (%s): %d instructions" % (kernel, " ".join(extra) or "default build", sum(1 for t in body if is_op(t))))
    print("%-12s %5s %7s %8s %6s %5s %7s %7s %6s %7s" % ("loop", "depth", "instr", "branches", "waits", "lds", "global", "scratch", "moves", "selects"))
    for lab, depth, a, b in loops:
        ops = [t.split()[0] for t in body[a:b + 1] if is_op(t)]
        if len(ops) < floor:
            continue
        c = lambda p: sum(1 for o in ops if re.match(p, o))  # noqa: E731
        print("%-12s %5d %7d %8d %6d %5d %7d %7d %6d %7d" % (lab, depth, len(ops), c(r"s_cbranch"), c(r"s_waitcnt"), c(r"ds_"), c(r"global_|buffer_|flat_"),
                                                              c(r"scratch_"), c(r"v_mov"), c(r"v_cndmask")))


if __name__ == "__main__":
    main()
