#!/usr/bin/env python3
"""Static instruction mix of the loops of one function of an existing assembly file (hipcc -S output): per loop its depth and
the VALU / SALU / LDS / vector-memory / branch / wait instructions it holds, EXCLUSIVE of the loops nested inside it, so
that counts can be multiplied by trip counts known from the host emulation (tests/study/k1_steps.py).

    python profiles/asm_loops.py /tmp/k.s mz_chase_emit [min_instructions=20]
"""
import re
import sys


def main():
    path, fn = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    text = open(path).read().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZL?\d+%s\w*:" % re.escape(fn), l))
    end = next(i for i in range(start, len(text)) if re.match(r"^\.Lfunc_end\d+:", text[i]))  # (a function may return in several places)
    body = text[start:end + 1]
    label_at = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"Loop Header: Depth=(\d+)", l)
        if not m:
            continue
        lab_i = max(j for j in label_at if j <= i)
        lab = re.match(r"^\.(LBB\d+_\d+):", body[lab_i]).group(1)[1:]
        named = [j for j, t in enumerate(body) if j > lab_i and re.search(r"(Header=|Parent Loop )%s\b" % re.escape(lab), t)]
        last = max(named) if named else lab_i
        last_block = max(j for j in label_at if j <= last)
        nxt = [j for j in label_at if j > last_block]
        loops.append([lab, int(m.group(1)), lab_i, (nxt[0] - 1) if nxt else len(body) - 1])

    def is_op(t):
        t = t.strip()
        return t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":") and not re.match(r"^\.?\w+:", t)

    owner = [None] * len(body)  # innermost loop of every line
    for k, (lab, depth, a, b) in enumerate(loops):
        for j in range(a, b + 1):
            if owner[j] is None or loops[owner[j]][1] < depth:
                owner[j] = k
    cats = ["valu", "salu", "lds", "vmem", "branch", "wait", "scratch"]

    def cat(op):
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            return "branch"
        if op.startswith("s_waitcnt"):
            return "wait"
        if op.startswith("v_"):
            return "valu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith("scratch_"):
            return "scratch"
        if re.match(r"global_|buffer_|flat_", op):
            return "vmem"
        return "salu"

    rows = {k: dict.fromkeys(cats, 0) for k in range(len(loops))}
    rows[None] = dict.fromkeys(cats, 0)
    for j, t in enumerate(body):
        if is_op(t):
            rows[owner[j]][cat(t.split()[0])] += 1
    print("%s: %d instructions" % (fn, sum(sum(r.values()) for r in rows.values())))
    print("%-12s %5s %6s %6s | " % ("loop", "depth", "line", "total") + " ".join("%7s" % c for c in cats))
    r = rows[None]
    print("%-12s %5s %6s %6d | " % ("(no loop)", "", "", sum(r.values())) + " ".join("%7d" % r[c] for c in cats))
    for k, (lab, depth, a, b) in enumerate(loops):
        r = rows[k]
        if sum(r.values()) < floor:
            continue
        print("%-12s %5d %6d %6d | " % (lab, depth, a, sum(r.values())) + " ".join("%7d" % r[c] for c in cats))


if __name__ == "__main__":
    main()
