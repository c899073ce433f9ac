"""Reads a PCGX_STRICT_KEYS dump: per row, the runs of equal windows (key:length), keys < 0 as n (no window) / p (pending)."""
import sys
blocks = open(sys.argv[1]).read().split("#\n")
for bi in (int(x) for x in sys.argv[2:]):
    print("iteration", bi)
    for r, line in enumerate(blocks[bi].strip().split("\n")):
        ks = [tuple(map(int, t.split(":"))) for t in line.split()]
        runs = []
        for k, c in ks:
            if runs and k >= 0 and runs[-1][0] == k:
                runs[-1][1] += 1
            else:
                runs.append([k, 1, c])
        print(" row %d: %d runs:" % (r, len(runs)), " ".join(("%x" % k if k >= 0 else "n") + ("*" if (c >> 8) else "") + ":%d" % n for k, n, c in runs))
