import numpy as np
blocks = open("gpurun_out/rowdiag.txt").read().split("#\n")
for bi in range(len(blocks)-1):
    rows = [list(map(int, l.split())) for l in blocks[bi].strip().split("\n")]
    a = np.array(rows, dtype=np.uint64)
    walks=[(int(a[r,15])-int(a[r,11]))/100 for r in range(9)]
    print("iteration %d: shader clock while the rows were walked: %s MHz" % (bi, " ".join("%.0f" % (int(a[r,10]) / max(walks[r], 1e-9)) for r in range(8))))
    if max(walks) < 30: continue
    print("iteration %d" % bi)
    for r in range(9):
        w14 = int(a[r, 14])
        print("  row %d: in kernel %.1f us (records in after %.1f, walk from %.1f); run pieces %d, failed %d, records failed %d, table look-ups %d" % (
            r, walks[r], (int(a[r, 12]) - int(a[r, 11])) / 100, (int(a[r, 13]) - int(a[r, 11])) / 100,
            w14 & 0xffff, (w14 >> 16) & 0xffff, (w14 >> 32) & 0xffff, w14 >> 48))
