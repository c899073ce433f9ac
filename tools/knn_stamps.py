"""Analysis of the tile search's per-workgroup stamps (PCGX_KNN_TILE_STATS=1 PCGX_KNN_TILE_STAMPS=<file>): how long a
workgroup's set-up and search take, how many are resident per CU over the launch."""
import sys, collections
rows = [l.split() for l in open(sys.argv[1])]
it = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4], 16)) for r in rows if int(r[3]) > 0]
t0 = min(r[1] for r in it)
setup = sorted((r[2] - r[1]) / 100.0 for r in it if r[2])
search = sorted((r[3] - r[2]) / 100.0 for r in it if r[2])
span = (max(r[3] for r in it) - t0) / 100.0
print("workgroups", len(it), "launch span %.1f us" % span)
print("set-up us: median %.1f p90 %.1f max %.1f" % (setup[len(setup) // 2], setup[len(setup) * 9 // 10], setup[-1]))
print("search us: median %.1f p90 %.1f max %.1f" % (search[len(search) // 2], search[len(search) * 9 // 10], search[-1]))
# residency per (xcc, se, cu)
def cu_of(h):
    hw, xcc = h & 0xffffffff, h >> 32
    return (xcc & 0xf, (hw >> 13) & 0x7, (hw >> 8) & 0xf)   # xcc, se_id, cu_id (gfx9 HW_ID layout)
per = collections.defaultdict(list)
for r in it:
    per[cu_of(r[4])].append((r[1], r[3]))
print("CUs seen", len(per), "workgroups per CU: min %d max %d" % (min(len(v) for v in per.values()), max(len(v) for v in per.values())))
tot = 0.0
for v in per.values():
    tot += sum(e - s for s, e in v) / 100.0
print("mean resident workgroups per CU over the span: %.2f" % (tot / len(per) / span))
starts = sorted((r[1] - t0) / 100.0 for r in it)
print("starts us: first %.1f, 25%% %.1f, 50%% %.1f, 75%% %.1f, last %.1f" % (starts[0], starts[len(starts) // 4], starts[len(starts) // 2], starts[len(starts) * 3 // 4], starts[-1]))
