"""Count the min/max pairs of merge-based selection networks after constant (+inf pad) folding and
dead-comparator elimination -- the two things the compiler does to the fully unrolled HIP code.

Model: slots hold either a real value or the literal +inf pad.  cex(i,j) (i<j) on (real,real) costs
one min + one max; on (inf,real) it is a move; (real,inf)/(inf,inf) is nothing.  A comparator is
kept only if one of its outputs can reach a requested output slot."""
import sys


def oemerge(lo, n, r, out):
    m = r * 2
    if m < n:
        oemerge(lo, n, m, out)
        oemerge(lo + r, n, m, out)
        for i in range(lo + r, lo + n - r, m):
            out.append((i, i + r))
    else:
        out.append((lo, lo + r))


def oesort(lo, n, out):
    if n > 1:
        oesort(lo, n // 2, out)
        oesort(lo + n // 2, n // 2, out)
        oemerge(lo, n, 1, out)


def count(comparators, real, wanted):
    """real: list of bools per slot (True = real value).  Returns (#min, #max) kept."""
    # forward: value ids; each op produces new ids
    cur = list(range(len(real)))            # value id in each slot
    isreal = list(real)
    ops = {}                                # value id -> (kind, src ids)
    nid = len(real)
    for i, j in comparators:
        a, b = isreal[i], isreal[j]
        if a and b:
            lo_id, hi_id = nid, nid + 1
            nid += 2
            ops[lo_id] = ("min", cur[i], cur[j])
            ops[hi_id] = ("max", cur[i], cur[j])
            cur[i], cur[j] = lo_id, hi_id
        elif (not a) and b:                 # inf below a real: swap (move)
            cur[i], cur[j] = cur[j], cur[i]
            isreal[i], isreal[j] = True, False
    need = set()
    stack = [cur[w] for w in wanted]
    while stack:
        v = stack.pop()
        if v in need:
            continue
        need.add(v)
        if v in ops:
            stack.extend(ops[v][1:])
    nmin = sum(1 for v in need if v in ops and ops[v][0] == "min")
    nmax = sum(1 for v in need if v in ops and ops[v][0] == "max")
    return nmin, nmax


if __name__ == "__main__":
    # current kernel: sort 128 (125 real), output 62
    c = []
    oesort(0, 128, c)
    print("sort128 -> rank 62      :", count(c, [s < 125 for s in range(128)], [62]), "of", len(c), "comparators")
    # stage 1: sort 5 (in 8 slots)
    c = []
    oesort(0, 8, c)
    print("sort 5 (8 slots)        :", count(c, [s < 5 for s in range(8)], range(5)))
    # stage 2: five sorted 5-lists (each padded to 8) -> sorted 25 in a 64-slot array: merge levels only
    c = []
    for blk in range(0, 64, 16):
        oemerge(blk, 16, 1, c)
    for blk in range(0, 64, 32):
        oemerge(blk, 32, 1, c)
    oemerge(0, 64, 1, c)
    real = [(s % 8) < 5 and s < 40 for s in range(64)]
    print("merge 5x5 -> sorted 25  :", count(c, real, range(25)))
    # stage 3: five sorted 25-lists (padded to 32) in 256 slots -> rank 62
    c = []
    for blk in range(0, 256, 64):
        oemerge(blk, 64, 1, c)
    for blk in range(0, 256, 128):
        oemerge(blk, 128, 1, c)
    oemerge(0, 256, 1, c)
    real = [(s % 32) < 25 and s < 160 for s in range(256)]
    print("merge 5x25 -> rank 62   :", count(c, real, [62]))
    # alternative stage 3 with lists trimmed (see notes): only the ranks that can still be the median


def two_outputs_per_thread():
    """Thread computes the medians at x and x+1: the 100 shared window elements are sorted once
    (only ranks 37..62 of them can hold either median), each output then sorts its private slab of
    25 and selects rank 62-37 = 25 of (26 shared candidates + 25 private)."""
    c = []
    oesort(0, 128, c)
    shared = count(c, [s < 100 for s in range(128)], range(37, 63))
    c = []
    oesort(0, 32, c)
    slab = count(c, [s < 25 for s in range(32)], range(25))
    # merge 26 (padded to 32) with 25 (padded to 32) -> rank 25
    c = []
    oemerge(0, 64, 1, c)
    real = [(s < 26) or (32 <= s < 57) for s in range(64)]
    sel = count(c, real, [25])
    print("2 outputs/thread: shared sort100->ranks 37..62", shared, " slab sort25", slab, " final select", sel)
    print("   per output: %.0f min/max pairs" % (shared[0] / 2 + slab[0] + sel[0]))


if __name__ == "__main__":
    two_outputs_per_thread()


def merged_slabs():
    """Shared set of an output pair as a merge of four pre-sorted 25-slabs (exchanged through LDS)."""
    c = []
    for blk in range(0, 128, 64):
        oemerge(blk, 64, 1, c)
    oemerge(0, 128, 1, c)
    real = [(s % 32) < 25 for s in range(128)]
    print("merge 4 sorted 25-slabs -> ranks 37..62:", count(c, real, range(37, 63)))


if __name__ == "__main__":
    merged_slabs()
