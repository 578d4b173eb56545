"""Static schedules of OpenCV's cyclic one-sided Jacobi (JacobiSVDImpl_, modules/core/src/lapack.cpp) for the wave solver of
csrc/svo_epnp_ord_dev.h: n = 3, 4, 5 (the small decompositions of EPnP) and n = 12 (M^T M), floor(n / 2) pairs per step.

JacobiSVDImpl_ visits the pairs (0,1) (0,2) ... (n-2,n-1) one after the other, sweep after sweep.  A pair only touches its two
rows, so two pairs on disjoint rows commute EXACTLY (same operands, same operations, same bits): any execution order that
keeps, for every row, the order in which the pairs touching that row follow each other gives the same result as the sequential
loop.  This script list-schedules the infinite sequence (sweep 0, sweep 1, ...) greedily, finds the point from which the
schedule is periodic, and checks that running the rotations in schedule order - including the pairs of the next sweep that
start before the current one has ended - gives bit-identical rows to the sequential loop on random matrices.

Sweeps overlap: OpenCV stops after the first sweep that rotated nothing.  Pairs of sweep s + 1 executed before sweep s has
ended see rows that sweep s is done with; if sweep s rotated nothing they see exactly what their counterparts of sweep s saw
and rotate nothing either, so stopping at the end of sweep s leaves the state of the sequential loop.

Output: csrc/svo_epnp_ord_tab.h.  Per (step, row) one 16-bit entry:
  bits 0-3 partner row (== the row itself: the row idles in this step), bit 4: the row is the pair's j (second) row,
  bits 5-6 the pair's sweep relative to the period base, bit 7: the pair (n-2, n-1) of sweep (bits 8-9) runs in this step,
  i.e. that sweep is complete after it (set in every row's entry of the step).
"""
import math
import random
import sys


def pairs_of(n):
    return [(i, j) for i in range(n - 1) for j in range(i + 1, n)]


def schedule(n, sweeps):
    PAIRS = pairs_of(n)
    P = len(PAIRS)
    slots = n // 2
    seq = [(s, i, j) for s in range(sweeps) for (i, j) in PAIRS]
    last, deps = {}, []
    for idx, (s, i, j) in enumerate(seq):
        deps.append({last[r] for r in (i, j) if r in last})
        last[i] = last[j] = idx
    done, steps, pending, t = {}, [], list(range(len(seq))), 0
    while pending:
        chosen = []
        for idx in pending[:4 * P]:
            if len(chosen) >= slots:
                break
            if all(d in done and done[d] < t for d in deps[idx]):
                chosen.append(idx)
        for c in chosen:
            done[c] = t
        pending = [p for p in pending if p not in done]
        steps.append(chosen)
        t += 1
    return seq, steps


def cv_hypot(a, b):
    a, b = abs(a), abs(b)
    if a > b:
        b /= a
        return a * math.sqrt(1 + b * b)
    if b > 0:
        a /= b
        return b * math.sqrt(1 + a * a)
    return 0.0


def pair_op(A, W, i, j, m):
    """One (i, j) visit of JacobiSVDImpl_<double>; returns True if it rotated."""
    eps = sys.float_info.epsilon * 10
    Ai, Aj = A[i], A[j]
    a, b, p = W[i], W[j], 0.0
    for k in range(m):
        p += Ai[k] * Aj[k]
    if abs(p) <= eps * math.sqrt(a * b):
        return False
    p *= 2
    beta = a - b
    gamma = cv_hypot(p, beta)
    if beta < 0:
        delta = (gamma - beta) * 0.5
        s = math.sqrt(delta / gamma)
        c = p / (gamma * s * 2)
    else:
        c = math.sqrt((gamma + beta) / (gamma * 2))
        s = p / (gamma * c * 2)
    a = b = 0.0
    for k in range(m):
        t0 = c * Ai[k] + s * Aj[k]
        t1 = -s * Ai[k] + c * Aj[k]
        Ai[k], Aj[k] = t0, t1
        a += t0 * t0
        b += t1 * t1
    W[i], W[j] = a, b
    return True


def sum_sq(r):
    s = 0.0
    for x in r:
        s += x * x
    return s


def run_sequential(A, n, m):
    A = [r[:] for r in A]
    W = [sum_sq(r) for r in A]
    for it in range(30):
        changed = False
        for (i, j) in pairs_of(n):
            changed |= pair_op(A, W, i, j, m)
        if not changed:
            break
    return A, W, it


def run_scheduled(A, n, m, tab, pro, per, spp):
    """tab[t] = list of (srel, i, j); the pair (n-2, n-1) closes its sweep."""
    A = [r[:] for r in A]
    W = [sum_sq(r) for r in A]
    chg = set()
    tt, sbase, nsteps = 0, 0, 0
    while True:
        closing = None
        for (srel, i, j) in tab[tt]:
            s = sbase + srel
            if (i, j) == (n - 2, n - 1):
                closing = s
            if s >= 30:
                continue
            if pair_op(A, W, i, j, m):
                chg.add(s)
        nsteps += 1
        if closing is not None and (closing >= 29 or closing not in chg):
            return A, W, closing, nsteps
        tt += 1
        if tt == pro + per:
            tt = pro
            sbase += spp


def build(n, m, trials, rnd):
    P = len(pairs_of(n))
    seq, steps = schedule(n, 12)
    found = None
    for spp in (1, 2, 3):                      # sweeps per period
        for per in range(1, 60):
            for pro in range(0, 40):
                if pro + 2 * per + 10 > len(steps) - 20:
                    continue
                if all(sorted(steps[t]) == sorted(x - spp * P for x in steps[t + per]) for t in range(pro, len(steps) - per - 25)):
                    found = (pro, per, spp)
                    break
            if found:
                break
        if found:
            break
    assert found, "no period found for n = %d" % n
    pro, per, spp = found
    tab = [[(seq[x][0], seq[x][1], seq[x][2]) for x in steps[t]] for t in range(pro + per)]
    assert max(e[0] for row in tab for e in row) <= 3
    longest = 0
    for trial in range(trials):
        rank = rnd.choice([n, n, max(n - 2, 1)])
        B = [[rnd.gauss(0, 1) * 10 ** rnd.uniform(-2, 3) for _ in range(m)] for _ in range(rank)]
        if n == m:      # symmetric Gram matrix like M^T M (rank deficient now and then)
            A = [[sum(B[r][a] * B[r][b] for r in range(rank)) for b in range(m)] for a in range(n)]
        else:
            A = [[rnd.gauss(0, 1) * 10 ** rnd.uniform(-1, 1) for _ in range(m)] for _ in range(n)]
        As, Ws, it = run_sequential(A, n, m)
        Ap, Wp, sc, nsteps = run_scheduled(A, n, m, tab, pro, per, spp)
        assert sc == it, (n, sc, it)
        assert all(x.hex() == y.hex() for ra, rb in zip(As, Ap) for x, y in zip(ra, rb)), "rows differ (n %d trial %d)" % (n, trial)
        assert all(x.hex() == y.hex() for x, y in zip(Ws, Wp))
        longest = max(longest, nsteps)
    sys.stderr.write("n = %2d: prologue %d + period %d steps (%d sweep(s) per period), %d steps in all; verified on %d matrices, longest run %d steps\n"
                     % (n, pro, per, spp, pro + per, trials, longest))
    return tab, pro, per, spp


def emit(n, tab, pro, per, spp, out):
    out.append("#define EO_TAB%d_PROLOGUE %d" % (n, pro))
    out.append("#define EO_TAB%d_STEPS %d       // prologue + one period" % (n, pro + per))
    out.append("#define EO_TAB%d_SWEEPS %d      // sweeps per period" % (n, spp))
    out.append("static __constant__ unsigned short c_tab%d[EO_TAB%d_STEPS][%d] = {" % (n, n, n))
    for row in tab:
        e = [r for r in range(n)]            # idle: partner = the row itself
        closing = [s for (s, i, j) in row if (i, j) == (n - 2, n - 1)]
        for (s, i, j) in row:
            e[i] = j | (s << 5)
            e[j] = i | (1 << 4) | (s << 5)
        if closing:
            e = [x | 0x80 | (closing[0] << 8) for x in e]
        out.append("  {" + ", ".join("0x%03x" % x for x in e) + "},")
    out.append("};")


def main():
    rnd = random.Random(5)
    out = ["// generated by tools/gen_jacobi_schedule.py - do not edit",
           "// 16-bit entry per (step, row): bits 0-3 partner row (the row itself: idle), bit 4: this row is the pair's second (j) row,",
           "// bits 5-6 the pair's sweep (relative to the period's base), bit 7: the sweep (bits 8-9, relative) is complete after this step"]
    for n, m, trials in ((3, 3, 300), (4, 6, 300), (5, 6, 300), (12, 12, 200)):
        tab, pro, per, spp = build(n, m, trials, rnd)
        emit(n, tab, pro, per, spp, out)
    out.append("#define EO_TAB_TOTAL (EO_TAB3_STEPS * 3 + EO_TAB4_STEPS * 4 + EO_TAB5_STEPS * 5 + EO_TAB12_STEPS * 12)")
    print("\n".join(out))


if __name__ == "__main__":
    main()
