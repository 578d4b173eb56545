"""Static schedule of OpenCV's cyclic one-sided Jacobi (JacobiSVDImpl_, lapack.cpp) for a 12 x 12 matrix on FOUR pair slots.

JacobiSVDImpl_ visits the pairs (0,1) (0,2) ... (10,11) one after the other, sweep after sweep.  A pair only touches its two
rows, so two pairs on disjoint rows commute EXACTLY (same operands, same operations, same bits): any execution order that
keeps, for every row, the order in which the pairs touching that row follow each other gives the same result as the sequential
loop.  This script list-schedules the infinite sequence (sweep 0, sweep 1, ...) greedily on four slots, checks that the
schedule becomes periodic (6 steps of prologue, then 33 steps = two sweeps = 132 pairs, all four slots busy) and that running
the rotations in schedule order - including the pairs of the next sweep that start before the current one has ended - gives
bit-identical rows to the sequential loop on random matrices, then prints the table for csrc/svo_epnp_ord_dev.h.

Sweeps overlap: OpenCV stops after the first sweep that rotated nothing.  Pairs of sweep s + 1 executed before sweep s has
ended see rows that sweep s is done with; if sweep s rotated nothing they see exactly what their counterparts of sweep s saw
and rotate nothing either, so stopping at the end of sweep s leaves the state of the sequential loop.
"""
import math
import random
import sys

N, SLOTS = 12, 4
PAIRS = [(i, j) for i in range(N - 1) for j in range(i + 1, N)]
P = len(PAIRS)


def schedule(sweeps):
    seq = [(s, i, j) for s in range(sweeps) for (i, j) in PAIRS]
    last, deps = {}, []
    for idx, (s, i, j) in enumerate(seq):
        deps.append({last[r] for r in (i, j) if r in last})
        last[i] = last[j] = idx
    done, steps, pending, t = {}, [], list(range(len(seq))), 0
    while pending:
        chosen = []
        for idx in pending[:4 * P]:
            if len(chosen) >= SLOTS:
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


def pair_op(A, W, i, j):
    """One (i, j) visit of JacobiSVDImpl_<double>; returns True if it rotated."""
    eps = sys.float_info.epsilon * 10
    Ai, Aj = A[i], A[j]
    a, b, p = W[i], W[j], 0.0
    for k in range(N):
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
    for k in range(N):
        t0 = c * Ai[k] + s * Aj[k]
        t1 = -s * Ai[k] + c * Aj[k]
        Ai[k], Aj[k] = t0, t1
        a += t0 * t0
        b += t1 * t1
    W[i], W[j] = a, b
    return True


def run_sequential(A):
    A = [r[:] for r in A]
    W = [sum_sq(r) for r in A]
    for it in range(30):
        changed = False
        for (i, j) in PAIRS:
            changed |= pair_op(A, W, i, j)
        if not changed:
            break
    return A, W, it


def sum_sq(r):
    s = 0.0
    for x in r:
        s += x * x
    return s


def run_scheduled(A, tab, close, pro, per):
    A = [r[:] for r in A]
    W = [sum_sq(r) for r in A]
    chg = set()
    tt, sbase, nsteps = 0, 0, 0
    while True:
        for e in tab[tt]:
            if e is None:
                continue
            srel, i, j = e
            s = sbase + srel
            if s >= 30:
                continue
            if pair_op(A, W, i, j):
                chg.add(s)
        nsteps += 1
        if close[tt] is not None:
            sc = sbase + close[tt]
            if sc >= 29 or sc not in chg:
                return A, W, sc, nsteps
        tt += 1
        if tt == pro + per:
            tt = pro
            sbase += 2


def main():
    seq, steps = schedule(10)
    pro, per = 6, 33
    for t in range(pro, len(steps) - per - 40):
        assert sorted(steps[t]) == sorted(x - 2 * P for x in steps[t + per]), "not periodic at step %d" % t
    tab, close = [], []
    for t in range(pro + per):
        row = [(seq[x][0], seq[x][1], seq[x][2]) for x in steps[t]]
        row += [None] * (SLOTS - len(row))
        tab.append(row)
        cl = [seq[x][0] for x in steps[t] if (seq[x][1], seq[x][2]) == (N - 2, N - 1)]
        close.append(cl[0] if cl else None)
    # every sweep's pairs are all done when its (10, 11) pair runs
    for s in range(6):
        t_close = [t for t, st in enumerate(steps) if s * P + P - 1 in st][0]
        assert all(t <= t_close for t, st in enumerate(steps) for x in st if x // P == s)
    rnd = random.Random(5)
    worst = 0
    for trial in range(300):
        rank = rnd.choice([12, 10, 10, 10, 7])
        B = [[rnd.gauss(0, 1) * 10 ** rnd.uniform(-2, 3) for _ in range(12)] for _ in range(rank)]
        A = [[sum(B[r][a] * B[r][b] for r in range(rank)) for b in range(12)] for a in range(12)]
        As, Ws, it = run_sequential(A)
        Ap, Wp, sc, nsteps = run_scheduled(A, tab, close, pro, per)
        assert sc == it, (sc, it)
        assert all(x.hex() == y.hex() for ra, rb in zip(As, Ap) for x, y in zip(ra, rb)), "rows differ (trial %d)" % trial
        assert all(x.hex() == y.hex() for x, y in zip(Ws, Wp))
        worst = max(worst, nsteps)
    sys.stderr.write("schedule verified on 300 matrices (bit-identical rows and W, same final sweep); longest run %d steps\n" % worst)
    print("// generated by tools/gen_jacobi_schedule.py - do not edit")
    print("// entry: bits 0-3 row i, bits 4-7 row j, bits 8-9 sweep (relative to the period's base), 0xffff: slot idle")
    print("#define EO_J12_PROLOGUE %d" % pro)
    print("#define EO_J12_STEPS %d      // prologue + one period (two sweeps)" % (pro + per))
    print("static __constant__ unsigned short c_j12_tab[EO_J12_STEPS][4] = {")
    for row in tab:
        print("  {" + ", ".join("0x%04x" % (0xffff if e is None else (e[1] | e[2] << 4 | e[0] << 8)) for e in row) + "},")
    print("};")
    print("// sweep (relative) whose last pair (10, 11) runs in this step, -1: none")
    print("static __constant__ int c_j12_close[EO_J12_STEPS] = {" + ", ".join(str(-1 if c is None else c) for c in close) + "};")


if __name__ == "__main__":
    main()
