"""Generates csrc/svo_epnp_ord_asm.h: the step loop of the wave Jacobi engine (jacobi_rows of svo_epnp_ord_dev.h) as ONE
inline-assembly block per column count M in {12, 6, 3} - the compiler's version of the same loop spends more than half of a
step on copies of the loop-carried rows, lane-mask bookkeeping and branches (measured: 675 of 1200 ticks per step with the
arithmetic and the LDS traffic removed).

What a step does is documented where it is used (svo_epnp_ord_dev.h); this file only fixes registers and instruction order.
Wait states follow LLVM's GCNHazardRecognizer for gfx940/gfx950: 1 after a transcendental (v_rcp_f64 / v_rsq_f64) before its
result is read, 2 between a VALU write of an SGPR / VCC and a VALU read of it as a lane mask, 2 between a VALU write of a VGPR
and a DMFMA read of it, 4 between dependent 4x4x4 DMFMAs (SrcC), 6 before a VALU read and 9 before an LDS read of a DMFMA
result.

Registers: the block owns v140..v253 and s60..s71 (clobbers); the row (x0, x1, x2) lives in v150..v155 while the loop runs.
"""

import os
LOOP_ALIGN = int(os.environ.get("JACOBI_LOOP_ALIGN", "6"))    # log2 bytes
LOOP_NOPS = int(os.environ.get("JACOBI_LOOP_NOPS", "0"))      # 4-byte s_nops between the alignment and the loop's first instruction


# ---- fixed registers -------------------------------------------------------------------------------------------------
def pair(lo):
    return "v[%d:%d]" % (lo, lo + 1)


X0, X1, X2, WW = 150, 152, 154, 156
RCP, ERR, REM, QQ, TA, TB, TC = 160, 162, 164, 166, 168, 170, 172
ONE, T0, T1, T2, AB, Y, G, H, RR, D = 180, 182, 184, 186, 188, 190, 192, 194, 196, 198
P, THR, PLO, P2, WP, BETA, HI, LO, GAM, R1, R2, CC, SS, N0, N1, N2 = 200, 202, 204, 208, 210, 212, 214, 216, 218, 220, 222, 224, 226, 228, 230, 232
Q0, Q1 = PLO, PLO + 2
U0, U1, U2, V0, V1, V2 = 140, 142, 144, 146, 148, 174
E, E1, E2, TT, TP, SB, SBE, SBE1, SGN, SWBIT, AP, AW, TMP, TMP2, ROW, QI, ONEI, STEPS, PRO, TSTEP = range(234, 254)
ACT, VALID, ROT, SAVE, CL, ST = "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", "s[68:69]", "s[70:71]"


def v(n):
    return "v%d" % n


def ndiv(dst, a, b):
    """dst = a / b, the compiler's IEEE division without range scaling / fix-up (see ndiv in svo_epnp_ord_dev.h)."""
    r, e, rem = pair(RCP), pair(ERR), pair(REM)
    return [
        "v_rcp_f64 %s, %s" % (r, b),
        "s_nop 0",
        "v_fma_f64 %s, -%s, %s, 1.0" % (e, b, r),
        "v_fma_f64 %s, %s, %s, %s" % (r, r, e, r),
        "v_fma_f64 %s, -%s, %s, 1.0" % (e, b, r),
        "v_fma_f64 %s, %s, %s, %s" % (r, r, e, r),
        "v_mul_f64 %s, %s, %s" % (dst, a, r),
        "v_fma_f64 %s, -%s, %s, %s" % (rem, b, dst, a),
        "v_fma_f64 %s, %s, %s, %s" % (dst, rem, r, dst),
    ]


def nsqrt(dst, x):
    """dst = sqrt(x), x > 0, the compiler's IEEE square root without range scaling (see nsqrt in svo_epnp_ord_dev.h)."""
    y, h, r, d = pair(Y), pair(H), pair(RR), pair(D)
    return [
        "v_rsq_f64 %s, %s" % (y, x),
        "s_nop 0",
        "v_mul_f64 %s, %s, %s" % (dst, x, y),
        "v_mul_f64 %s, %s, 0.5" % (h, y),
        "v_fma_f64 %s, -%s, %s, 0.5" % (r, h, dst),
        "v_fma_f64 %s, %s, %s, %s" % (dst, dst, r, dst),
        "v_fma_f64 %s, %s, %s, %s" % (h, h, r, h),
        "v_fma_f64 %s, -%s, %s, %s" % (d, dst, dst, x),
        "v_fma_f64 %s, %s, %s, %s" % (dst, d, h, dst),
        "v_fma_f64 %s, -%s, %s, %s" % (d, dst, dst, x),
        "v_fma_f64 %s, %s, %s, %s" % (dst, d, h, dst),
    ]


def mfma(dst, b, c):
    return "v_mfma_f64_4x4x4_4b_f64 %s, %s, %s, %s" % (dst, pair(ONE), b, c)


def pair_test(M):
    """P = sum_k mine[k] theirs[k], WW = sum_k mine[k]^2 (= W of my row), WP = sum_k theirs[k]^2 (= W of the partner row) over the A
    columns: three interleaved DMFMA chains (dependent DMFMAs end up 4 wait states apart)."""
    p, wa, wb = pair(P), pair(WW), pair(WP)
    regs = ((T0, U0, V0, X0, Q0), (T1, U1, V1, X1, Q1), (T2, U2, V2, X2, P2))
    nq = {12: 3, 6: 2, 3: 1}[M]
    out = []
    for q in range(nq):      # the squares of my own row need no partner: they go in front of the wait for its arrival
        t, u, w, x, y = regs[q]
        out.append("v_mul_f64 %s, %s, %s" % (pair(u), pair(x), pair(x)))
    out.append("s_waitcnt lgkmcnt(1)")             # the partner's row (requested at the end of the previous step)
    for q in range(nq):
        t, u, w, x, y = regs[q]
        out += ["v_mul_f64 %s, %s, %s" % (pair(t), pair(x), pair(y)), "v_mul_f64 %s, %s, %s" % (pair(w), pair(y), pair(y))]
    if M != 12:          # the partly filled column group: V columns count as +0.0
        t, u, w, x, y = regs[nq - 1]
        out += ["v_mul_f64 %s, %s, %%[mk]" % (pair(r), pair(r)) for r in (t, u, w)]
        out.append("s_nop 0")
    for q in range(nq):
        t, u, w, x, y = regs[q]
        c = (lambda r: "0") if q == 0 else (lambda r: r)
        if q:
            out.append("s_nop 1")
        out += [mfma(wa, pair(u), c(wa)), mfma(wb, pair(w), c(wb)), mfma(p, pair(t), c(p))]
    out.append("s_nop 4")     # 6 wait states between a chain's last DMFMA and the VALU read of its sum: decide() reads WW and WP
                              # first (5 + the DMFMA into P), then P (5 + one VALU instruction)
    return out


def decide():
    """ROT = VALID & !(|p| <= eps sqrt(W[i] W[j])).  Decided on the squares with a margin of 2^-40 (the roundings of either side are
    of the order 2^-52); a pair inside the margin - or with a NaN - takes the exact expression for the whole wave."""
    return [
        "v_mul_f64 %s, %s, %s" % (pair(AB), pair(WW), pair(WP)),
        "v_mul_f64 %s, %s, %s" % (pair(G), pair(P), pair(P)),
        "v_mul_f64 %s, %s, %%[eps2hi]" % (pair(H), pair(AB)),
        "v_mul_f64 %s, %s, %%[eps2lo]" % (pair(RR), pair(AB)),
        "v_cmp_gt_f64 vcc, %s, %s" % (pair(G), pair(H)),            # p^2 well above: rotates
        "v_cmp_lt_f64 %s, %s, %s" % (ST, pair(G), pair(RR)),        # well below: does not
        "s_or_b64 %s, vcc, %s" % (ST, ST),
        "s_andn2_b64 %s, %s, %s" % (ST, VALID, ST),                 # undecided (valid pairs only)
        "s_cbranch_scc0 L_decided_%=",
    ] + nsqrt(pair(THR), pair(AB)) + [
        "v_mul_f64 %s, %s, %%[eps]" % (pair(THR), pair(THR)),
        "v_cmp_nle_f64 vcc, |%s|, %s" % (pair(P), pair(THR)),
        "L_decided_%=:",
        "s_and_b64 %s, vcc, %s" % (ROT, VALID),
    ]


def decode(e, sbe):
    """The pair of the step whose entry is in `e` (sweep base `sbe`): lane mask VALID, partner addresses, sign, sweep bit;
    then the partner's row and W are requested."""
    return [
        "v_and_b32 %s, 15, %s" % (v(TMP), v(e)),
        "v_cmp_ne_u32 vcc, %s, %s" % (v(TMP), v(QI)),
        "s_and_b64 %s, vcc, %s" % (VALID, ACT),
        "v_add_u32 %s, %s, %%[base]" % (v(TMP), v(TMP)),
        "v_cndmask_b32_e64 %s, %s, %s, %s" % (v(TMP), v(ROW), v(TMP), VALID),
        "v_lshl_add_u32 %s, %s, 7, %%[axch]" % (v(AP), v(TMP)),
        "v_and_b32 %s, 0x80000000, %s" % (v(SGN), v(e)),
        "v_bfe_u32 %s, %s, 5, 2" % (v(TMP2), v(e)),
        "v_add_u32 %s, %s, %s" % (v(TMP2), v(TMP2), v(sbe)),
        "v_lshlrev_b32 %s, %s, %s" % (v(SWBIT), v(TMP2), v(ONEI)),
        "ds_read_b128 v[%d:%d], %s" % (PLO, PLO + 3, v(AP)),
        "ds_read_b64 %s, %s offset:16" % (pair(P2), v(AP)),
    ]


def program(M):
    o = []
    a = o.append
    # ---- set-up
    for dst, src in ((X0, "%[x0]"), (X1, "%[x1]"), (X2, "%[x2]")):
        a("v_mov_b64 %s, %s" % (pair(dst), src))
    a("v_mov_b64 %s, 1.0" % pair(ONE))
    a("v_mov_b32 %s, 1" % v(ONEI))
    a("v_mov_b32 %s, %%[e0]" % v(E))
    a("v_mov_b32 %s, %%[e1]" % v(E1))
    a("v_mov_b32 %s, %%[e1]" % v(E2))
    a("v_mov_b32 %s, %%[tt]" % v(TT))
    a("v_mov_b32 %s, %%[tp]" % v(TP))
    a("v_mov_b32 %s, %%[sb]" % v(SB))
    a("v_mov_b32 %s, 0" % v(SBE))
    a("v_mov_b32 %s, %%[sb]" % v(SBE1))
    a("v_and_b32 %s, 0xff, %%[cfg]" % v(STEPS))
    a("v_bfe_u32 %s, %%[cfg], 8, 8" % v(PRO))
    a("v_lshrrev_b32 %s, 16, %%[cfg]" % v(TSTEP))
    a("v_and_b32 %s, 15, %%[lane]" % v(ROW))
    a("v_sub_u32 %s, %s, %%[base]" % (v(QI), v(ROW)))
    a("v_cmp_ne_u32 %s, 0, %%[act]" % ACT)
    a("s_nop 1")
    o += decode(E, SBE)
    a("s_cmp_lg_u64 %s, 0" % ACT)
    a("s_cbranch_scc0 L_done_%=")
    # ---- the step loop
    a(".p2align %d" % LOOP_ALIGN)                    # the loop's address relative to the fetch lines must not depend on the code in front of it
    for _ in range(LOOP_NOPS):
        a("s_nop 0")
    a("L_loop_%=:")
    a("s_waitcnt lgkmcnt(2)")                      # the entry requested a step ago has arrived
    a("v_mov_b32 %s, %s" % (v(E1), v(E2)))
    a("v_add_u32 %s, 1, %s" % (v(TT), v(TT)))
    a("v_add_u32 %s, %s, %s" % (v(TP), v(TP), v(TSTEP)))
    a("v_cmp_eq_u32 vcc, %s, %s" % (v(TT), v(STEPS)))
    a("s_nop 1")
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(TT), v(TT), v(PRO)))
    a("v_cndmask_b32 %s, %s, %%[twrap], vcc" % (v(TP), v(TP)))
    a("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (v(SB), v(SB)))
    a("ds_read_b32 %s, %s" % (v(E2), v(TP)))
    o += pair_test(M)
    o += decide()
    a("s_cbranch_scc0 L_skip_%=")
    # rotation
    a("v_add_f64 %s, %s, %s" % (pair(P), pair(P), pair(P)))
    a("v_add_f64 %s, %s, -%s" % (pair(BETA), pair(WW), pair(WP)))
    a("v_xor_b32 %s, %s, %s" % (v(BETA + 1), v(BETA + 1), v(SGN)))
    a("v_cmp_gt_f64 vcc, |%s|, |%s|" % (pair(P), pair(BETA)))
    a("s_nop 1")
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(HI), v(BETA), v(P)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(HI + 1), v(BETA + 1), v(P + 1)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(LO), v(P), v(BETA)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(LO + 1), v(P + 1), v(BETA + 1)))
    o += ndiv(pair(QQ), "|%s|" % pair(LO), "|%s|" % pair(HI))
    a("v_mul_f64 %s, %s, %s" % (pair(TA), pair(QQ), pair(QQ)))
    a("v_add_f64 %s, %s, 1.0" % (pair(TA), pair(TA)))
    o += nsqrt(pair(TB), pair(TA))
    a("v_mul_f64 %s, |%s|, %s" % (pair(GAM), pair(HI), pair(TB)))
    a("v_add_f64 %s, %s, |%s|" % (pair(TA), pair(GAM), pair(BETA)))
    a("v_add_f64 %s, %s, %s" % (pair(TC), pair(GAM), pair(GAM)))
    o += ndiv(pair(QQ), pair(TA), pair(TC))
    o += nsqrt(pair(R1), pair(QQ))
    a("v_mul_f64 %s, %s, %s" % (pair(TA), pair(GAM), pair(R1)))
    a("v_add_f64 %s, %s, %s" % (pair(TA), pair(TA), pair(TA)))
    o += ndiv(pair(R2), pair(P), pair(TA))
    a("v_cmp_gt_f64 vcc, 0, %s" % pair(BETA))
    a("s_nop 1")
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(SS), v(R2), v(R1)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(SS + 1), v(R2 + 1), v(R1 + 1)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(CC), v(R1), v(R2)))
    a("v_cndmask_b32 %s, %s, %s, vcc" % (v(CC + 1), v(R1 + 1), v(R2 + 1)))
    a("v_xor_b32 %s, %s, %s" % (v(SS + 1), v(SS + 1), v(SGN)))
    for n_, x_, q_, t_ in ((N0, X0, Q0, T0), (N1, X1, Q1, T1), (N2, X2, P2, T2)):
        a("v_mul_f64 %s, %s, %s" % (pair(n_), pair(CC), pair(x_)))
        a("v_mul_f64 %s, %s, %s" % (pair(t_), pair(SS), pair(q_)))
    for n_, t_ in ((N0, T0), (N1, T1), (N2, T2)):
        a("v_add_f64 %s, %s, %s" % (pair(n_), pair(n_), pair(t_)))
    # commit for the rows that rotate
    for x_, n_ in ((X0, N0), (X1, N1), (X2, N2)):
        a("v_cndmask_b32_e64 %s, %s, %s, %s" % (v(x_), v(x_), v(n_), ROT))
        a("v_cndmask_b32_e64 %s, %s, %s, %s" % (v(x_ + 1), v(x_ + 1), v(n_ + 1), ROT))
    a("v_cndmask_b32_e64 %s, 0, %s, %s" % (v(TMP), v(SWBIT), ROT))
    a("v_or_b32 %%[chg], %%[chg], %s" % v(TMP))
    a("s_mov_b64 %s, exec" % SAVE)
    a("s_mov_b64 exec, %s" % ROT)
    a("ds_write2_b64 %%[amine], %s, %s offset1:1" % (pair(X0), pair(X1)))
    a("ds_write_b64 %%[amine], %s offset:16" % pair(X2))
    a("s_mov_b64 exec, %s" % SAVE)
    a("L_skip_%=:")
    # the next step's pair (requested before the bookkeeping: more instructions between the request and the use)
    o += decode(E1, SBE1)
    # sweep bookkeeping
    a("v_and_b32 %s, 0x80, %s" % (v(TMP), v(E)))
    a("v_cmp_ne_u32 vcc, 0, %s" % v(TMP))
    a("s_and_b64 %s, vcc, %s" % (CL, ACT))
    a("s_cbranch_scc0 L_open_%=")
    a("v_bfe_u32 %s, %s, 8, 2" % (v(TMP), v(E)))
    a("v_add_u32 %s, %s, %s" % (v(TMP), v(TMP), v(SBE)))          # the sweep that is complete
    a("v_lshrrev_b32 %s, %s, %%[chg]" % (v(TMP2), v(TMP)))
    a("v_and_b32 %s, 1, %s" % (v(TMP2), v(TMP2)))
    a("v_cmp_ne_u32 vcc, 0, %s" % v(TMP2))
    a("s_and_b64 %s, vcc, %s" % (ST, CL))                        # rows (of closing problems) that rotated in it
    a("v_mov_b32 %s, s70" % v(TMP2))
    a("v_and_b32 %s, %s, %%[pm]" % (v(TMP2), v(TMP2)))
    a("v_cmp_eq_u32 vcc, 0, %s" % v(TMP2))                       # no row of my problem did: JacobiSVDImpl_ stops
    a("v_cmp_le_u32 %s, 24, %s" % (ST, v(TMP)))                   # 25 sweeps: not reproduced here (flag)
    a("s_or_b64 vcc, vcc, %s" % ST)
    a("s_and_b64 vcc, vcc, %s" % CL)
    a("s_andn2_b64 %s, %s, vcc" % (ACT, ACT))
    a("s_and_b64 %s, %s, %s" % (ST, ST, CL))
    a("v_cndmask_b32_e64 %s, 0, 1, %s" % (v(TMP), ST))
    a("v_or_b32 %%[flag], %%[flag], %s" % v(TMP))
    a("s_and_b64 %s, %s, %s" % (VALID, VALID, ACT))      # a problem that has just stopped takes no part in the next step
    a("L_open_%=:")
    a("v_mov_b32 %s, %s" % (v(E), v(E1)))
    a("v_mov_b32 %s, %s" % (v(SBE), v(SBE1)))
    a("v_mov_b32 %s, %s" % (v(SBE1), v(SB)))
    a("s_cmp_lg_u64 %s, 0" % ACT)
    a("s_cbranch_scc1 L_loop_%=")
    a("L_done_%=:")
    a("s_waitcnt lgkmcnt(0)")
    for dst, src in (("%[x0]", X0), ("%[x1]", X1), ("%[x2]", X2)):
        a("v_mov_b64 %s, %s" % (dst, pair(src)))
    return [l.replace('%%[', '%[') for l in o]


def main():
    print("// generated by tools/gen_jacobi_asm.py - do not edit")
    print("// The step loop of jacobi_rows (svo_epnp_ord_dev.h) for M = 12, 6, 3 columns of A; registers v140..v253, s60..s71.")
    for M in (12, 6, 3):
        lines = program(M)
        print("#define EO_JACOBI_ASM_%d \\" % M)
        for i, l in enumerate(lines):
            end = " \\" if i + 1 < len(lines) else ""
            print('  "%s\\n\\t"%s' % (l, end))
        print("")
    clob = ", ".join('"v%d"' % i for i in range(140, 254)) + ", " + ", ".join('"s%d"' % i for i in range(60, 72)) + ', "vcc", "scc", "memory"'
    print("#define EO_JACOBI_ASM_CLOBBERS " + clob)


if __name__ == "__main__":
    main()
