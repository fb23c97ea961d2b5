#!/usr/bin/env python3
"""Generator of the hand-scheduled K-slot skewed block-step pass (qe_skew_asm.inc).

Why.  hipcc's schedule of run64_skew<4> keeps the instructions of one block step next to each other: the average distance
from an instruction to the producer of its operands is under two instructions.  On gfx950 an in-order wave pays for that:
tools/valu_rate.hip Part A measures 8.25 / 6.25 / 5.25 / 4.8 cycles per instruction for 1 / 2 / 4 / 8 independent chains in
round-robin -- cost ~ 4.25 + 4 / d cycles for an instruction whose nearest producer is d instructions back -- and two waves
per SIMD just add up (D = 1, w = 2: 4.13 per instruction).  The pass below is the same arithmetic (bit-identical: the
emulator in this file checks it against a plain Myers / Hyyro column loop) issued in an order in which every instruction's
producers are >= DMIN instructions behind it wherever the dependency graph allows, with the carries of the slot boundaries in
registers whose high halves are zero (no v_mov to build the 64-bit addend of v_lshl_add_u64).

What it emits.  One `asm volatile` block per (K, NCOL) = one pass over NCOL text columns of K vertically adjacent 64-row
blocks, slot k one column behind slot k - 1 (run64_skew's own order of cells), on FIXED physical VGPRs:
   inputs   P[k], M[k], A[k] (code plane 0), B[k] (code plane 1)  as 64-bit pairs; T0, T1: text code-plane words of the NCOL
            columns; HP, HM: carry-in words of the top slot (bit c = column c)
   outputs  P[k], M[k] in place; GP, GM: the bottom slot's carry-out bits, MSB-first (bit-reverse = bit c is column c)
The C++ wrapper binds its variables to those registers with "{vN}" constraints (qe_kernels.hip: run64_skew_asm).

    python3 tools/gen_skew_asm.py --check          # emulate every generated pass against the reference recurrence
    python3 tools/gen_skew_asm.py --emit quicked_amd/csrc/qe_skew_asm.inc
"""
import argparse
import random
import sys

MASK32 = 0xFFFFFFFF


def tt(f):
    """truth table of v_bitop3_b32: bit (a << 2 | b << 1 | c) = f(a, b, c)"""
    v = 0
    for i in range(8):
        a, b, c = (i >> 2) & 1, (i >> 1) & 1, i & 1
        if f(a, b, c) & 1:
            v |= 1 << i
    return v


TT_EQ = tt(lambda a, b, c: (1 - a) & (1 - (b ^ c)))          # Eq = ~x & ~(b ^ m1),  x = a ^ m0
TT_PH = tt(lambda a, b, c: a | (1 - (b | c)))                  # M | ~(s | q)   and   Mhs | ~(Xv | Phs)
TT_MH = tt(lambda a, b, c: a & ((b ^ a) | c))                  # P & ((s ^ P) | Eqc)


class Ins:
    __slots__ = ("op", "dst", "src", "imm", "slow", "tag", "reads", "writes", "text")

    def __init__(self, op, dst, src, imm=None, slow=False, tag=""):
        self.op, self.dst, self.src, self.imm, self.slow, self.tag = op, dst, src, imm, slow, tag
        self.reads, self.writes = set(), set()
        for s in src:
            if isinstance(s, tuple) and s[0] == "v":
                self.reads.add(s[1])
            if isinstance(s, tuple) and s[0] == "p":
                self.reads.update((s[1], s[1] + 1))
        if dst[0] == "v":
            self.writes.add(dst[1])
        else:
            self.writes.update((dst[1], dst[1] + 1))


def V(r):
    return ("v", r)


def PAIR(r):
    assert r % 2 == 0
    return ("p", r)


class Layout:
    """fixed physical registers of a K-slot pass, from `base` up"""

    def __init__(self, K, base):
        r = base + (base & 1)
        self.K = K
        self.P, self.M, self.A, self.B = [], [], [], []
        for _ in range(K):
            self.P.append(r); self.M.append(r + 2); r += 4
        for _ in range(K):
            self.A.append(r); self.B.append(r + 2); r += 4
        self.T0, self.T1, self.HP, self.HM = r, r + 1, r + 2, r + 3
        r += 4
        self.GP, self.GM = r, r + 1
        r += 2
        self.first_tmp = r
        # column masks {m0, m1}: K + 1 generations in flight (the slots are K columns deep, one more so that the next column's
        # extraction need not wait for the last reader)
        self.NM = K + 1
        self.MASK = []
        for _ in range(self.NM):
            self.MASK.append(r); r += 2
        # carries between slot k and k + 1: pairs {carry, 0}; two generations
        self.CP, self.CM = [], []
        for k in range(K):          # index k: the carry INTO slot k (k = 0: extracted from HP / HM)
            gen_p, gen_m = [], []
            for _ in range(2):
                gen_p.append(r); gen_m.append(r + 2); r += 4
            self.CP.append(gen_p); self.CM.append(gen_m)
        # per-slot temporaries
        self.X, self.XV, self.S, self.Q, self.MH, self.ECL = [], [], [], [], [], []
        for _ in range(K):
            self.X.append(r); self.XV.append(r + 2); self.S.append(r + 4); self.Q.append(r + 6); self.MH.append(r + 8); r += 10
        for _ in range(K):
            self.ECL.append(r); r += 1
        self.end = r
        self.zero_his = [p + 1 for k in range(K) for g in range(2) for p in (self.CP[k][g], self.CM[k][g])]


TT_AND = tt(lambda a, b, c: a & b)
TT_OR = tt(lambda a, b, c: a | b)
TT_XOR = tt(lambda a, b, c: a ^ b)
VOP3_ALL = False        # experiments: and / or / xor as v_bitop3 (8-byte encodings throughout)
ZERO_G = True           # False: GP / GM are in-out (a caller that runs the pass over successive column groups accumulates them)


def build(K, NCOL, L):
    """instruction list of one pass in program order (slot-major inside a step: exactly run64_skew's order); the scheduler
    reorders it"""
    ins = []
    for r in L.zero_his:
        ins.append(Ins("v_mov_b32", V(r), [0], tag="zero"))
    if ZERO_G:
        ins.append(Ins("v_mov_b32", V(L.GP), [0], tag="zero"))
        ins.append(Ins("v_mov_b32", V(L.GM), [0], tag="zero"))
    for s in range(NCOL + K - 1):
        if s < NCOL:
            m = L.MASK[s % L.NM]
            ins.append(Ins("v_bfe_i32", V(m), [V(L.T0), s, 1], slow=True, tag=f"m0 c{s}"))
            ins.append(Ins("v_bfe_i32", V(m + 1), [V(L.T1), s, 1], slow=True, tag=f"m1 c{s}"))
        for k in range(K - 1, -1, -1):
            c = s - k
            if c < 0 or c >= NCOL:
                continue
            g = c & 1                                   # generation of the carry registers
            m = L.MASK[c % L.NM]
            P, M, A, B = L.P[k], L.M[k], L.A[k], L.B[k]
            X, XV, S, Q, MH, ECL = L.X[k], L.XV[k], L.S[k], L.Q[k], L.MH[k], L.ECL[k]
            CP, CM = L.CP[k][g], L.CM[k][g]
            t = f"k{k} c{c}"
            if k == 0:
                ins.append(Ins("v_bfe_u32", V(CP), [V(L.HP), c, 1], slow=True, tag=t + " cinP"))
                ins.append(Ins("v_bfe_u32", V(CM), [V(L.HM), c, 1], slow=True, tag=t + " cinM"))
            ins.append(Ins("v_xor_b32", V(X), [V(m), V(A)], tag=t + " x"))
            ins.append(Ins("v_xor_b32", V(X + 1), [V(m), V(A + 1)], tag=t + " x"))
            ins.append(Ins("v_bitop3_b32", V(X), [V(X), V(B), V(m + 1)], imm=TT_EQ, tag=t + " E"))
            ins.append(Ins("v_bitop3_b32", V(X + 1), [V(X + 1), V(B + 1), V(m + 1)], imm=TT_EQ, tag=t + " E"))
            ins.append(Ins("v_or_b32", V(XV), [V(X), V(M)], tag=t + " xv"))
            ins.append(Ins("v_or_b32", V(XV + 1), [V(X + 1), V(M + 1)], tag=t + " xv"))
            ins.append(Ins("v_or_b32", V(ECL), [V(X), V(CM)], tag=t + " ecl"))
            ins.append(Ins("v_and_b32", V(S), [V(ECL), V(P)], tag=t + " t"))
            ins.append(Ins("v_and_b32", V(S + 1), [V(X + 1), V(P + 1)], tag=t + " t"))
            ins.append(Ins("v_or_b32", V(Q), [V(ECL), V(P)], tag=t + " q"))
            ins.append(Ins("v_or_b32", V(Q + 1), [V(X + 1), V(P + 1)], tag=t + " q"))
            ins.append(Ins("v_lshl_add_u64", PAIR(S), [PAIR(S), 0, PAIR(P)], slow=True, tag=t + " sum"))
            ins.append(Ins("v_bitop3_b32", V(Q), [V(M), V(S), V(Q)], imm=TT_PH, tag=t + " ph"))
            ins.append(Ins("v_bitop3_b32", V(Q + 1), [V(M + 1), V(S + 1), V(Q + 1)], imm=TT_PH, tag=t + " ph"))
            ins.append(Ins("v_bitop3_b32", V(MH), [V(P), V(S), V(ECL)], imm=TT_MH, tag=t + " mh"))
            ins.append(Ins("v_bitop3_b32", V(MH + 1), [V(P + 1), V(S + 1), V(X + 1)], imm=TT_MH, tag=t + " mh"))
            if k + 1 < K:
                g2 = c & 1                              # slot k + 1 works on column c one step later: same column, same generation
                ins.append(Ins("v_lshrrev_b32", V(L.CP[k + 1][g2]), [31, V(Q + 1)], tag=t + " coutP"))
                ins.append(Ins("v_lshrrev_b32", V(L.CM[k + 1][g2]), [31, V(MH + 1)], tag=t + " coutM"))
            else:
                ins.append(Ins("v_alignbit_b32", V(L.GP), [V(L.GP), V(Q + 1), 31], slow=True, tag=t + " gP"))
                ins.append(Ins("v_alignbit_b32", V(L.GM), [V(L.GM), V(MH + 1), 31], slow=True, tag=t + " gM"))
            ins.append(Ins("v_lshl_add_u64", PAIR(Q), [PAIR(Q), 1, PAIR(CP)], slow=True, tag=t + " phs"))
            ins.append(Ins("v_lshl_add_u64", PAIR(MH), [PAIR(MH), 1, PAIR(CM)], slow=True, tag=t + " mhs"))
            ins.append(Ins("v_bitop3_b32", V(P), [V(MH), V(XV), V(Q)], imm=TT_PH, tag=t + " P'"))
            ins.append(Ins("v_bitop3_b32", V(P + 1), [V(MH + 1), V(XV + 1), V(Q + 1)], imm=TT_PH, tag=t + " P'"))
            ins.append(Ins("v_and_b32", V(M), [V(Q), V(XV)], tag=t + " M'"))
            ins.append(Ins("v_and_b32", V(M + 1), [V(Q + 1), V(XV + 1)], tag=t + " M'"))
    if VOP3_ALL:
        for x in ins:
            if x.op in ("v_and_b32", "v_or_b32", "v_xor_b32"):
                x.imm = {"v_and_b32": TT_AND, "v_or_b32": TT_OR, "v_xor_b32": TT_XOR}[x.op]
                x.op = "v_bitop3_b32"
                x.src = [x.src[0], x.src[1], x.src[0]]
    return ins


def deps(ins):
    """RAW / WAR / WAW edges on physical registers, program order = the order of `ins`.  -> (pred lists, raw-pred lists)"""
    last_write, readers = {}, {}
    pred = [set() for _ in ins]
    raw = [set() for _ in ins]
    for i, x in enumerate(ins):
        for r in x.reads:
            if r in last_write:
                pred[i].add(last_write[r]); raw[i].add(last_write[r])
        for r in x.writes:
            if r in last_write:
                pred[i].add(last_write[r])
            for j in readers.get(r, ()):
                if j != i:
                    pred[i].add(j)
        for r in x.writes:
            last_write[r] = i
            readers[r] = []
        for r in x.reads:
            if r not in x.writes:
                readers.setdefault(r, []).append(i)
    return pred, raw


def schedule(ins, dmin=8, window=400):
    """list scheduling: the oldest ready instruction (program order) whose true producers are >= dmin positions back; if none,
    the ready one whose nearest producer is furthest back.  Only the first `window` unscheduled instructions are candidates
    (keeps a slot's steps from running ahead and the live ranges short)."""
    pred, raw = deps(ins)
    n = len(ins)
    succ = [[] for _ in range(n)]
    npred = [len(p) for p in pred]
    for i, p in enumerate(pred):
        for j in p:
            succ[j].append(i)
    pos = [None] * n
    order = []
    ready = sorted(i for i in range(n) if npred[i] == 0)
    lowest = 0                                              # every index below is scheduled
    while len(order) < n:
        while lowest < n and pos[lowest] is not None:
            lowest += 1
        here = len(order)
        best, best_d = None, -1
        for i in ready:
            if i > lowest + window:
                break
            d = min((here - pos[j] for j in raw[i]), default=1 << 30)
            if d >= dmin:
                best = i
                break
            if d > best_d:
                best, best_d = i, d
        ready.remove(best)
        pos[best] = here
        order.append(best)
        for j in succ[best]:
            npred[j] -= 1
            if npred[j] == 0:
                # keep `ready` sorted by program order
                lo, hi = 0, len(ready)
                while lo < hi:
                    mid = (lo + hi) // 2
                    if ready[mid] < j:
                        lo = mid + 1
                    else:
                        hi = mid
                ready.insert(lo, j)
    return [ins[i] for i in order]


def model_cycles(seq, waves=2):
    """cycles per SIMD of `waves` waves running `seq` under the issue law of tools/valu_rate.hip Part A: a wave issues an
    instruction 4.25 + 4 / d cycles after its previous one (d = distance to the nearest producer; no producer in the block:
    4.25); an instruction of the slow class (v_lshl_add_u64, v_bfe, v_alignbit) holds the pipe ~4.45 cycles, a fast one ~2.1."""
    last_write = {}
    wave, pipe, dist_hist = 0.0, 0.0, {}
    for i, x in enumerate(seq):
        d = min((i - last_write[r] for r in x.reads if r in last_write), default=None)
        for r in x.writes:
            last_write[r] = i
        c = 4.25 + (4.0 / d if d else 0.0)
        wave += c
        pipe += 4.45 if x.slow else 2.1
        k = min(d, 9) if d else 0
        dist_hist[k] = dist_hist.get(k, 0) + 1
    return max(wave / waves, pipe), wave, pipe, dist_hist


# -----------------------------------------------------------------------------------------------------------------------
# emulation: the generated sequence on 32-bit registers against the plain column loop (bpm_commons.h:49-68)
# -----------------------------------------------------------------------------------------------------------------------
def emulate(seq, regs):
    def rd(s):
        if isinstance(s, tuple):
            if s[0] == "v":
                return regs.get(s[1], 0xDEADBEEF)
            return regs.get(s[1], 0xDEADBEEF) | (regs.get(s[1] + 1, 0xDEADBEEF) << 32)
        return s
    for x in seq:
        a = [rd(s) for s in x.src]
        if x.op == "v_mov_b32":
            v = a[0] & MASK32
        elif x.op == "v_xor_b32":
            v = a[0] ^ a[1]
        elif x.op == "v_or_b32":
            v = a[0] | a[1]
        elif x.op == "v_and_b32":
            v = a[0] & a[1]
        elif x.op == "v_lshrrev_b32":
            v = a[1] >> a[0]
        elif x.op == "v_bfe_u32":
            v = (a[0] >> a[1]) & ((1 << a[2]) - 1)
        elif x.op == "v_bfe_i32":
            v = MASK32 if (a[0] >> a[1]) & 1 else 0
            assert a[2] == 1
        elif x.op == "v_alignbit_b32":
            v = (((a[0] << 32) | a[1]) >> a[2]) & MASK32
        elif x.op == "v_bitop3_b32":
            v = 0
            for bit in range(32):
                idx = (((a[0] >> bit) & 1) << 2) | (((a[1] >> bit) & 1) << 1) | ((a[2] >> bit) & 1)
                v |= ((x.imm >> idx) & 1) << bit
        elif x.op == "v_lshl_add_u64":
            v = ((a[0] << a[1]) + a[2]) & 0xFFFFFFFFFFFFFFFF
        else:
            raise ValueError(x.op)
        if x.dst[0] == "v":
            regs[x.dst[1]] = v & MASK32
        else:
            regs[x.dst[1]] = v & MASK32
            regs[x.dst[1] + 1] = (v >> 32) & MASK32


def reference(K, NCOL, P, M, A, B, T0, T1, HP, HM):
    """column by column, block by block (bpm_commons.h:49-68 with Eq from the two code planes); -> P, M, carry-out words"""
    P, M = list(P), list(M)
    oP = oM = 0
    ones = (1 << 64) - 1
    for c in range(NCOL):
        m0 = ones if (T0 >> c) & 1 else 0
        m1 = ones if (T1 >> c) & 1 else 0
        ph_in, mh_in = (HP >> c) & 1, (HM >> c) & 1
        for k in range(K):
            Eq = ~(A[k] ^ m0) & ~(B[k] ^ m1) & ones
            Pv, Mv = P[k], M[k]
            Xv = Eq | Mv
            Eqc = Eq | mh_in
            Xh = ((((Eqc & Pv) + Pv) & ones) ^ Pv) | Eqc
            Ph = (Mv | ~(Xh | Pv)) & ones
            Mh = Pv & Xh
            ph_out, mh_out = Ph >> 63, Mh >> 63
            Ph = ((Ph << 1) | ph_in) & ones
            Mh = ((Mh << 1) | mh_in) & ones
            P[k] = (Mh | ~(Xv | Ph)) & ones
            M[k] = Ph & Xv
            ph_in, mh_in = ph_out, mh_out
        oP |= ph_in << c
        oM |= mh_in << c
    return P, M, oP, oM


def check(K, NCOL, seq, L, trials=60, seed=1):
    rng = random.Random(seed)
    for _ in range(trials):
        P, M, A, B = [], [], [], []
        for k in range(K):
            x, y = rng.getrandbits(64), rng.getrandbits(64)
            P.append(x & ~y); M.append(y & ~x); A.append(rng.getrandbits(64)); B.append(rng.getrandbits(64))
        T0, T1 = rng.getrandbits(NCOL), rng.getrandbits(NCOL)
        h1, h2 = rng.getrandbits(NCOL), rng.getrandbits(NCOL)
        HP, HM = h1 & ~h2, h2 & ~h1
        regs = {}
        for k in range(K):
            for name, val in ((L.P[k], P[k]), (L.M[k], M[k]), (L.A[k], A[k]), (L.B[k], B[k])):
                regs[name] = val & MASK32; regs[name + 1] = val >> 32
        regs[L.T0], regs[L.T1], regs[L.HP], regs[L.HM] = T0, T1, HP, HM
        emulate(seq, regs)
        rP, rM, oP, oM = reference(K, NCOL, P, M, A, B, T0, T1, HP, HM)
        rev = lambda w: int(format(w, f"0{NCOL}b")[::-1], 2)       # noqa: E731  (the pass collects MSB-first)
        for k in range(K):
            assert regs[L.P[k]] | (regs[L.P[k] + 1] << 32) == rP[k], ("P", k)
            assert regs[L.M[k]] | (regs[L.M[k] + 1] << 32) == rM[k], ("M", k)
        assert rev(regs[L.GP]) == oP and rev(regs[L.GM]) == oM, "carry words"
    return True


# -----------------------------------------------------------------------------------------------------------------------
def fmt_operand(s):
    if isinstance(s, tuple):
        return f"v{s[1]}" if s[0] == "v" else f"v[{s[1]}:{s[1] + 1}]"
    return str(s)


def fmt(x):
    ops = ", ".join([fmt_operand(x.dst)] + [fmt_operand(s) for s in x.src])
    if x.op == "v_bitop3_b32":
        return f"v_bitop3_b32 {ops} bitop3:0x{x.imm:02x}"
    return f"{x.op} {ops}"


def emit(path, variants, base, dmin):
    out = ["// GENERATED by tools/gen_skew_asm.py -- do not edit.  Hand-scheduled K-slot skewed block-step passes on fixed VGPRs:",
           "// the arithmetic of run64_skew<K> (qe_kernels.hip), every instruction >= DMIN instructions behind its producers where the",
           "// dependency graph allows (an in-order gfx950 wave pays ~4.25 + 4 / d cycles for an instruction d behind its producer).",
           f"// DMIN = {dmin}; registers from v{base}; python3 tools/gen_skew_asm.py --check emulates every pass against the column loop.",
           ""]
    for K, NCOL in variants:
        L = Layout(K, base)
        seq = schedule(build(K, NCOL, L), dmin)
        check(K, NCOL, seq, L, trials=12)
        simd, wave, pipe, hist = model_cycles(seq)
        bc = K * NCOL
        name = f"QE_SKEW_ASM_K{K}_C{NCOL}"
        out.append(f"// K = {K}, {NCOL} columns: {len(seq)} instructions ({len(seq) / bc:.2f} per block-column), model {simd / bc:.1f} cycles per "
                   f"block-column per SIMD at two waves (wave-bound {wave / 2 / bc:.1f}, pipe-bound {pipe / bc:.1f})")
        out.append(f"#define {name}_FIRST {base + (base & 1)}")
        out.append(f"#define {name}_END {L.end}")
        for nm, regs in (("P", L.P), ("M", L.M), ("A", L.A), ("B", L.B)):
            for k, r in enumerate(regs):
                out.append(f"#define {name}_{nm}{k} \"{{v[{r}:{r + 1}]}}\"")
        for nm, r in (("T0", L.T0), ("T1", L.T1), ("HP", L.HP), ("HM", L.HM), ("GP", L.GP), ("GM", L.GM)):
            out.append(f"#define {name}_{nm} \"{{v{r}}}\"")
        clob = ", ".join(f"\"v{r}\"" for r in range(L.first_tmp, L.end))
        out.append(f"#define {name}_CLOBBERS {clob}")
        out.append(f"#define {name}_TEXT \\")
        for x in seq:
            out.append(f"    \"{fmt(x)}\\n\\t\" \\")
        out.append("    \"\"")
        out.append("")
    with open(path, "w") as f:
        f.write("\n".join(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--emit")
    ap.add_argument("--base", type=int, default=96)
    ap.add_argument("--dmin", type=int, default=8)
    ap.add_argument("--variants", default="4x32,2x32")
    ap.add_argument("--vop3", action="store_true", help="and / or / xor as v_bitop3 (8-byte encodings throughout: what pairs best at two waves per SIMD)")
    args = ap.parse_args()
    global VOP3_ALL
    VOP3_ALL = args.vop3
    variants = [tuple(int(v) for v in s.split("x")) for s in args.variants.split(",")]
    if args.check or not args.emit:
        for K, NCOL in variants:
            L = Layout(K, args.base)
            prog = build(K, NCOL, L)
            for label, seq in (("program order", prog), (f"scheduled (dmin {args.dmin})", schedule(prog, args.dmin))):
                ok = check(K, NCOL, seq, L)
                simd, wave, pipe, hist = model_cycles(seq)
                bc = K * NCOL
                print(f"K={K} C={NCOL} {label:24s}: {len(seq)} instr = {len(seq) / bc:.2f}/bc, regs v{args.base}..v{L.end - 1} ({L.end - args.base}), "
                      f"model {simd / bc:.1f} cyc/bc/SIMD @2 waves (wave {wave / 2 / bc:.1f}, pipe {pipe / bc:.1f}), emulation {'ok' if ok else 'BAD'}; "
                      f"producer distance histogram {dict(sorted(hist.items()))}")
    if args.emit:
        emit(args.emit, variants, args.base, args.dmin)
        print("wrote", args.emit)


if __name__ == "__main__":
    sys.exit(main())
