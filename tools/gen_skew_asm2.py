#!/usr/bin/env python3
"""Generator of the K-slot skewed block-step pass WITHOUT quarter-rate instructions (qe_skew_asm.inc).

What was measured first (tools/skew_asm_bench.hip, tools/bin/mix_bench; profiles/r06_b_*):
  * the ORDER of the instructions of the pass does not matter on gfx950: hipcc's schedule, program order and a list schedule
    that keeps every instruction >= 8 or >= 32 instructions behind its producers run within 2 % of one another;
  * what matters at two waves per SIMD is the instruction CLASS mix.  Synthetic streams without any dependency: full-rate
    instructions alone pair up across the two waves (2.1 cycles per instruction for VOP2, 2.6 for VOP3 encodings); a
    quarter-rate instruction (v_lshl_add_u64, v_bfe, v_alignbit, v_lshlrev: 4.45 cycles) takes its issue window alone AND
    unpairs the full-rate instructions around it: four full-rate + one quarter-rate run at 3.44 (VOP3) / 4.05 (VOP2) cycles
    per instruction instead of the 2.6-3.0 their rates add up to.  run64_skew<4> has 4.7 quarter-rate instructions per
    block-column (the 64-bit sum, the two "<< 1 | carry" shifts, the bit extracts and collects at the pass's edges).
So this pass has none in its loop:
  * the 64-bit sum is v_add_co_u32 + v_addc_co_u32 through VCC;
  * "(x << 1) | carry-in" is x + x + carry-in: two v_addc_co_u32 whose carry-IN comes from an SGPR pair (a lane mask) and
    whose carry-OUT -- bit 63 of x, exactly the block's PHout / MHout -- goes to the SGPR pair the slot below reads: no
    extraction, no 64-bit addend pair;
  * column masks are the sign of a running bit-reversed text word (v_ashrrev_i32 31; the word doubles every column), the
    top slot's carry-ins v_cmp_gt_i32 on running bit-reversed carry words, the bottom slot's carry-outs are collected
    MSB-first by g + g + carry (v_addc_co_u32);
  * MHin is also needed as a value (Eq | MHin in bit 0): one v_lshrrev_b32 per block step.
28.5 full-rate instructions per block-column instead of 22.4 + 4.7.  Bit-identical to run64_skew: the emulator below checks
every generated pass against the plain column loop (bpm_commons.h:49-68), tools/skew_asm_bench.hip on the GPU against
run64_multi on 2 M random passes.

    python3 tools/gen_skew_asm2.py --check
    python3 tools/gen_skew_asm2.py --emit quicked_amd/csrc/qe_skew_asm.inc
"""
import argparse
import random
import sys

MASK32 = 0xFFFFFFFF


def tt(f):
    v = 0
    for i in range(8):
        a, b, c = (i >> 2) & 1, (i >> 1) & 1, i & 1
        if f(a, b, c) & 1:
            v |= 1 << i
    return v


TT_EQ = tt(lambda a, b, c: (1 - a) & (1 - (b ^ c)))          # Eq = ~x & ~(b ^ m1),  x = a ^ m0
TT_PH = tt(lambda a, b, c: a | (1 - (b | c)))                  # M | ~(s | q)   and   Mhs | ~(Xv | Phs)
TT_MH = tt(lambda a, b, c: a & ((b ^ a) | c))                  # P & ((s ^ P) | Eqc)
TT_AND = tt(lambda a, b, c: a & b)
TT_OR = tt(lambda a, b, c: a | b)
TT_XOR = tt(lambda a, b, c: a ^ b)

VCC = "vcc"


def V(r):
    return ("v", r)


def S(r):
    return ("s", r)          # the SGPR pair s[r:r+1] (a lane mask)


class Ins:
    __slots__ = ("op", "dst", "src", "imm", "tag", "reads", "writes", "sdst", "ssrc")

    def __init__(self, op, dst, src, imm=None, tag="", sdst=None, ssrc=None):
        """dst / src: VGPR operands and immediates; sdst / ssrc: the carry-out / carry-in mask (an SGPR pair or VCC)"""
        self.op, self.dst, self.src, self.imm, self.tag, self.sdst, self.ssrc = op, dst, src, imm, tag, sdst, ssrc
        self.reads, self.writes = set(), set()
        for s in src:
            if isinstance(s, tuple):
                self.reads.add(s)
        if ssrc is not None:
            self.reads.add(ssrc)
        if dst is not None:
            self.writes.add(dst)
        if sdst is not None:
            self.writes.add(sdst)


class Layout:
    def __init__(self, K, vbase, sbase):
        r = vbase + (vbase & 1)
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
        self.NM = K + 1
        self.MASK = []
        for _ in range(self.NM):
            self.MASK.append(r); r += 2
        self.CMV = []                    # MHin of slot k as a value (0 / 1)
        for _ in range(K):
            self.CMV.append(r); r += 1
        r += r & 1
        self.X, self.XV, self.S, self.Q, self.MH, self.ECL = [], [], [], [], [], []
        for _ in range(K):
            self.X.append(r); self.XV.append(r + 2); self.S.append(r + 4); self.Q.append(r + 6); self.MH.append(r + 8); r += 10
        for _ in range(K):
            self.ECL.append(r); r += 1
        self.vend = r
        s = sbase + (sbase & 1)
        self.SP, self.SM = [], []        # carry masks INTO slot k (k = K: out of the bottom slot)
        for _ in range(K + 1):
            self.SP.append(s); self.SM.append(s + 2); s += 4
        self.SDUMMY = s
        s += 2
        self.sbase, self.send = sbase + (sbase & 1), s


def build(K, NCOL, L, vop3_all=False, zero_g=True):
    ins = []
    a = ins.append
    # running bit-reversed words: bit 31 is the current column's bit, the word doubles after every column
    for r in (L.T0, L.T1, L.HP, L.HM):
        a(Ins("v_bfrev_b32", V(r), [V(r)], tag="rev"))
    if zero_g:
        a(Ins("v_mov_b32", V(L.GP), [0], tag="zero"))
        a(Ins("v_mov_b32", V(L.GM), [0], tag="zero"))
    for s in range(NCOL + K - 1):
        if s < NCOL:
            m = L.MASK[s % L.NM]
            a(Ins("v_ashrrev_i32", V(m), [31, V(L.T0)], tag=f"m0 c{s}"))
            a(Ins("v_ashrrev_i32", V(m + 1), [31, V(L.T1)], tag=f"m1 c{s}"))
            a(Ins("v_add_u32", V(L.T0), [V(L.T0), V(L.T0)], tag="t0 <<= 1"))
            a(Ins("v_add_u32", V(L.T1), [V(L.T1), V(L.T1)], tag="t1 <<= 1"))
        for k in range(K - 1, -1, -1):
            c = s - k
            if c < 0 or c >= NCOL:
                continue
            m = L.MASK[c % L.NM]
            P, M, A, B = L.P[k], L.M[k], L.A[k], L.B[k]
            X, XV, SS, Q, MH, ECL = L.X[k], L.XV[k], L.S[k], L.Q[k], L.MH[k], L.ECL[k]
            t = f"k{k} c{c}"
            if k == 0:
                a(Ins("v_cmp_gt_i32", None, [0, V(L.HP)], sdst=S(L.SP[0]), tag=t + " cinP"))
                a(Ins("v_cmp_gt_i32", None, [0, V(L.HM)], sdst=S(L.SM[0]), tag=t + " cinM"))
                a(Ins("v_lshrrev_b32", V(L.CMV[0]), [31, V(L.HM)], tag=t + " cinM value"))
                a(Ins("v_add_u32", V(L.HP), [V(L.HP), V(L.HP)], tag="hp <<= 1"))
                a(Ins("v_add_u32", V(L.HM), [V(L.HM), V(L.HM)], tag="hm <<= 1"))
            a(Ins("v_xor_b32", V(X), [V(m), V(A)], tag=t + " x"))
            a(Ins("v_xor_b32", V(X + 1), [V(m), V(A + 1)], tag=t + " x"))
            a(Ins("v_bitop3_b32", V(X), [V(X), V(B), V(m + 1)], imm=TT_EQ, tag=t + " E"))
            a(Ins("v_bitop3_b32", V(X + 1), [V(X + 1), V(B + 1), V(m + 1)], imm=TT_EQ, tag=t + " E"))
            a(Ins("v_or_b32", V(XV), [V(X), V(M)], tag=t + " xv"))
            a(Ins("v_or_b32", V(XV + 1), [V(X + 1), V(M + 1)], tag=t + " xv"))
            a(Ins("v_or_b32", V(ECL), [V(X), V(L.CMV[k])], tag=t + " ecl"))
            a(Ins("v_and_b32", V(SS), [V(ECL), V(P)], tag=t + " t"))
            a(Ins("v_and_b32", V(SS + 1), [V(X + 1), V(P + 1)], tag=t + " t"))
            a(Ins("v_or_b32", V(Q), [V(ECL), V(P)], tag=t + " q"))
            a(Ins("v_or_b32", V(Q + 1), [V(X + 1), V(P + 1)], tag=t + " q"))
            a(Ins("v_add_co_u32", V(SS), [V(SS), V(P)], sdst=VCC, tag=t + " sum lo"))
            a(Ins("v_addc_co_u32", V(SS + 1), [V(SS + 1), V(P + 1)], sdst=VCC, ssrc=VCC, tag=t + " sum hi"))
            a(Ins("v_bitop3_b32", V(Q), [V(M), V(SS), V(Q)], imm=TT_PH, tag=t + " ph"))
            a(Ins("v_bitop3_b32", V(Q + 1), [V(M + 1), V(SS + 1), V(Q + 1)], imm=TT_PH, tag=t + " ph"))
            a(Ins("v_bitop3_b32", V(MH), [V(P), V(SS), V(ECL)], imm=TT_MH, tag=t + " mh"))
            a(Ins("v_bitop3_b32", V(MH + 1), [V(P + 1), V(SS + 1), V(X + 1)], imm=TT_MH, tag=t + " mh"))
            if k + 1 < K:
                a(Ins("v_lshrrev_b32", V(L.CMV[k + 1]), [31, V(MH + 1)], tag=t + " MHout value"))
            # (ph << 1) | PHin = ph + ph + PHin; the carry out of the high half is bit 63 of ph = PHout
            a(Ins("v_addc_co_u32", V(Q), [V(Q), V(Q)], sdst=VCC, ssrc=S(L.SP[k]), tag=t + " phs lo"))
            a(Ins("v_addc_co_u32", V(Q + 1), [V(Q + 1), V(Q + 1)], sdst=S(L.SP[k + 1]), ssrc=VCC, tag=t + " phs hi"))
            a(Ins("v_addc_co_u32", V(MH), [V(MH), V(MH)], sdst=VCC, ssrc=S(L.SM[k]), tag=t + " mhs lo"))
            a(Ins("v_addc_co_u32", V(MH + 1), [V(MH + 1), V(MH + 1)], sdst=S(L.SM[k + 1]), ssrc=VCC, tag=t + " mhs hi"))
            if k + 1 == K:
                a(Ins("v_addc_co_u32", V(L.GP), [V(L.GP), V(L.GP)], sdst=S(L.SDUMMY), ssrc=S(L.SP[K]), tag=t + " gP"))
                a(Ins("v_addc_co_u32", V(L.GM), [V(L.GM), V(L.GM)], sdst=S(L.SDUMMY), ssrc=S(L.SM[K]), tag=t + " gM"))
            a(Ins("v_bitop3_b32", V(P), [V(MH), V(XV), V(Q)], imm=TT_PH, tag=t + " P'"))
            a(Ins("v_bitop3_b32", V(P + 1), [V(MH + 1), V(XV + 1), V(Q + 1)], imm=TT_PH, tag=t + " P'"))
            a(Ins("v_and_b32", V(M), [V(Q), V(XV)], tag=t + " M'"))
            a(Ins("v_and_b32", V(M + 1), [V(Q + 1), V(XV + 1)], tag=t + " M'"))
    if vop3_all:
        for x in ins:
            if x.op in ("v_and_b32", "v_or_b32", "v_xor_b32"):
                x.imm = {"v_and_b32": TT_AND, "v_or_b32": TT_OR, "v_xor_b32": TT_XOR}[x.op]
                x.op = "v_bitop3_b32"
                x.src = [x.src[0], x.src[1], x.src[0]]
    return ins


def deps(ins):
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


def schedule(ins, dmin=4, window=300):
    """oldest ready instruction whose true producers are >= dmin positions back, else the one whose nearest producer is
    furthest back.  VCC is a register like any other here: a VCC writer cannot slip between a writer and its reader."""
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
    lowest = 0
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
                lo, hi = 0, len(ready)
                while lo < hi:
                    mid = (lo + hi) // 2
                    if ready[mid] < j:
                        lo = mid + 1
                    else:
                        hi = mid
                ready.insert(lo, j)
    return [ins[i] for i in order]


# -----------------------------------------------------------------------------------------------------------------------
# emulation of ONE lane (a lane mask is that lane's bit) against the plain column loop
# -----------------------------------------------------------------------------------------------------------------------
def emulate(seq, regs):
    def rd(s):
        return regs.get(s, 0xDEADBEEF) if isinstance(s, tuple) else s
    for x in seq:
        a = [rd(s) for s in x.src]
        cin = (regs.get(x.ssrc, 7) if x.ssrc is not None else 0)
        assert cin in (0, 1), (x.tag, "carry-in read before written")
        cout = None
        if x.op == "v_mov_b32":
            v = a[0]
        elif x.op == "v_xor_b32":
            v = a[0] ^ a[1]
        elif x.op == "v_or_b32":
            v = a[0] | a[1]
        elif x.op == "v_and_b32":
            v = a[0] & a[1]
        elif x.op == "v_lshrrev_b32":
            v = a[1] >> a[0]
        elif x.op == "v_ashrrev_i32":
            assert a[0] == 31
            v = MASK32 if a[1] >> 31 else 0
        elif x.op == "v_add_u32":
            v = a[0] + a[1]
        elif x.op == "v_bfrev_b32":
            v = int(format(a[0], "032b")[::-1], 2)
        elif x.op == "v_bitop3_b32":
            v = 0
            for bit in range(32):
                idx = (((a[0] >> bit) & 1) << 2) | (((a[1] >> bit) & 1) << 1) | ((a[2] >> bit) & 1)
                v |= ((x.imm >> idx) & 1) << bit
        elif x.op in ("v_add_co_u32", "v_addc_co_u32"):
            v = a[0] + a[1] + cin
            cout = v >> 32
        elif x.op == "v_cmp_gt_i32":
            v = None
            cout = 1 if (a[1] >> 31) & 1 else 0          # 0 > x  <=>  x negative
            assert a[0] == 0
        else:
            raise ValueError(x.op)
        if x.dst is not None:
            regs[x.dst] = v & MASK32
        if x.sdst is not None:
            regs[x.sdst] = cout


def reference(K, NCOL, P, M, A, B, T0, T1, HP, HM):
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


def check(K, NCOL, seq, L, trials=40, seed=1, valid_states_only=False):
    """NCOL == 32 only: the pass reads its text / carry words from bit 31 down after the bit reversal"""
    assert NCOL == 32
    rng = random.Random(seed)
    for _ in range(trials):
        P, M, A, B = [], [], [], []
        for k in range(K):
            x, y = rng.getrandbits(64), rng.getrandbits(64)
            if valid_states_only:
                P.append(x & ~y); M.append(y & ~x)
            else:                                   # any words: the cooperative kernels run slots on invalid encodings too (Pv = 0, Mv = ~0)
                P.append(x); M.append(y)
            A.append(rng.getrandbits(64)); B.append(rng.getrandbits(64))
        T0, T1 = rng.getrandbits(NCOL), rng.getrandbits(NCOL)
        h1, h2 = rng.getrandbits(NCOL), rng.getrandbits(NCOL)
        HP, HM = h1 & ~h2, h2 & ~h1
        regs = {}
        for k in range(K):
            for name, val in ((L.P[k], P[k]), (L.M[k], M[k]), (L.A[k], A[k]), (L.B[k], B[k])):
                regs[V(name)] = val & MASK32; regs[V(name + 1)] = val >> 32
        regs[V(L.T0)], regs[V(L.T1)], regs[V(L.HP)], regs[V(L.HM)] = T0, T1, HP, HM
        emulate(seq, regs)
        rP, rM, oP, oM = reference(K, NCOL, P, M, A, B, T0, T1, HP, HM)
        rev = lambda w: int(format(w, f"0{NCOL}b")[::-1], 2)       # noqa: E731
        for k in range(K):
            assert regs[V(L.P[k])] | (regs[V(L.P[k] + 1)] << 32) == rP[k], ("P", k)
            assert regs[V(L.M[k])] | (regs[V(L.M[k] + 1)] << 32) == rM[k], ("M", k)
        assert rev(regs[V(L.GP)]) == oP and rev(regs[V(L.GM)]) == oM, "carry words"
    return True


# -----------------------------------------------------------------------------------------------------------------------
def fo(s):
    if s == VCC:
        return "vcc"
    if isinstance(s, tuple):
        return f"v{s[1]}" if s[0] == "v" else f"s[{s[1]}:{s[1] + 1}]"
    return str(s)


def fmt(x):
    if x.op == "v_bitop3_b32":
        return f"v_bitop3_b32 {fo(x.dst)}, {fo(x.src[0])}, {fo(x.src[1])}, {fo(x.src[2])} bitop3:0x{x.imm:02x}"
    if x.op == "v_add_co_u32":
        return f"v_add_co_u32_e32 {fo(x.dst)}, vcc, {fo(x.src[0])}, {fo(x.src[1])}"
    if x.op == "v_addc_co_u32":
        if x.sdst == VCC and x.ssrc == VCC:
            return f"v_addc_co_u32_e32 {fo(x.dst)}, vcc, {fo(x.src[0])}, {fo(x.src[1])}, vcc"
        return f"v_addc_co_u32_e64 {fo(x.dst)}, {fo(x.sdst)}, {fo(x.src[0])}, {fo(x.src[1])}, {fo(x.ssrc)}"
    if x.op == "v_cmp_gt_i32":
        return f"v_cmp_gt_i32_e64 {fo(x.sdst)}, {fo(x.src[0])}, {fo(x.src[1])}"
    return f"{x.op} {fo(x.dst)}, " + ", ".join(fo(s) for s in x.src)


def emit(path, variants, vbase, sbase, dmin):
    out = ["// GENERATED by tools/gen_skew_asm2.py -- do not edit.  The K-slot skewed block-step pass of run64_skew<K> (qe_kernels.hip)",
           "// without quarter-rate instructions, on fixed VGPRs / SGPR pairs: the 64-bit sum is v_add_co + v_addc through VCC, every",
           "// '(x << 1) | carry' is x + x + carry with the carries of the slot boundaries as lane masks in SGPR pairs (the carry out of",
           "// the high half IS the block's PHout / MHout), column masks and the top slot's carry-ins come from running bit-reversed",
           "// words.  python3 tools/gen_skew_asm2.py --check emulates every pass against the plain column loop.",
           ""]
    for K, NCOL in variants:
        L = Layout(K, vbase, sbase)
        seq = schedule(build(K, NCOL, L), dmin)
        check(K, NCOL, seq, L, trials=12)
        bc = K * NCOL
        name = f"QE_SKEW2_K{K}"
        out.append(f"// K = {K}, {NCOL} columns: {len(seq)} instructions ({len(seq) / bc:.2f} per block-column); VGPRs v{vbase + (vbase & 1)} .. v{L.vend - 1}, "
                   f"SGPRs s{L.sbase} .. s{L.send - 1}")
        for nm, regs in (("P", L.P), ("M", L.M), ("A", L.A), ("B", L.B)):
            for k, r in enumerate(regs):
                out.append(f"#define {name}_{nm}{k} \"{{v[{r}:{r + 1}]}}\"")
        for nm, r in (("T0", L.T0), ("T1", L.T1), ("HP", L.HP), ("HM", L.HM), ("GP", L.GP), ("GM", L.GM)):
            out.append(f"#define {name}_{nm} \"{{v{r}}}\"")
        clob = ", ".join([f"\"v{r}\"" for r in range(L.first_tmp, L.vend)] + [f"\"s{r}\"" for r in range(L.sbase, L.send)] + ["\"vcc\""])
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
    ap.add_argument("--vbase", type=int, default=96)
    ap.add_argument("--sbase", type=int, default=40)
    ap.add_argument("--dmin", type=int, default=4)
    ap.add_argument("--variants", default="4x32,2x32")
    args = ap.parse_args()
    variants = [tuple(int(v) for v in s.split("x")) for s in args.variants.split(",")]
    if args.check or not args.emit:
        for K, NCOL in variants:
            L = Layout(K, args.vbase, args.sbase)
            prog = build(K, NCOL, L)
            for label, seq in (("program order", prog), (f"scheduled (dmin {args.dmin})", schedule(prog, args.dmin))):
                ok = check(K, NCOL, seq, L)
                ops = {}
                for x in seq:
                    ops[x.op] = ops.get(x.op, 0) + 1
                print(f"K={K} C={NCOL} {label:22s}: {len(seq)} instr = {len(seq) / (K * NCOL):.2f}/bc, VGPRs {L.vend - args.vbase}, SGPRs {L.send - L.sbase}, "
                      f"emulation {'ok' if ok else 'BAD'}; {dict(sorted(ops.items(), key=lambda kv: -kv[1]))}")
    if args.emit:
        emit(args.emit, variants, args.vbase, args.sbase, args.dmin)
        print("wrote", args.emit)


if __name__ == "__main__":
    sys.exit(main())
