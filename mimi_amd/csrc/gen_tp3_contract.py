#!/usr/bin/env python3
"""Generator of the hand-scheduled column loop of tp3_contract_asm_kernel (csrc/tensor_p3.hip) -> tp3_contract_loop.inc.

What the loop computes is documented at tp3_contract_kernel (the C++ form of the same arithmetic: S1 / S2 / S3, the carry
along the column); this file only decides WHERE every instruction goes and which register holds what.  The facts it is built
on (scratch/issue_bench.hip): on gfx950, with the one 512-register wave per SIMD this kernel runs at, no vector
instruction overlaps a `v_mfma_f64_16x16x4` (64 cycles) -- but LDS, global-memory and scalar instructions placed directly
behind a matrix instruction issue in its shadow, and cost ~4 issue cycles each anywhere else.  The compiler's schedule of
the C++ form leaves ~540 such instructions, 54 register-file moves and ~140 address computations per block in the vector
stretches and ends every element with `s_waitcnt vmcnt(0)` right behind its last stores.  Here:

  * every LDS / memory / scalar instruction of the loop sits behind a matrix instruction (a slot scheduler deals them out: at
    most one global-memory instruction, two LDS instructions and one group of scalar instructions per shadow);
  * the four accumulator tiles are double-buffered (tile set = b1 & 1): the stores and the carry writes of pair column b1
    go out in the shadows of pair column b1 + 1, the carry reads of b1 + 1 in the second half of b1's shadows;
  * the pieces leave with 16-byte stores: the pair columns 2 h, 2 h + 1 are adjacent in a piece row, register 0 of the even
    pair column's tiles is moved beside the odd one's (8 v_accvgpr_mov per pair), the finals are read from LDS into adjacent
    registers -- 16 stores per block instead of 32 (a store that meets a full queue holds the wave, which has no second
    wave behind it: profiles/r05_cfg3_contract_ablations_v*.txt, r05_cfg3_contract_x4_stores.txt);
  * addresses are scalar bases + one per-lane offset register + immediates: no vector address arithmetic in the loop;
  * ONE counted `s_waitcnt vmcnt(N)` per element at the loop top, N = the stores issued behind the last operand load
    (gfx9 has one in-order counter for global loads and stores -- no vscnt --, which is what the compiler's own counted
    waits rely on);
  * the first element of a column has no predecessor to store for: its stand-in stores (zeros) go to its own piece and are
    overwritten, in order, by the real ones -- no branch in the loop;
  * per value, the floating-point operations and their order are those of tp3_contract_kernel: the sums are bitwise equal
    (tests/test_tensor_p3_gpu.py compares the two kernels).

--drop=store,load,lds,valu,swap,mfma leaves classes of loop instructions out: timing experiments (scratch/p3_asm_variants.sh).

Register map (fixed; the asm statement clobbers exactly these, the compiler keeps v0..v[V0-1], the low SGPRs):
  VGPR  D (S1 results) 144 | E (S2 accumulators; bS2 operands during S1) 32 | W, cb, ca 24 | TA..TD 8 | t2 16 | 19 ints
  AGPR  aop 54 | two tile sets 64 + 8 staging (register 0 of the even tiles beside the odd ones': 16-byte stores) | bS0, bS0y 12 | final entries 32
The parameter block arrives in LDS (the carry area, which the loop zeroes afterwards): slot k = 512 bytes, lane-indexed.

usage: python gen_tp3_contract.py [out.inc]     (tests/test_isa_lint_cpu.py checks that the committed file is current)
"""
import os
import sys

NB, NQ = 4, 5

# ---------------------------------------------------------------- register map
V0 = 11                      # first VGPR of the asm block
VD = 12                      # D: DU[mn] = VD + 8 mn, DV[mn] = VD + 72 + 8 mn (8-aligned + 4: fine, tuples need even starts)
VE = VD + 144                # E[g][a1] = VE + 2 (4 g + a1); during S1: bS2[4], bS2U[4], bS2V[4]
VW = VE + 32                 # W3, W1, W2a, W2b, W0a, W0b
VCB = VW + 12                # cb[2] (double-buffered over the slots)
VCA = VCB + 4                # ca[4]
VT = VCA + 8                 # TA, TB, TC, TD
VT2 = VT + 8                 # t2[s][k]: VT2 + 2 (4 s + k)
VI = VT2 + 16                # 32-bit per-lane values
(I_OFFU, I_OFFX, I_OFFV, I_OFFU8, I_OFFX8, I_OFFV8, I_OT2, I_CIN0, I_CIN1, I_CIN2, I_CIN3, I_FIN, I_OUT0, I_OUT1, I_MU, I_MV,
 I_TMP0, I_TMP1, I_CL) = range(VI, VI + 19)
assert I_TMP0 % 2 == 0           # (a 64-bit operand needs an even register)
I_ZERO0, I_ZERO1 = I_TMP0, I_TMP1      # (prologue only, after the scalar parameters are read)
VEND = VI + 19
assert VEND <= 256, VEND

AA = 0                       # aop[mn][t] = AA + 18 t + 2 mn (a 16-byte load fills mn = 2 q, 2 q + 1 of one t)
AT = 56                      # tile set 0 (even pair columns), tile a1: AT + 8 a1
AT1 = AT + 32                # tile set 1 (odd pair columns): per a1 [2 registers: register 0 of the EVEN tile, moved here][the tile, 8]
AB0 = 128                    # bS0[g] = AB0 + 2 g, bS0y = AB0 + 8
AF = 140                     # final entries: pair h = b1 / 2, tile a1: AF + 16 (h & 1) + 4 a1 + 2 (b1 & 1)
AEND = 172
assert AT1 + 40 <= AB0 and AB0 + 12 <= AF

S0 = 36                      # first SGPR of the asm block
(S_REC0, S_REC1, S_REC2, S_RSTRIDE, S_CURA, S_CURB, S_CUR1, S_PRVA, S_PRVB, S_PRV1, S_PSTRIDE, S_B2, S_SPARE, S_TMP, S_EFF) = \
    [S0 + 2 * k for k in range(15)]
S_ES, S_NSEQ, S_TSTRIDE = S0 + 30, S0 + 31, S0 + 32
SEND = S0 + 34

# parameter slots in LDS (slot k: bytes [512 k, 512 k + 512), 8 bytes per lane)
P_TA, P_BS0, P_BS0X = 0, 4, 8
P_INT = 10                   # 16 ints in the order of I_OFFU .. I_MV
P_REC, P_RSTRIDE, P_PIECE, P_PSTRIDE, P_B2, P_D2, P_NSEQ = 26, 27, 28, 29, 30, 31, 32     # (P_B2: the packed table values; P_D2 unused)
N_PARAM = 33

REC_FIELD = 1024             # bytes per record field (128 points)
PIECE_OUT1 = 3072 * 8        # byte offset of the rows a2 >= 1 inside a piece


def v2(r):
    return f"v[{r}:{r + 1}]"


def a2(r):
    return f"a[{r}:{r + 1}]"


def s2r(r):
    return f"s[{r}:{r + 1}]"


def DU(mn, r=None):
    base = VD + 8 * mn
    return base if r is None else base + 2 * r


def DV(mn, r=None):
    base = VD + 72 + 8 * mn
    return base if r is None else base + 2 * r


def E(g, a1):
    return VE + 2 * (4 * g + a1)


W3, W1, W2a, W2b, W0a, W0b = [VW + 2 * k for k in range(6)]
T_REG = [VT, VT + 2, VT + 4, VT + 6]
BS2 = [VE + 2 * v for v in range(4)]
BS2U = [VE + 8 + 2 * v for v in range(4)]
BS2V = [VE + 16 + 2 * v for v in range(4)]


def T2(s, k):
    return VT2 + 2 * (4 * s + k)


def AOP(mn, t):
    return AA + 18 * t + 2 * mn


def TILE(s, a1, r=None):
    base = AT + 8 * a1 if s == 0 else AT1 + 10 * a1 + 2
    return base if r is None else base + 2 * r


def STAGE(a1):
    """four registers: [register 0 of even tile a1 (moved)][register 0 of odd tile a1 (in place)] = what one 16-byte store takes"""
    return AT1 + 10 * a1


def FIN(b1, a1):
    return AF + 16 * ((b1 >> 1) & 1) + 4 * a1 + 2 * (b1 & 1)


class Out:
    def __init__(self, drop=()):
        self.lines = []
        self.counts = {}
        self.drop = set(drop)      # timing experiments only (scratch/p3_asm_variants.sh): classes of loop instructions left out
        self.in_loop = False

    def emit(self, text, kind=None):
        op = text.split()[0]
        if op.startswith(".Ltp3_loop"):
            self.in_loop = True
        if self.in_loop and self.drop:
            cls = ("store" if op.startswith("global_store") else "load" if op.startswith("global_load") else
                   "lds" if op.startswith("ds_") else "mfma" if op.startswith("v_mfma") else
                   "swap" if op.startswith("v_permlane") else "valu" if op.startswith("v_") else "other")
            if cls in self.drop:
                return
        self.lines.append(text)
        k = kind or ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else
                     "lds" if op.startswith("ds_") else "vmem" if op.startswith("global_") else "salu")
        self.counts[k] = self.counts.get(k, 0) + 1

    def comment(self, text):
        self.lines.append("; " + text)


def mfma(d, a, b, c, d_agpr=False):
    dd = f"a[{d}:{d + 7}]" if d_agpr else f"v[{d}:{d + 7}]"
    cc = "0" if c is None else (f"a[{c}:{c + 7}]" if d_agpr else f"v[{c}:{c + 7}]")
    return f"v_mfma_f64_16x16x4_f64 {dd}, {a}, {b}, {cc}"


def fmac_bc(acc, table, w, n):
    return f"v_fmac_f64_dpp {v2(acc)}, {v2(table)}, {v2(w)} row_newbcast:{n} row_mask:0xf bank_mask:0xf"


def mov_bc(dst, table, n):
    return f"v_mov_b64_dpp {v2(dst)}, {v2(table)} row_newbcast:{n} row_mask:0xf bank_mask:0xf"


def mul(dst, x, y):
    return f"v_mul_f64 {v2(dst)}, {v2(x)}, {v2(y)}"


# ---------------------------------------------------------------- pieces of the schedule
def s2_pass(o, b1, plane, after_first_slot=None):
    """S2 of pair column b1: the points q0 < 4 (five slots) or the plane q0 = 4 (three slots).  Per value the operations
    of tp3_contract_kernel's s2 lambda; ordered so that no instruction reads a register written by one of the two
    instructions before it (fp64 results take 8.5 cycles, an instruction issues every 4.9)."""
    n_slot = 3 if plane else NQ
    cb = [VCB, VCB + 2]
    ca = [VCA + 2 * k for k in range(4)]

    def slot_tab(s):
        slot = NQ + s if plane else s
        T = T_REG[0] if slot < 2 else T_REG[1] if slot < 4 else T_REG[2] if slot < 6 else T_REG[3]
        cB = (slot & 1) * 8
        return T, cB, cB + 4

    def X(s, mn):
        if plane:
            return DV(mn, s + 1)
        return DU(mn, s) if s < 4 else DV(mn, 0)

    T, cB, cD = slot_tab(0)
    o.emit(mov_bc(cb[0], T, cB + b1))
    for a1 in range(NB):
        if a1 != b1:
            o.emit(mov_bc(ca[a1], T, cB + a1))
    for s in range(n_slot):
        T, cB, cD = slot_tab(s)
        c = cb[s & 1]
        o.emit(mul(W3, c, X(s, 0)))
        o.emit(mul(W1, c, X(s, 2)))
        o.emit(mul(W0a, c, X(s, 5)))
        o.emit(mul(W0b, c, X(s, 8)))
        o.emit(mul(W2a, c, X(s, 3)))
        o.emit(mul(W2b, c, X(s, 6)))
        o.emit(fmac_bc(W1, T, X(s, 1), cD + b1))
        o.emit(fmac_bc(W0a, T, X(s, 4), cD + b1))
        o.emit(fmac_bc(W0b, T, X(s, 7), cD + b1))
        if s + 1 < n_slot:
            Tn, cBn, _ = slot_tab(s + 1)
            o.emit(mov_bc(cb[(s + 1) & 1], Tn, cBn + b1))
        if s == 0:
            cax = lambda a1: c if a1 == b1 else ca[a1]
            for a1 in range(NB):
                o.emit(mul(E(3, a1), cax(a1), W3))
            for a1 in range(NB):
                o.emit(mul(E(2, a1), cax(a1), W2b))
            for a1 in range(NB):
                o.emit(mul(E(1, a1), cax(a1), W1))
            for a1 in range(NB):
                o.emit(mul(E(0, a1), cax(a1), W0b))
        else:
            for a1 in range(NB):
                o.emit(fmac_bc(E(3, a1), T, W3, cB + a1))
            for a1 in range(NB):
                o.emit(fmac_bc(E(2, a1), T, W2b, cB + a1))
            for a1 in range(NB):
                o.emit(fmac_bc(E(1, a1), T, W1, cB + a1))
            for a1 in range(NB):
                o.emit(fmac_bc(E(0, a1), T, W0b, cB + a1))
        for a1 in range(NB):
            o.emit(fmac_bc(E(2, a1), T, W2a, cD + a1))
        for a1 in range(NB):
            o.emit(fmac_bc(E(0, a1), T, W0a, cD + a1))
        if s == 0 and after_first_slot is not None:
            after_first_slot(o)          # (34 vector instructions behind the matrix instructions whose results it reads)


def carry_in(b1, tset):
    out = []
    for a1 in range(NB):
        for s in range(4):
            out.append(f"ds_read_b64 {a2(TILE(tset, a1, s))}, v{I_CIN0 + s} offset:{(a1 * 16 + b1 * 4 + s) * 512}")
    return out


def carry_out(b1, tset):
    out = []
    for r in range(1, 4):
        for a1 in range(NB):
            out.append(f"ds_write_b64 v{I_CL}, {a2(TILE(tset, a1, r))} offset:{(a1 * 16 + b1 * 4 + r - 1) * 512}")
    return out


def finals_in(b1):
    return [f"ds_read_b64 {a2(FIN(b1, a1))}, v{I_FIN} offset:{(a1 * 16 + b1 * 4) * 512}" for a1 in range(NB)]


def finals_store(h):
    # out1_prev[a1 4 48 + h 8 (+ b1 % 2)]: the finals of the pair columns 2 h, 2 h + 1 with one 16-byte store per tile; the base
    # register is piece + 2304 bytes so that the immediates fit 13 signed bits
    return [f"global_store_dwordx4 v{I_OUT1}, a[{FIN(2 * h, a1)}:{FIN(2 * h, a1) + 3}], {s2r(S_PRV1)} offset:{a1 * 1536 + h * 64 - 2304}"
            for a1 in range(NB)]


def stage_even(o):
    """register 0 of the even pair column's four tiles -> beside register 0 of the odd one's (vector instructions: in a vector stretch)"""
    for a1 in range(NB):
        o.emit(f"v_accvgpr_mov_b32 a{STAGE(a1)}, a{TILE(0, a1, 0)}")
        o.emit(f"v_accvgpr_mov_b32 a{STAGE(a1) + 1}, a{TILE(0, a1, 0) + 1}")


def out_store(h, prev):
    # out0[a1 4 192 + h 32 (+ b1 % 2)] = register 0 of tile a1 of the pair columns 2 h, 2 h + 1; bases A (a1 = 0, 1) and B (a1 = 2, 3)
    # sit 3072 bytes inside their range
    out = []
    for a1 in range(NB):
        base = (S_PRVA if prev else S_CURA) if a1 < 2 else (S_PRVB if prev else S_CURB)
        out.append(f"global_store_dwordx4 v{I_OUT0}, a[{STAGE(a1)}:{STAGE(a1) + 3}], {s2r(base)} offset:{(a1 & 1) * 6144 - 3072 + h * 256}")
    return out


def operand_loads():
    """the 27 record values of the next element (bases already point at it) as (first S1 slot it may follow, instruction):
    per operand kind t (0: tile U first k-step, 1: second k-step of both tiles, 2: tile V first k-step) four 16-byte loads
    of the pair fields q (mn = 2 q, 2 q + 1) and one 8-byte load of the single field mn = 8.  A load overwrites operand
    registers of matrix instructions and may only follow the last of them: S1 order DU first k-step (reads aop[mn][0]) at
    slot mn, DU second (aop[mn][1]) 9 + mn, DV first (aop[mn][2]) 18 + mn, DV second (aop[mn][1]) 27 + mn."""
    out = []
    first = {0: 0, 2: 18, 1: 27}
    for t, off16, off8 in ((0, I_OFFU, I_OFFU8), (2, I_OFFV, I_OFFV8), (1, I_OFFX, I_OFFX8)):
        for q in range(4):
            base = S_REC0 if q < 2 else S_REC1
            out.append((first[t] + 2 * q + 1,
                        f"global_load_dwordx4 a[{AOP(2 * q, t)}:{AOP(2 * q, t) + 3}], v{off16}, {s2r(base)} offset:{(q & 1) * 2 * REC_FIELD}"))
        out.append((first[t] + 8, f"global_load_dwordx2 {a2(AOP(8, t))}, v{off8}, {s2r(S_REC2)}"))
    return out


def table_loads():
    return [f"global_load_dwordx4 v[{T2(0, 0) + 4 * q}:{T2(0, 0) + 4 * q + 3}], v{I_OT2}, {s2r(S_B2)} offset:{16 * q}" for q in range(4)]


def advance_operand_bases():
    """rec / table bases -> the element after the one just requested, clamped at the last one (it is requested once
    more by the last iteration: no branch, nothing out of range).  es still counts the element being contracted."""
    return [
        [f"s_add_u32 s{S_TMP}, s{S_ES}, 2", f"s_cmp_lt_u32 s{S_TMP}, s{S_NSEQ}",
         f"s_cselect_b32 s{S_EFF}, s{S_RSTRIDE}, 0", f"s_cselect_b32 s{S_EFF + 1}, s{S_RSTRIDE + 1}, 0",
         f"s_cselect_b32 s{S_TMP}, s{S_TSTRIDE}, 0"],
        [f"s_add_u32 s{S_REC0}, s{S_REC0}, s{S_EFF}", f"s_addc_u32 s{S_REC0 + 1}, s{S_REC0 + 1}, s{S_EFF + 1}"],
        [f"s_add_u32 s{S_REC1}, s{S_REC1}, s{S_EFF}", f"s_addc_u32 s{S_REC1 + 1}, s{S_REC1 + 1}, s{S_EFF + 1}"],
        [f"s_add_u32 s{S_REC2}, s{S_REC2}, s{S_EFF}", f"s_addc_u32 s{S_REC2 + 1}, s{S_REC2 + 1}, s{S_EFF + 1}"],
        [f"s_add_u32 s{S_B2}, s{S_B2}, s{S_TMP}", f"s_addc_u32 s{S_B2 + 1}, s{S_B2 + 1}, 0"],
    ]


def rotate_piece_bases():
    out = []
    for prv, cur in ((S_PRVA, S_CURA), (S_PRVB, S_CURB), (S_PRV1, S_CUR1)):
        out.append([f"s_mov_b64 {s2r(prv)}, {s2r(cur)}",
                    f"s_add_u32 s{cur}, s{cur}, s{S_PSTRIDE}", f"s_addc_u32 s{cur + 1}, s{cur + 1}, s{S_PSTRIDE + 1}"])
    return out


def s1_mfmas():
    out = []
    v2_of = lambda mn: (1 if mn // 3 == 2 else 0) + (2 if mn % 3 == 2 else 0)
    for mn in range(9):
        out.append(mfma(DU(mn), a2(AOP(mn, 0)), v2(BS2[v2_of(mn)]), None))
    for mn in range(9):
        out.append(mfma(DU(mn), a2(AOP(mn, 1)), v2(BS2U[v2_of(mn)]), DU(mn)))
    for mn in range(9):
        out.append(mfma(DV(mn), a2(AOP(mn, 2)), v2(BS2[v2_of(mn)]), None))
    for mn in range(9):
        out.append(mfma(DV(mn), a2(AOP(mn, 1)), v2(BS2V[v2_of(mn)]), DV(mn)))
    return out


def s3_main(tset):
    out = []
    for a1 in range(NB):
        for g in range(4):
            out.append(mfma(TILE(tset, a1), v2(E(g, a1)), a2(AB0 + 2 * g), TILE(tset, a1), d_agpr=True))
    return out


def s3_plane(tset):
    # one instruction per tile: k = lane group = table variant g of direction 0, the A operand holds the plane's sum over q1
    # of variant g in lane group g (plane_totals); two instructions per tile with the partial sums of two variants each
    # until round 5
    return [mfma(TILE(tset, a1), v2(E(0, a1)), a2(AB0 + 8), TILE(tset, a1), d_agpr=True) for a1 in range(NB)]


def plane_totals(o):
    """The plane q0 = 4 leaves S2 as two partial sums per variant g: lane group 0 holds q1 = 0..2, lane group 1 q1 = 3, 4
    (lane groups 2, 3 hold zeros).  v_permlane16_swap (rows of 16 lanes: rows 1, 3 of the first operand <-> rows 0, 2 of the
    second) turns the pair (Ex[g], Ex[g + 1]) into [p0_g, p0_g+1, 0, 0], [p1_g, p1_g+1, 0, 0]; their sum is [t_g, t_g+1, 0, 0];
    v_permlane32_swap of the two sums gives [t_0, t_1, t_2, t_3]: the A operand of ONE matrix instruction whose k index
    is the variant -- 4 instead of 8 plane instructions per pair column for 16 more vector instructions (64 cycles each
    against 5).  In place: the result is in E[0][a1]."""
    for g in (2, 0):                    # (E[0][*] were written by the last instructions of S2: the pairs (2, 3) first)
        for a1 in range(NB):
            o.emit(f"v_permlane16_swap_b32 v{E(g, a1)}, v{E(g + 1, a1)}")
            o.emit(f"v_permlane16_swap_b32 v{E(g, a1) + 1}, v{E(g + 1, a1) + 1}")
    for g in (2, 0):
        for a1 in range(NB):
            o.emit(f"v_add_f64 {v2(E(g, a1))}, {v2(E(g, a1))}, {v2(E(g + 1, a1))}")
    for a1 in range(NB):
        o.emit(f"v_permlane32_swap_b32 v{E(0, a1)}, v{E(2, a1)}")
        o.emit(f"v_permlane32_swap_b32 v{E(0, a1) + 1}, v{E(2, a1) + 1}")


def bs2_operands(o):
    """B operands of S1 from the raw direction-2 tables: bS2[v] (first k-step), and the second k-step's product masked to
    lane group 0 (tile U) / 1 (tile V) -- `kk == 0 ? x : 0.0` as a bitwise AND with an all-ones / zero lane mask."""
    for v in range(4):
        o.emit(mul(BS2[v], T2(0, 1 if v & 1 else 0), T2(0, 3 if v & 2 else 2)))
    for v in range(4):
        o.emit(mul(BS2U[v], T2(1, 1 if v & 1 else 0), T2(1, 3 if v & 2 else 2)))
    for v in range(4):
        o.emit(f"v_and_b32 v{BS2V[v]}, v{I_MV}, v{BS2U[v]}")
        o.emit(f"v_and_b32 v{BS2V[v] + 1}, v{I_MV}, v{BS2U[v] + 1}")
    for v in range(4):
        o.emit(f"v_and_b32 v{BS2U[v]}, v{I_MU}, v{BS2U[v]}")
        o.emit(f"v_and_b32 v{BS2U[v] + 1}, v{I_MU}, v{BS2U[v] + 1}")


def generate(opts=None):
    opts = opts or {}
    o = Out(opts.get("drop", ()))
    # ------------------------------------------------------------ prologue: parameters, zero carry, first requests
    o.comment("parameters (LDS slot k at 512 k + 8 lane; %0 = this lane's address of slot 0)")
    o.emit(f"v_mov_b32 v{I_CL}, %0")
    o.emit("s_waitcnt lgkmcnt(0)")
    for k in range(4):
        o.emit(f"ds_read_b64 {v2(T_REG[k])}, v{I_CL} offset:{(P_TA + k) * 512}")
    for k in range(6):
        o.emit(f"ds_read_b64 {a2(AB0 + 2 * k)}, v{I_CL} offset:{(P_BS0 + k) * 512}")
    for k in range(16):
        o.emit(f"ds_read_b32 v{VI + k}, v{I_CL} offset:{(P_INT + k) * 512}")
    o.emit("s_waitcnt lgkmcnt(0)")
    scal = [(P_REC, S_REC0, 2), (P_RSTRIDE, S_RSTRIDE, 2), (P_PIECE, S_CURA, 2), (P_PSTRIDE, S_PSTRIDE, 2), (P_B2, S_B2, 2),
            (P_NSEQ, S_NSEQ, 1)]
    for slot, sreg, n in scal:
        o.emit(f"ds_read_b64 {v2(I_TMP0)}, v{I_CL} offset:{slot * 512}")
        o.emit("s_waitcnt lgkmcnt(0)")
        o.emit(f"v_readfirstlane_b32 s{sreg}, v{I_TMP0}")
        if n == 2:
            o.emit(f"v_readfirstlane_b32 s{sreg + 1}, v{I_TMP1}")
    o.emit("s_nop 4")                      # (vector write of a scalar register -> its use as a memory base)
    o.comment("derived bases")
    o.emit(f"s_add_u32 s{S_REC1}, s{S_REC0}, {4 * REC_FIELD}")          # pair fields q = 2, 3
    o.emit(f"s_addc_u32 s{S_REC1 + 1}, s{S_REC0 + 1}, 0")
    o.emit(f"s_add_u32 s{S_REC2}, s{S_REC0}, {8 * REC_FIELD}")          # the single field mn = 8
    o.emit(f"s_addc_u32 s{S_REC2 + 1}, s{S_REC0 + 1}, 0")
    o.emit(f"s_add_u32 s{S_CUR1}, s{S_CURA}, {PIECE_OUT1 + 2304}")
    o.emit(f"s_addc_u32 s{S_CUR1 + 1}, s{S_CURA + 1}, 0")
    o.emit(f"s_add_u32 s{S_CURB}, s{S_CURA}, {3072 + 12288}")
    o.emit(f"s_addc_u32 s{S_CURB + 1}, s{S_CURA + 1}, 0")
    o.emit(f"s_add_u32 s{S_CURA}, s{S_CURA}, 3072")
    o.emit(f"s_addc_u32 s{S_CURA + 1}, s{S_CURA + 1}, 0")
    for prv, cur in ((S_PRVA, S_CURA), (S_PRVB, S_CURB), (S_PRV1, S_CUR1)):
        o.emit(f"s_mov_b64 {s2r(prv)}, {s2r(cur)}")     # (element 0 has no predecessor: its stand-in stores go to its own piece)
    o.emit(f"s_mov_b32 s{S_TSTRIDE}, {64 * 8 * 8}")                    # table values per span: [64 lanes][8]
    o.emit(f"s_mov_b32 s{S_ES}, 0")
    o.comment("element 0's operands, then the bases move on to element 1 (to element 0 again if the column has one element)")
    for x in [x for _, x in operand_loads()] + table_loads():
        o.emit(x)
    o.emit(f"s_cmp_lt_u32 1, s{S_NSEQ}")
    o.emit(f"s_cselect_b32 s{S_EFF}, s{S_RSTRIDE}, 0")
    o.emit(f"s_cselect_b32 s{S_EFF + 1}, s{S_RSTRIDE + 1}, 0")
    o.emit(f"s_cselect_b32 s{S_TMP}, s{S_TSTRIDE}, 0")
    for grp in advance_operand_bases()[1:]:
        for x in grp:
            o.emit(x)
    o.comment("zero carry (the parameters are read), zero tile set 1 and final set 1 (flushed, as stand-ins, during element 0)")
    o.emit(f"v_mov_b32 v{I_ZERO0}, 0")
    o.emit(f"v_mov_b32 v{I_ZERO1}, 0")
    for k in range(64):
        o.emit(f"ds_write_b64 v{I_CL}, {v2(I_ZERO0)} offset:{k * 512}")
    for r in range(40):
        o.emit(f"v_accvgpr_write_b32 a{AT1 + r}, 0")
    for r in range(32):
        o.emit(f"v_accvgpr_write_b32 a{AF + r}, 0")
    for x in carry_in(0, 0):
        o.emit(x)
    o.emit("s_waitcnt vmcnt(0)")
    o.emit(".Ltp3_loop_%=:")
    # ------------------------------------------------------------ one element
    # Matrix instruction slots of an element: S1 0..35, then per pair column b1 S3 main 36 + 20 b1 .. + 15 and S3 plane .. + 19.
    # Everything that is not a vector instruction is dealt out into the shadows behind them: at most `vm_cap` global-memory
    # instructions per shadow (the CU's address unit takes 16 - 21 cycles per wave instruction and is shared by four waves:
    # two stores in one shadow already outlast it -- measured, profiles/r05_cfg3_contract_ablations_v1.txt), `lds_cap` LDS
    # instructions, one group of scalar instructions.
    vm_cap, lds_cap = opts.get("vm_cap", 1), opts.get("lds_cap", 2)
    NM, NP = 16, 4                          # S3 main / S3 plane matrix instructions per pair column
    n_slot = 36 + 4 * (NM + NP)
    m_of = lambda b1: 36 + (NM + NP) * b1          # first S3 main slot of pair column b1
    p_of = lambda b1: 36 + (NM + NP) * b1 + NM     # first S3 plane slot
    mf = s1_mfmas()
    for b1 in range(4):
        mf += s3_main(b1 & 1) + s3_plane(b1 & 1)
    assert len(mf) == n_slot
    # (kind, earliest slot, latest slot, instruction or group).  Order inside a kind = issue order.
    ops = []

    def add(kind, lo, hi, items):
        for x in items:
            ops.append([kind, lo, hi, x])

    # outputs of the previous element's pair column 3 (tile set 1, `prev` bases) -- before tile set 1 is read into again
    for i, x in enumerate(out_store(1, prev=True)):
        add("vm", i, 35, [x])
    add("lds", 0, 35, carry_out(3, 1))
    # next element's operands (operand_loads); the tables' registers are free once the B operands are formed
    last_load = opts.get("last_load", m_of(2) - 1)
    load_spread, store_spread = opts.get("load_spread", 1), opts.get("store_spread", 2)
    loads = [(0, x) for x in table_loads()] + sorted(operand_loads(), key=lambda t: t[0])
    first_load = opts.get("first_load", 4)
    for i, (lo, x) in enumerate(loads):
        add("vm", max(lo, first_load + i * load_spread), last_load, [x])
    add("sc", last_load + 1, n_slot - 1, advance_operand_bases())
    for b1 in range(4):
        tset = b1 & 1
        m, pl = m_of(b1), p_of(b1)
        # first half of the S3 main shadows: the outputs of pair column b1 - 1 (the other tile set) and the finals the next pair
        # column will store; second half: this pair column's finals (found by the previous one, waited for at the head of S3
        # main) and -- once nothing reads the other tile set any more -- the start values of the next pair column's tiles
        if b1 == 2:
            for i, x in enumerate(out_store(0, prev=False)):      # pair columns 0, 1 (the odd one's tiles are read into again from m + 8)
                add("vm", m + store_spread * i, m + 7, [x])
        if b1 > 0:
            add("lds", m, m + 7, carry_out(b1 - 1, 1 - tset))
        add("lds", m, m + 7, finals_in((b1 + 1) % 4))
        if b1 & 1:
            for i, x in enumerate(finals_store(b1 >> 1)):         # the finals of pair columns b1 - 1, b1 (found by the shadows before)
                add("vm", m + 8 + store_spread * i, pl - 1, [x])
        add("lds", m + 8 if b1 > 0 else m, pl + NP - 1, carry_in((b1 + 1) % 4, 1 - tset))
    add("sc", p_of(3), n_slot - 1, rotate_piece_bases())
    # assignment: slot by slot, per kind the eligible instructions with the earliest deadline first
    cap = {"vm": vm_cap, "lds": lds_cap, "sc": 1}
    shadow = [[] for _ in range(n_slot)]
    for k in range(n_slot):
        for kind in ("vm", "lds", "sc"):
            ready = sorted((q for q in ops if q[3] is not None and q[0] == kind and q[1] <= k), key=lambda q: q[2])
            for q in ready[:cap[kind]]:
                assert k <= q[2], (k, q)
                shadow[k].append(q[3])
                q[3] = None
    late = [q for q in ops if q[3] is not None]
    assert not late, late[:3]
    flat = [y for k in range(n_slot) for x in shadow[k] for y in (x if isinstance(x, list) else [x])]
    last = max(k for k, y in enumerate(flat) if y.startswith("global_load"))
    n_vm_after_last_load = sum(1 for y in flat[last + 1:] if y.startswith("global_"))
    o.comment("B operands of S1")
    assert n_vm_after_last_load < 60
    o.emit(f"s_waitcnt vmcnt({n_vm_after_last_load})")     # this element's operands (vector memory instructions of a wave complete in order on gfx9)
    bs2_operands(o)
    for k, m in enumerate(mf):
        if k >= 36 and (k - 36) % (NM + NP) == 0:
            b1 = (k - 36) // (NM + NP)
            o.comment(f"pair column b1 = {b1}: S2 (points q0 < 4), S3 main")
            s2_pass(o, b1, plane=False, after_first_slot=stage_even if b1 & 1 else None)
            o.emit("s_waitcnt lgkmcnt(0)")
        if k >= 36 and (k - 36) % (NM + NP) == NM:
            b1 = (k - 36) // (NM + NP)
            o.comment("S2 (plane q0 = 4), S3 plane")
            s2_pass(o, b1, plane=True)
            plane_totals(o)
        o.emit(m)
        for x in shadow[k]:
            for y in (x if isinstance(x, list) else [x]):
                o.emit(y)
    o.emit(f"s_add_u32 s{S_ES}, s{S_ES}, 1")
    o.emit(f"s_cmp_lt_u32 s{S_ES}, s{S_NSEQ}")
    o.emit(f"s_cbranch_scc1 .Ltp3_loop_%=")
    # ------------------------------------------------------------ after the last element: its pair column 3
    o.comment("flush: the last element's pair column 3 (the piece bases were rotated: it is `prev`)")
    o.emit("s_nop 15")
    o.emit("s_nop 3")
    for x in out_store(1, prev=True) + carry_out(3, 1):
        o.emit(x)
    o.emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return o


def clobbers():
    regs = [f"v{k}" for k in range(V0, 256)] + [f"a{k}" for k in range(0, AEND)] + [f"s{k}" for k in range(S0, SEND)]
    return regs + ["vcc", "scc", "memory"]


def render(o):
    body = [f'    "{ln}\\n\\t"' for ln in o.lines if not ln.startswith(";")]
    cl = ", ".join(f'"{r}"' for r in clobbers())
    head = ("// GENERATED by gen_tp3_contract.py -- do not edit (tests/test_isa_lint_cpu.py checks that it is current).\n"
            "// The column loop of tp3_contract_asm_kernel as ONE asm statement; register map and schedule: see the generator.\n"
            f"// instructions (prologue + one element + flush): {o.counts}\n")
    return (head + "#define T3_ASM_LOOP(cl_addr) asm volatile( \\\n" + " \\\n".join(body) +
            " \\\n    :: \"v\"(cl_addr) \\\n    : " + cl + ")\n")


def constants_header():
    names = ["P_TA", "P_BS0", "P_BS0X", "P_INT", "P_REC", "P_RSTRIDE", "P_PIECE", "P_PSTRIDE", "P_B2", "P_D2", "P_NSEQ", "N_PARAM"]
    g = globals()
    return "".join(f"constexpr int T3A_{n} = {g[n]};\n" for n in names)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {}
    for a in sys.argv[1:]:
        if a.startswith("--drop="):
            opts["drop"] = a[7:].split(",")
        elif a.startswith("--") and "=" in a:
            opts[a[2:a.index("=")]] = int(a[a.index("=") + 1:])
    out = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "tp3_contract_loop.inc")
    o = generate(opts)
    text = render(o) + constants_header()
    with open(out, "w") as f:
        f.write(text)
    print(out, o.counts, "lines", len(o.lines))


if __name__ == "__main__" and "--check" not in sys.argv:
    main()


# ---------------------------------------------------------------- self-check of the schedule (no GPU, no compiler)
def check_schedule(o, n_iter=3):
    """Walks prologue + n_iter unrolled elements + flush in program order and checks what the disassembly lint does not:
    every register a memory instruction fills is waited for (s_waitcnt) before its first use or overwrite -- LDS
    instructions of a wave return in order, so lgkmcnt(N) covers all but the youngest N; global loads and stores share one
    in-order counter on gfx9 (no vscnt: what the compiler's own waits rely on), so vmcnt(N) covers all but the youngest N -- and a register pair a
    vector instruction wrote is not read by a v_permlane32_swap / a DPP operand within two wait states.
    Returns the list of findings (empty = fine)."""
    import re
    reg_re = re.compile(r"\b([vas])(?:(\d+)|\[(\d+):(\d+)\])(?![\w.])")

    def regs(text):
        out = set()
        for m in reg_re.finditer(text):
            lo, hi = (int(m.group(2)),) * 2 if m.group(2) is not None else (int(m.group(3)), int(m.group(4)))
            out.update((m.group(1), k) for k in range(lo, hi + 1))
        return out

    lines = [ln for ln in o.lines if not ln.startswith(";")]
    start = next(k for k, ln in enumerate(lines) if ln.startswith(".Ltp3_loop"))
    end = next(k for k, ln in enumerate(lines) if ln.startswith("s_cbranch_scc1"))
    seq = lines[:start] + lines[start + 1:end] * n_iter + lines[end + 1:]
    pending = {}              # register -> ("lds", number) | ("vm", number)
    n_lds = n_vm = 0
    done_lds = 0              # LDS instructions 1..done_lds are known to be complete
    last_valu_write = {}      # register -> instruction index
    bad = []
    for idx, ln in enumerate(seq):
        op = ln.split()[0]
        rest = ln[len(op):]
        operands = [x.strip() for x in rest.split(",")] if rest.strip() else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ln)
            if m:
                done_lds = max(done_lds, n_lds - int(m.group(1)))
            m = re.search(r"vmcnt\((\d+)\)", ln)
            if m:
                done_vm = n_vm - int(m.group(1))       # (global loads and stores of a wave complete in order on gfx9: one counter)
                for r in [r for r, (kind, num) in pending.items() if kind == "vm" and num <= done_vm]:
                    del pending[r]
            for r in [r for r, (kind, num) in pending.items() if kind == "lds" and num <= done_lds]:
                del pending[r]
            continue
        touched = regs(rest)
        for r in touched:
            if r in pending:
                bad.append((idx, ln, f"{r[0]}{r[1]} is still being filled by a {pending[r][0]} instruction"))
        if op.startswith("v_permlane32_swap") or "_dpp" in op:
            watch = regs(operands[1]) if "_dpp" in op else touched
            for r in watch:
                if idx - last_valu_write.get(r, -10) <= 2:
                    bad.append((idx, ln, f"{r[0]}{r[1]} written {idx - last_valu_write[r]} instruction(s) before"))
        if op.startswith("ds_read"):
            n_lds += 1
            for r in regs(operands[0]):
                pending[r] = ("lds", n_lds)
        elif op.startswith("ds_write"):
            n_lds += 1
        elif op.startswith("global_load"):
            n_vm += 1
            for r in regs(operands[0]):
                pending[r] = ("vm", n_vm)
        elif op.startswith("global_store"):
            n_vm += 1
        elif op.startswith("v_") and operands:
            for r in regs(operands[0]):
                last_valu_write[r] = idx
            if op.startswith("v_permlane32_swap"):
                for r in regs(operands[1]):
                    last_valu_write[r] = idx
    return bad


if __name__ == "__main__" and "--check" in sys.argv:
    findings = check_schedule(generate())
    for f in findings[:20]:
        print(f)
    print(len(findings), "findings")
