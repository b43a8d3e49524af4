"""ISA lint of the hand-scheduled fp64 matrix instructions (no GPU).

csrc/tensor_p3.hip issues its 132 `v_mfma_f64_16x16x4_f64` per contraction block from inline asm, where LLVM's hazard
recogniser pads nothing; mimi_amd/isa_lint.py checks the compiled code: every use of a matrix result is far enough behind
its producer (wait states as the compiler counts them, requirements read back from the compiler's own padding of the
builtin), no vector write / EXEC write sits too close in front of an asm matrix instruction, and the kernels spill no
register.  The degree-2 kernels use the builtin -- compiler-padded code, which must pass the same lint with EVERY pair
checked: that holds the lint's model against the compiler's (its nearest pairs sit exactly at 19 / 11 / 2 wait states).
The reference has no analogue; this protects SURVEY 8 rows a7 / a8 at n_dof 64 and 27."""
import os
import subprocess
import tempfile

import pytest

from mimi_amd import isa_lint as L

# (a CPU box without ROCm: nothing to lint -- skip instead of erroring; with it, a failed compile shows the compiler's stderr)
pytestmark = pytest.mark.skipif(not os.path.exists(L.HIPCC), reason="no hipcc: nothing to compile")


@pytest.fixture(scope="module")
def need():
    return L.calibrate()


def test_requirements_read_back_from_the_compiler(need):
    # gfx950: v_mfma_f64_16x16x4 runs 16 passes; result -> vector read 19, -> memory 18, -> A / B of a matrix instruction 19;
    # vector write -> matrix read 2.  (If a compiler bump changes its padding, this says so before anything else.)
    assert need["valu_read"] == 19 and need["mem_read"] == 18 and need["mfma_srcab"] == 19 and need["valu_def"] == 2


def test_degree3_contraction_kernel_is_hazard_free_and_spill_free(need):
    (bad, stats), = L.report("tensor_p3.hip", ["tp3_contract_kernel"], need, asm_only=True).values()
    assert stats["mfma"] == 116 and stats["mfma_from_asm"] == 116          # (all of them hand-placed)
    assert not bad, "\n".join(f"{w}: {d} < {r}\n  {a}\n  {b}" for a, b, d, r, w in bad[:10])
    assert stats["vgpr_spill_count"] == 0
    # the margins of the shipped schedule (informative: a change here is a change of the schedule, not yet a hazard)
    assert stats["nearest_valu_read"] >= need["valu_read"] and stats["nearest_mem"] >= need["mem_read"]
    assert stats["nearest_dpp_def"] is None or stats["nearest_dpp_def"] >= need["dpp_def"]
    print("tp3_contract_kernel", stats)


def test_the_lint_turns_red_on_the_shipped_kernel_without_its_results_guard(need):
    """tensor_p3.hip's t3_results_guard (one `s_nop 15` behind the S3 matrix instructions of the plane tiles) is what keeps
    the stores of those tiles 18 wait states behind their producers: compiled away (-DT3_LINT_DROP_RESULTS_GUARD, a switch
    that exists for this test) the first stores follow after 3, and the lint must say so.  (The `s_nop` pair rounds 2-3 had
    behind S1 protected nothing -- the lint showed 42 wait states to the first reader with or without it -- and is gone;
    the pair behind S3 was 32 wait states where 16 are needed.)"""
    asm = L.assembly("tensor_p3.hip", out=os.path.join(tempfile.gettempdir(), "tensor_p3.noguard.lint.s"),
                     extra_flags=["-DT3_LINT_DROP_RESULTS_GUARD"])
    bad, stats = L.lint_kernel(L.parse_kernel(asm, "tp3_contract_kernel"), need, asm_only=True)
    assert len(bad) > 20 and stats["nearest_mem"] < need["mem_read"]
    assert all("memory / LDS" in what for *_, what in bad)


def test_degree2_kernels_compiler_padded_code_passes_the_same_lint(need):
    rep = L.report("domain.hip", ["tensor_wgsym_kernel", "tensor_wgs_kernel"], need, asm_only=False)
    for kernel, (bad, stats) in rep.items():
        assert stats["mfma"] > 200 and not bad, (kernel, bad[:5])
        assert stats["vgpr_spill_count"] == 0
        # the compiler pads to the requirement and no further: the lint's distances are the compiler's
        assert stats["nearest_valu_read"] == need["valu_read"] and stats["nearest_valu_write"] == need["valu_write"]
        assert stats["nearest_valu_def"] == need["valu_def"]
        print(kernel, stats)


def test_residual_column_kernel_dpp_operands_are_linted(need):
    """VERDICT round 4 weak 3 / ADVICE round 4: the lint's DPP rule was only ever run on a synthetic kernel.  The kernel it was
    written for -- tensor_residual_col_kernel (csrc/kernels_tensor_residual.hpp), 188 inline-asm `row_newbcast` instructions
    whose table registers are written by vector instructions -- is linted here: every asm DPP instruction seen, no finding,
    no spilled register.  And the lint must be ABLE to turn red on this kernel's instruction stream: with this compiler the
    kernel's RC_DPP_FENCE statements are a second line (compiled away, the nearest write -> DPP read is still beyond 8
    wait states), so the negative case mutates the compiled code -- a vector write of the table register placed directly in
    front of each of ten DPP reads of it is flagged ten times, and with two wait states in between it is not."""
    asm = L.assembly("domain.hip")
    instrs = L.parse_kernel(asm, "tensor_residual_col_kernel")
    dpp = [k for k, x in enumerate(instrs) if x.from_asm and "_dpp" in x.op]
    assert len(dpp) >= 150
    bad, stats = L.lint_kernel(instrs, need, asm_only=True)
    assert not bad, bad[:5]
    assert stats["nearest_dpp_def"] is None or stats["nearest_dpp_def"] >= need["dpp_def"]
    full = next(n for n in L.spill_counts(asm) if "tensor_residual_col_kernel" in n)
    assert L.spill_counts(asm)[full] == 0

    def mutated(pad):
        out = list(instrs[:dpp[0]])
        for a, b in zip(dpp[:10], dpp[1:11]):
            x = instrs[a]
            w = L.Instr()
            w.op, w.text, w.line, w.from_asm = "v_max_f64", f"v_max_f64 {x.operands[1]}, {x.operands[1]}, {x.operands[1]}", x.line, True
            w.operands = [x.operands[1]] * 3
            w.defs = w.uses = L._regs(x.operands[1])
            w.nop, w.target = 1, None
            out.append(w)
            for _ in range(pad):
                n = L.Instr()
                n.op, n.text, n.line, n.from_asm, n.operands = "s_nop", "s_nop 0", x.line, True, ["0"]
                n.defs, n.uses, n.nop, n.target = set(), set(), 1, None
                out.append(n)
            out.extend(instrs[a:b])
        # (branch targets are instruction indices of the unmutated list: the mutated prefix is straight-line code up to
        # the first branch behind it, which is all the walk from the inserted writes needs)
        for x in out:
            if x.target is not None:
                x.target = None
        return out
    bad, stats = L.lint_kernel(mutated(0), need, asm_only=True)
    assert len([b for b in bad if "DPP operand" in b[4]]) >= 10 and stats["nearest_dpp_def"] == 0
    bad, stats = L.lint_kernel(mutated(2), need, asm_only=True)
    assert not [b for b in bad if "DPP operand" in b[4]]


def test_no_shipped_kernel_spills_more_than_a_handful_of_registers():
    """the colour-partitioned fallback kernel spilled 938 registers and the 3-D tangent
    instantiations of the general kernels for J2Simo / J2Log 480 - 520.  Round 5: the colour kernel is gone (such patches take
    the general kernels), and the other materials' tangent comes from a material pre-pass with the register file to itself
    (general_material_kernel, one direction at a time: 0 spills).  Every kernel of every translation unit of the library
    now spills fewer than 64 registers (the largest: 58, the closed-form 3-D tangent of the general path at two waves
    per SIMD)."""
    worst = {}
    n_kernels = 0
    for src in ("domain.hip", "tensor_p3.hip", "contact.hip", "krylov.hip", "exchange.hip"):
        for name, count in L.spill_counts(L.assembly(src)).items():
            n_kernels += 1
            if count:
                worst[name] = count
    assert n_kernels > 150
    assert max(worst.values(), default=0) < 64, sorted(worst.items(), key=lambda t: -t[1])[:5]


def test_horner_steps_from_inline_asm_never_read_a_transcendental_result_directly(need):
    """Round 5: pow_positive's Horner steps are `v_fma_f64` from inline asm (csrc/materials.hpp horner_step) -- instructions
    the compiler's hazard recogniser does not look into.  The one hazard that applies to an ordinary vector instruction on
    gfx940+ is the transcendental-unit forwarding one: a result of v_rcp / v_rsq / v_sqrt / v_exp / v_log read by a
    non-transcendental vector instruction needs one wait state.  By construction the operands of a Horner step come from
    multiplies, multiply-adds and moves; the lint checks it on the shipped J2 kernels of the degree-3 pre-pass (189 asm vector
    instructions each: nine inlined calls) and is part of the build's gate for them.  Red case: the same stream with a v_rcp_f64
    writing a Horner step's operand right in front of it is flagged, and with one instruction in between it is not."""
    asm = L.assembly("tensor_p3.hip")
    for tag in (0, 1, 2):
        instrs = L.parse_kernel(asm, "tp3_point_kernelILi0ELi%d" % tag)
        bad, stats = L.lint_kernel(instrs, need, asm_only=True)
        assert not bad, bad[:5]
        assert stats["valu_from_asm"] >= 150 and (stats["nearest_trans_use"] is None or stats["nearest_trans_use"] >= need["trans_use"])
    steps = [k for k, x in enumerate(instrs) if x.from_asm and x.op == "v_fma_f64"]

    def mutated(pad):
        out = []
        k = steps[0]
        x = instrs[k]
        out.extend(instrs[:k])
        w = L.Instr()
        src = x.operands[1]
        w.op, w.text, w.line, w.from_asm = "v_rcp_f64_e32", f"v_rcp_f64_e32 {src}, {src}", x.line, False
        w.operands = [src, src]
        w.defs = w.uses = L._regs(src)
        w.nop, w.target = 1, None
        out.append(w)
        for _ in range(pad):
            n = L.Instr()
            n.op, n.text, n.line, n.from_asm, n.operands = "s_nop", "s_nop 0", x.line, False, ["0"]
            n.defs, n.uses, n.nop, n.target = set(), set(), 1, None
            out.append(n)
        out.extend(instrs[k:steps[1]])
        for y in out:
            y.target = None
        return out
    bad, stats = L.lint_kernel(mutated(0), need, asm_only=True)
    assert len([b for b in bad if "transcendental" in b[4]]) == 1 and stats["nearest_trans_use"] == 0
    bad, stats = L.lint_kernel(mutated(1), need, asm_only=True)
    assert not [b for b in bad if "transcendental" in b[4]]


def test_every_kernel_with_asm_horner_steps_passes_the_lint(need):
    """The same rule over every kernel of the other translation units that inlines pow_positive (the degree-2 J2 kernels, the
    general path, the other materials: ~ 77 kernels, 16 000 asm vector instructions) -- about a minute."""
    import re
    n_kernels = n_asm = 0
    for src in ("domain.hip", "contact.hip"):
        asm = L.assembly(src)
        for name in re.findall(r"^(_Z\w+):", asm, re.M):
            instrs = L.parse_kernel(asm, name)
            if not any(x.from_asm and x.op == "v_fma_f64" for x in instrs):
                continue
            bad, stats = L.lint_kernel(instrs, need, asm_only=True)
            assert not [b for b in bad if "transcendental" in b[4]], (name, bad[:3])
            n_kernels += 1
            n_asm += stats["valu_from_asm"]
    assert n_kernels >= 40 and n_asm >= 5000, (n_kernels, n_asm)


def test_the_degree3_prepass_kernels_keep_their_waves_per_simd():
    """Round 5: the J2 modes of the degree-3 pre-pass are bound by the latency of the return-map iteration at the occupancy
    their registers and LDS allow: the residual-only mode and the state commit run at four waves per SIMD (123 /
    119 of <= 128 registers: F^-1 and the deviator parked, the coefficients of pow_positive in scalar registers, no inlined
    library pow), the residual+Jacobian mode at three (held to 168; a few doubles in scratch memory around the record), all
    with the 18.9 KB LDS pool (eight workgroups per CU).  An unrelated edit moved the commit from 123 to 143 registers in this
    round (3.43 -> 3.85 ms) without a test noticing: this is that test."""
    res = L.kernel_resources(L.assembly("tensor_p3.hip"))
    def one(tag):
        names = [n for n in res if "tp3_point_kernelILi0ELi%d" % tag in n]
        assert len(names) == 1, names
        return res[names[0]]
    for tag, waves, spills in ((0, 4, 0), (2, 4, 0), (1, 3, 32)):          # residual-only, commit, residual+Jacobian
        r = one(tag)
        assert r["spill"] <= spills, (tag, r)
        assert L.waves_per_simd(r, 128) >= waves, (tag, r, L.waves_per_simd(r, 128))
    gather = [res[n] for n in res if "tp3_gather_kernelILi1" in n]
    assert len(gather) == 1 and L.waves_per_simd(gather[0], 256) >= 4, gather
    contract = [res[n] for n in res if "tp3_contract_asm_kernel" in n]
    assert len(contract) == 1 and contract[0]["spill"] == 0 and contract[0]["vgpr"] <= 512, contract


def test_the_generated_contraction_loop_is_current_and_waits_for_what_it_uses():
    """csrc/tp3_contract_loop.inc is generated (csrc/gen_tp3_contract.py): the committed file must be what the generator
    emits, and the generator's own walk over prologue + three unrolled elements + flush must find every register a memory
    / LDS instruction fills waited for before its first use (what the disassembly lint does not see: the hazards it checks
    are those of the matrix and DPP instructions)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_tp3_contract", os.path.join(L.CSRC, "gen_tp3_contract.py"))
    gen = importlib.util.module_from_spec(spec)
    import sys
    keep = sys.dont_write_bytecode
    sys.dont_write_bytecode = True             # (no __pycache__ beside the kernel sources)
    try:
        spec.loader.exec_module(gen)
    finally:
        sys.dont_write_bytecode = keep
    o = gen.generate()
    assert gen.check_schedule(o) == []
    with open(os.path.join(L.CSRC, "tp3_contract_loop.inc")) as f:
        assert f.read() == gen.render(o) + gen.constants_header()
    # and the checker is able to see a missing wait
    k0 = next(k for k, ln in enumerate(o.lines) if ln.startswith(".Ltp3_loop"))
    broken = gen.Out()
    broken.lines = [ln for k, ln in enumerate(o.lines) if not (k > k0 and ln == "s_waitcnt lgkmcnt(0)")]
    assert len(gen.check_schedule(broken)) > 100


_HAZARD = r"""
#include <hip/hip_runtime.h>
typedef double d4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void hazard_kernel(const double* a, const double* b, double* out) {
  d4 c;
  const double x = a[threadIdx.x], y = b[threadIdx.x];
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(c) : "v"(x), "v"(y));
  asm volatile("s_nop %0" :: "n"(PAD));
  double s;
  asm volatile("v_add_f64 %0, %1, %2" : "=v"(s) : "v"(c[0]), "v"(c[1]));
  out[threadIdx.x] = s;
}
"""


@pytest.mark.parametrize("pad", [15, 7, 0])
def test_the_lint_turns_red_on_a_real_hazard(need, pad):
    """an asm matrix instruction whose result a vector instruction reads after pad + 1 < 19 wait states"""
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "hazard.hip")
        open(src, "w").write(_HAZARD)
        asm = L.assembly(src, out=os.path.join(tmp, "hazard.s"), extra_flags=[f"-DPAD={pad}"])
    bad, stats = L.lint_kernel(L.parse_kernel(asm, "hazard_kernel"), need, asm_only=True)
    assert stats["mfma_from_asm"] == 1
    assert bad and pad + 1 <= stats["nearest_valu_read"] < need["valu_read"]     # (the compiler may put a move in between)
    assert any("read by a vector instruction" in what for *_, what in bad)


def test_the_lint_stays_green_with_the_wait_states_in_place(need):
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "hazard.hip")
        open(src, "w").write(_HAZARD.replace('asm volatile("s_nop %0" :: "n"(PAD));', 'asm volatile("s_nop 15\\n\\ts_nop 2");'))
        asm = L.assembly(src, out=os.path.join(tmp, "hazard.s"))
    bad, stats = L.lint_kernel(L.parse_kernel(asm, "hazard_kernel"), need, asm_only=True)
    assert not bad and stats["nearest_valu_read"] >= 19


_DPP = r"""
#include <hip/hip_runtime.h>
extern "C" __global__ void dpp_kernel(const double* a, double* out) {
  double t = a[threadIdx.x], w = a[threadIdx.x + 64], acc = 0.5;
  asm volatile("v_add_f64 %0, %0, %0" : "+v"(t));                 // a vector write of the DPP operand ...
#ifdef FENCE
  asm volatile("s_nop 1" : "+v"(t));
#endif
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(w));   // ... read here
  out[threadIdx.x] = acc;
}
"""


@pytest.mark.parametrize("fence", [False, True])
def test_dpp_operand_hazard(need, fence):
    """the rule the residual column kernel's table registers are held to (RC_DPP_FENCE*): a DPP operand read less than two
    wait states behind the vector instruction that wrote it is flagged, with the `s_nop 1` in between it is not"""
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "dpp.hip")
        open(src, "w").write(_DPP)
        asm = L.assembly(src, out=os.path.join(tmp, "dpp.s"), extra_flags=["-DFENCE"] if fence else ["-DNOFENCE"])
    bad, stats = L.lint_kernel(L.parse_kernel(asm, "dpp_kernel"), need, asm_only=True)
    assert bool(bad) == (not fence)
    if not fence:
        assert any("DPP operand" in what for *_, what in bad)
