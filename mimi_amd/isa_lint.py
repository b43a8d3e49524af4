"""Disassembly-level lint for the hand-scheduled fp64 matrix instructions (development / test tool, no GPU).

The kernels of csrc/tensor_p3.hip and csrc/kernels_tensor_wgs*.hpp issue `v_mfma_f64_16x16x4_f64` from inline asm.  LLVM's
hazard recogniser does not look inside inline asm, so nothing in the compiler pads the wait states gfx950 requires between
such an instruction and the first use of its result -- the source does (s_nop pairs, instruction order), and a compiler
bump or an unrelated edit can undo that silently.  This module checks the compiled code itself:

  * hipcc -S of a translation unit -> per kernel a list of instructions with the registers they read and write;
  * from every matrix instruction a walk over the control-flow successors, counting wait states the way the compiler's own
    hazard model does (one per instruction, N + 1 for `s_nop N`), until the requirement is met: the first instruction
    that touches a result register must be far enough away for its class (vector ALU / memory or LDS / another matrix
    instruction reading it as A or B; a matrix instruction taking it as its accumulator input needs none);
  * the reverse hazards the source also pads by hand: a vector instruction writing a register that a matrix instruction
    reads, and a write of EXEC in front of a matrix instruction;
  * the kernels' `vgpr_spill_count` from the code-object metadata.

The required distances are not constants of this file: `calibrate()` compiles four probe kernels that use the BUILTIN
matrix instruction, where the compiler's hazard recogniser does pad, and reads the padding back.
"""
import os
import re
import shutil
import subprocess
import tempfile
from collections import deque

# the compiler the library is built with (mimi_amd/build.py resolves it the same way: what is linted is what ships)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
ASM_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-result", "-S",
             "--cuda-device-only"]

_REG = re.compile(r"\b([vas])(?:(\d+)|\[(\d+):(\d+)\])(?![\w.])")


def assembly(source, out=None, extra_flags=()):
    """hipcc -S --cuda-device-only of csrc/<source> (cached beside the library's objects by modification time)."""
    src = source if os.path.isabs(source) else os.path.join(CSRC, source)
    if out is None:
        objdir = os.path.join(os.path.dirname(CSRC), "lib", "obj")
        os.makedirs(objdir, exist_ok=True)
        out = os.path.join(objdir, os.path.basename(src).rsplit(".", 1)[0] + ".lint.s")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] if src.startswith(CSRC) else [src]
    if extra_flags or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        run = subprocess.run([HIPCC] + ASM_FLAGS + list(extra_flags) + ["-o", out, src], cwd=os.path.dirname(src),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if run.returncode != 0:
            raise RuntimeError(f"{HIPCC} -S {src} failed ({run.returncode}):\n{run.stderr[-4000:]}")
    with open(out) as f:
        return f.read()


class Instr:
    __slots__ = ("op", "text", "line", "defs", "uses", "operands", "nop", "target", "from_asm")

    def __repr__(self):
        return f"{self.line}: {self.text}"


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            lo = hi = int(m.group(2))
        else:
            lo, hi = int(m.group(3)), int(m.group(4))
        for k in range(lo, hi + 1):
            out.add((kind, k))
    return out


# instructions whose FIRST operand is not a destination (stores, LDS writes, compares into vcc are handled below)
_NO_DST = ("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "s_waitcnt", "s_nop", "s_barrier",
           "s_branch", "s_cbranch", "s_endpgm", "s_setprio", "s_sleep", "s_cmp", "s_bitcmp", "v_cmp_", "s_setreg",
           "s_sethalt", "s_trap", "s_icache", "s_dcache", "buffer_wbl2", "buffer_inv", "global_atomic", "ds_add",
           "s_waitcnt_", "s_memtime_dummy")


def parse_kernel(asm, name):
    """instructions of the kernel whose mangled name contains `name` (labels resolved to instruction indices)"""
    m = re.search(r"^(\w*%s\w*):" % re.escape(name), asm, re.M)
    if not m:
        raise KeyError(name)
    a = m.end()
    b = asm.index(".Lfunc_end", a)
    first_line = asm.count("\n", 0, a) + 1
    instrs, labels, pending = [], {}, []
    in_asm = False
    for k, raw in enumerate(asm[a:b].split("\n")):
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            lm = re.match(r"^(\.LBB\d+_\d+):", line)
            if lm:
                pending.append(lm.group(1))
            continue
        lm = re.match(r"^([\w.$]+):$", line)
        if lm:
            pending.append(lm.group(1))
            continue
        ins = Instr()
        ins.text, ins.line, ins.from_asm = line, first_line + k, in_asm
        parts = line.split(None, 1)
        ins.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        ins.operands = [o.strip() for o in rest.split(",")] if rest else []
        ins.nop = int(ins.operands[0], 0) + 1 if ins.op == "s_nop" else 1
        ins.target = None
        if ins.op.startswith(("s_branch", "s_cbranch")):
            ins.target = ins.operands[0]
        if ins.op.startswith(_NO_DST) or not ins.operands:
            ins.defs, ins.uses = set(), _regs(rest)
        else:
            n_dst = 1
            # carry-out forms write a second (scalar) destination; it holds no vector register
            ins.defs = _regs(ins.operands[0])
            ins.uses = _regs(",".join(ins.operands[n_dst:]))
            if ins.op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")) or "dpp" in ins.op and ins.op.startswith("v_fmac"):
                ins.uses |= ins.defs
        for lab in pending:
            labels[lab] = len(instrs)
        pending = []
        instrs.append(ins)
    for ins in instrs:
        if ins.target is not None:
            ins.target = labels.get(ins.target)
    return instrs


def _succ(instrs, i):
    ins = instrs[i]
    if ins.op == "s_endpgm":
        return []
    if ins.op == "s_branch":
        return [ins.target] if ins.target is not None else []
    out = [i + 1] if i + 1 < len(instrs) else []
    if ins.op.startswith("s_cbranch") and ins.target is not None:
        out.append(ins.target)
    return out


def is_mfma(ins):
    return ins.op.startswith("v_mfma")


def is_valu(ins):
    return ins.op.startswith("v_") and not is_mfma(ins)


def is_trans(ins):
    return ins.op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_"))


def is_mem(ins):
    return ins.op.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_"))


def _vec(regs):
    return {r for r in regs if r[0] in "va"}


def _walk(instrs, start, budget):
    """(index, wait states between `start` and that instruction) over every control-flow path, nearest first"""
    best = {}
    queue = deque((s, 0) for s in _succ(instrs, start))
    while queue:
        i, dist = queue.popleft()
        if dist >= budget or best.get(i, 1 << 30) <= dist:
            continue
        best[i] = dist
        for s in _succ(instrs, i):
            queue.append((s, dist + instrs[i].nop))
    return sorted(best.items(), key=lambda t: t[1])


def lint_kernel(instrs, need, asm_only=False):
    """violations of the matrix-instruction hazards in one kernel; `need` = calibrate()'s distances.  asm_only: only the
    pairs in which a matrix instruction comes from inline asm (what the compiler's hazard recogniser cannot see);
    otherwise every pair -- compiler-padded code must pass too, which checks this model against the compiler's.
    Returns (violations, statistics)."""
    bad = []
    stats = {"mfma": 0, "mfma_from_asm": 0, "nearest_valu_read": None, "nearest_valu_write": None, "nearest_mem": None,
             "nearest_mfma_ab": None, "nearest_exec_write": None, "nearest_valu_def": None, "nearest_dpp_def": None,
             "nearest_trans_use": None, "valu_from_asm": 0}

    def note(key, dist):
        if stats[key] is None or dist < stats[key]:
            stats[key] = dist

    def check(first, second, dist, key, req, what):
        note(key, dist)
        if dist < req:
            bad.append((first, second, dist, req, what))

    budget = 4 * max(need["valu_read"], need["mem_read"], need["mfma_srcab"])     # (beyond the requirement: for the statistics)
    for i, ins in enumerate(instrs):
        if is_mfma(ins):
            stats["mfma"] += 1
            stats["mfma_from_asm"] += ins.from_asm
            dst = _vec(ins.defs)
            for j, dist in _walk(instrs, i, budget):
                other = instrs[j]
                reads, writes = _vec(other.uses) & dst, _vec(other.defs) & dst
                if not (reads or writes) or (asm_only and not (ins.from_asm or (is_mfma(other) and other.from_asm))):
                    continue
                if is_mfma(other):
                    src_c = _vec(_regs(other.operands[3])) if len(other.operands) > 3 else set()
                    if _vec(_regs(",".join(other.operands[1:3]))) & dst:
                        check(ins, other, dist, "nearest_mfma_ab", need["mfma_srcab"], "result read as A / B operand of a matrix instruction")
                    elif src_c >= dst and src_c & dst:
                        pass          # accumulator input covering the result: forwarded, no wait states (the compiler emits it so)
                    else:
                        # partial overlap with the accumulator input, or the destination rewritten without being read
                        check(ins, other, dist, "nearest_mfma_ab", need["mfma_srcab"], "result partially overlapped by a matrix instruction")
                elif is_mem(other):
                    check(ins, other, dist, "nearest_mem", need["mem_read"], "result touched by a memory / LDS instruction")
                elif is_valu(other):
                    if reads:
                        check(ins, other, dist, "nearest_valu_read", need["valu_read"], "result read by a vector instruction")
                    else:
                        check(ins, other, dist, "nearest_valu_write", need["valu_write"], "result overwritten by a vector instruction")
        elif is_valu(ins) and _vec(ins.defs):
            w = _vec(ins.defs)
            stats["valu_from_asm"] += ins.from_asm
            if is_trans(ins):
                # (the compiler's own consumers are padded by its hazard recogniser; an inline-asm consumer is not)
                for j, dist in _walk(instrs, i, 4):
                    other = instrs[j]
                    if other.from_asm and is_valu(other) and not is_trans(other) and _vec(other.uses) & w:
                        check(ins, other, dist, "nearest_trans_use", need.get("trans_use", 1),
                              "asm vector instruction reads the result of a transcendental-unit instruction")
            for j, dist in _walk(instrs, i, 4 * max(need["valu_def"], need["dpp_def"])):
                other = instrs[j]
                if is_mfma(other) and _vec(other.uses) & w and (other.from_asm or not asm_only):
                    check(ins, other, dist, "nearest_valu_def", need["valu_def"], "matrix instruction reads a register a vector instruction just wrote")
                # a DPP operand (the first source: the lanes it is taken from are other lanes' registers) written by a vector
                # instruction less than two wait states before (the compiler pads its own DPP instructions, not asm ones)
                if other.from_asm and "_dpp" in other.op and len(other.operands) > 1 and _vec(_regs(other.operands[1])) & w:
                    check(ins, other, dist, "nearest_dpp_def", need["dpp_def"], "DPP operand read behind the vector instruction that wrote it")
        writes_exec = ins.op.startswith("v_cmpx") or (bool(ins.operands) and not ins.op.startswith(_NO_DST) and
                                                      re.match(r"exec(_lo|_hi)?$", ins.operands[0]) is not None)
        if writes_exec:
            # behind a scalar write the compiler pads nothing (and its own code is fine at 6): the bar below is the one
            # measured for ASM matrix instructions in round 3, and applies to those only
            vector = ins.op.startswith("v_")
            req = need["exec_valu"] if vector else need["exec_salu"]
            for j, dist in _walk(instrs, i, 2 * req):
                other = instrs[j]
                if is_mfma(other) and (other.from_asm or (vector and not asm_only)):
                    check(ins, other, dist, "nearest_exec_write", req, "matrix instruction behind a write of EXEC")
    return bad, stats


def spill_counts(asm):
    """{kernel name: vgpr_spill_count} from the code-object metadata"""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", asm):
        out[m.group(1)] = int(m.group(2))
    return out


def kernel_resources(asm):
    """{kernel name: {"vgpr", "agpr", "lds", "spill"}} from the code-object metadata (what decides the waves per SIMD: 512
    registers per lane and SIMD, 160 KB of LDS per CU on gfx950)"""
    out = {}
    md = asm[asm.index("amdhsa.kernels"):] if "amdhsa.kernels" in asm else ""
    for ent in md.split("  - .agpr_count")[1:]:
        ent = ".agpr_count" + ent
        get = lambda key: re.search(r"\.%s:\s*(\S+)" % key, ent)
        name = get("name")
        if not name:
            continue
        out[name.group(1)] = {"agpr": int(get("agpr_count").group(1)), "vgpr": int(get("vgpr_count").group(1)),
                              "lds": int(get("group_segment_fixed_size").group(1)), "spill": int(get("vgpr_spill_count").group(1))}
    return out


def waves_per_simd(res, threads_per_workgroup):
    """resident waves per SIMD a kernel's registers and LDS allow (gfx950: 512 registers per lane, allocated in blocks of 8;
    160 KB LDS per CU; at most 8 waves per SIMD)"""
    regs = -(-max(res["vgpr"], 1) // 8) * 8
    by_regs = min(8, 512 // regs)
    waves_per_wg = max(1, threads_per_workgroup // 64)
    wgs_by_lds = (160 * 1024) // res["lds"] if res["lds"] else 10 ** 6
    by_lds = wgs_by_lds * waves_per_wg / 4.0
    return min(by_regs, by_lds)


_PROBE = r"""
#include <hip/hip_runtime.h>
typedef double d4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void probe_valu_read(const double* a, const double* b, double* out) {
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = c[0] * 3.0 + c[1] + c[2] + c[3];
}
extern "C" __global__ void probe_mem_read(const double* a, const double* b, d4* out) {
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = c;
}
extern "C" __global__ void probe_mfma_srcab(const double* a, const double* b, d4* out) {
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(c[0], b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = e;
}
extern "C" __global__ void probe_valu_def(const double* a, const double* b, d4* out) {
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x] * 3.0, b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = c;
}
"""


def calibrate():
    """the wait states THIS compiler pads around the builtin fp64 matrix instruction on gfx950, read back from probe
    kernels: result -> vector read, -> memory read, -> A / B operand of the next matrix instruction; vector write ->
    matrix read.  EXEC: 4 wait states behind a vector write (the compiler's rule for matrix instructions); behind a
    scalar write the 8 wait states measured on the hardware in round 3 (an `s_or_b64 exec` directly in front
    of an asm matrix instruction did not reach it)."""
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "probe.hip")
        with open(src, "w") as f:
            f.write(_PROBE)
        asm = assembly(src, out=os.path.join(tmp, "probe.s"))

    def gap(kernel, first, second):
        ins = parse_kernel(asm, kernel)
        i = next(k for k, x in enumerate(ins) if x.op.startswith(first))
        dist = 0
        for x in ins[i + 1:]:
            if x.op.startswith(second):
                return dist
            dist += x.nop
        raise RuntimeError(f"probe {kernel}: no {second} behind {first}")

    need = {"valu_read": gap("probe_valu_read", "v_mfma", "v_f"), "mem_read": gap("probe_mem_read", "v_mfma", "global_store"),
            "mfma_srcab": gap("probe_mfma_srcab", "v_mfma", "v_mfma"), "valu_def": gap("probe_valu_def", "v_mul_f64", "v_mfma"),
            # a vector instruction that only OVERWRITES a result register: LLVM's DMFMA16x16WriteVgprVALUWriteWaitStates (not
            # probed: no source construct pins the registers); the compiler-padded kernels passing the lint bound it below
            "valu_write": 11,
            # vector write -> DPP read of the same register: 2 wait states (the ISA's rule for every DPP instruction)
            "dpp_def": 2,
            # result of a transcendental-unit instruction (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) read by an
            # ordinary vector instruction: one wait state on gfx940+ (LLVM pads its own instructions; not inline asm ones)
            "trans_use": 1,
            "exec_valu": 4, "exec_salu": 8}
    return need


def report(source, kernels, need=None, asm_only=False):
    need = need or calibrate()
    asm = assembly(source)
    spills = spill_counts(asm)
    out = {}
    for k in kernels:
        instrs = parse_kernel(asm, k)
        bad, stats = lint_kernel(instrs, need, asm_only)
        full = next((n for n in spills if k in n), None)
        stats["vgpr_spill_count"] = spills.get(full)
        out[k] = (bad, stats)
    return out


if __name__ == "__main__":
    import sys
    need = calibrate()
    print("required wait states (from the compiler's own padding):", need)
    for src, kernels in (("tensor_p3.hip", ["tp3_contract_kernel"]),
                         ("domain.hip", ["tensor_wgsym_kernel", "tensor_wgs_kernel"])):
        if len(sys.argv) > 1 and sys.argv[1] not in src:
            continue
        for k, (bad, stats) in report(src, kernels, need).items():
            print(k, stats)
            for first, second, dist, req, what in bad[:20]:
                print(f"  VIOLATION {what}: {dist} < {req}\n    {first}\n    {second}")
