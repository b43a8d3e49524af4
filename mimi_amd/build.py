"""In-tree build of libmimi_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmimi_hip.so")
SOURCES = ["domain.hip", "tensor_p3.hip", "contact.hip", "krylov.hip", "exchange.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mimi_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source of the package into mimi_amd/lib/libmimi_hip.so."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # one builder at a time (the ranks of a multi-GPU job all come through here at once); whoever waited finds it done
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(force, verbose, extra_flags)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


# kernels whose inline-asm matrix / DPP instructions nothing in the compiler pads: linted at the disassembly level (isa_lint.py)
# as part of the build -- same compiler, same flags, the very assembly that went into the object (a compiler
# bump or an unrelated edit must not ship a stale-result hazard because a test was not run)
LINT_GATE = {"tensor_p3.hip": ["tp3_contract_kernel", "tp3_contract_asm_kernel", "tp3_point_kernelILi0ELi0", "tp3_point_kernelILi0ELi1",
                               "tp3_point_kernelILi0ELi2"],
             "domain.hip": ["tensor_residual_col_kernel"]}


def lint_gate(rebuilt, objdir, verbose=False):
    todo = [s for s in rebuilt if s in LINT_GATE]
    if not todo or os.environ.get("MIMI_HIP_BUILD_NO_LINT") == "1":
        return
    from . import isa_lint
    need = isa_lint.calibrate()
    for src in todo:
        with open(os.path.join(objdir, src.replace(".hip", ".lint.s"))) as f:
            asm = f.read()
        spills = isa_lint.spill_counts(asm)
        for kernel in LINT_GATE[src]:
            try:
                instrs = isa_lint.parse_kernel(asm, kernel)
            except KeyError:
                continue
            bad, stats = isa_lint.lint_kernel(instrs, need, asm_only=True)
            full = next((n for n in spills if kernel in n), None)
            if verbose:
                print(f"lint {kernel}: {len(bad)} findings, {stats['mfma_from_asm']} asm matrix instructions, "
                      f"nearest reads {stats['nearest_valu_read']} / {stats['nearest_mem']}, spills {spills.get(full)}", flush=True)
            # (the point kernels' asm instructions are ordinary vector ones: a register in scratch memory is no hazard there)
            if bad or (spills.get(full) and "tp3_point_kernel" not in kernel):
                lines = "\n".join(f"  {what}: {dist} < {req}\n    {a}\n    {b}" for a, b, dist, req, what in bad[:10])
                raise RuntimeError(f"{src}: {kernel} fails the ISA hazard lint ({len(bad)} findings, "
                                   f"{spills.get(full)} spilled registers)\n{lines}")


def _build_locked(force, verbose, extra_flags):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mimi_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    rebuilt = []

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj
        # -save-temps=obj: the device assembly that IS assembled into this object stays beside it (<source>.lint.s) -- what
        # the hazard lint below and tests/test_isa_lint_cpu.py read is the shipped code, not a second compilation of it
        cmd = [hipcc] + cflags + ["-save-temps=obj", "-c", path, "-o", obj] + list(extra_flags)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        stem = src.replace(".hip", "")
        for f in os.listdir(objdir):
            if f.startswith(stem + "-h") or f.startswith(stem + ".hip-hip-"):
                full = os.path.join(objdir, f)
                if f == stem + "-hip-amdgcn-amd-amdhsa-gfx950.s":
                    os.replace(full, os.path.join(objdir, stem + ".lint.s"))
                else:
                    os.remove(full)
        rebuilt.append(src)
        return obj

    # one translation unit per source, compiled side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        objs = list(pool.map(compile_one, srcs))
    lint_gate(rebuilt, objdir, verbose)
    cmd = [hipcc] + FLAGS + ["-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
