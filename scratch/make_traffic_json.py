"""profiles/rNN_traffic.json from rocprofv3 PMC passes (run on the GPU box by scratch/profile_round.sh).
usage: python scratch/make_traffic_json.py OUT.json KEY:DIR_PIPE:DIR_FETCH:DIR_WRITE:ELEMENTS:B_ALG [...]
Per kernel and dispatch: FETCH_SIZE (KB, doubled: gfx950 tallies 128-B requests at 64 B -- MI355X guide, HBM section; the
factor holds for this code's 8-byte per-lane accesses, profiles/r01_pmc_calibration.txt), WRITE_SIZE (KB, exact),
SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE and instruction counts."""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def per_kernel(d):
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mimi_hip::", "").split("(")[0]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k].add(row["Dispatch_Id"])
    return {k: {c: v / len(cnt[k]) for c, v in acc[k].items()} for k in acc}


out = {}
for spec in sys.argv[2:]:
    key, d_pipe, d_fetch, d_write, elements, balg = spec.split(":")
    pipe, fetch, write = per_kernel(d_pipe), per_kernel(d_fetch), per_kernel(d_write)
    grad = "/grad/" in key
    names = [k for k in fetch if k.startswith(("tensor_", "tp3_")) and (("<0>" not in k and ", 0>" not in k) or "wgsym" in k or "p2" in k)] \
        if grad else [k for k in fetch if k.startswith(("tensor_", "tp3_"))]
    # kernels of one residual+Jacobian step: the integration / point / contraction kernels and the gather with the tangent
    # (", 2>": the state-commit mode of the pre-pass kernel, DomainPostTimeAdvance -- timed by bench.py after the step)
    step = [k for k in names if not (k.startswith("tp3_point_kernel<0, 0>") or k.startswith("tp3_point_kernel<1, 0>") or k.startswith("tp3_gather_kernel<0>")
                                     or k.startswith("tensor_residual") or ", 2>" in k)]
    if not grad:      # the residual-only assembly (bench.py --residual-only): its own kernels
        step = [k for k in names if k.startswith(("tp3_point_kernel<0, 0>", "tp3_gather_kernel<0>", "tensor_residual"))]
    f_kb = {k: fetch[k].get("FETCH_SIZE", 0.0) for k in step}
    w_kb = {k: write[k].get("WRITE_SIZE", 0.0) for k in step}
    total = sum(2 * f_kb[k] + w_kb[k] for k in step) * 1024
    busy = {}
    for k in step:
        p = pipe.get(k, {})
        if p.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the busy counter over the 1024 SIMDs
            cycles = p["GRBM_GUI_ACTIVE"] / 8.0
            busy[k] = dict(mfma_busy_frac=p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cycles,
                           valu_instructions_per_element=p.get("SQ_INSTS_VALU", 0.0) / float(elements),
                           mfma_instructions_per_element=p.get("SQ_INSTS_MFMA", 0.0) / float(elements),
                           active_inst_frac=p.get("SQ_ACTIVE_INST_ANY", 0.0) / max(p.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                           wait_inst_frac=p.get("SQ_WAIT_INST_ANY", 0.0) / max(p.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                           wait_any_frac=p.get("SQ_WAIT_ANY", 0.0) / max(p.get("SQ_WAVE_CYCLES", 1.0), 1.0))
    out[key] = dict(bytes_per_step=total, fetch_size_kb_raw=f_kb, write_size_kb=w_kb, pipe=busy,
                    algorithmic_bytes_per_step=int(balg) * int(elements),
                    kernel_sources_sha=bench.kernel_sources_sha(),
                    correction="FETCH_SIZE x2 (gfx950), WRITE_SIZE exact",
                    source="rocprofv3 --pmc, separate passes for FETCH_SIZE, WRITE_SIZE and the SQ / GRBM counters; bench.py --steps 2 --warmup 1")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: (v["bytes_per_step"], v["algorithmic_bytes_per_step"]) for k, v in out.items()}))
