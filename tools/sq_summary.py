"""Derived view of the SQ counters of the headline kernel: python tools/sq_summary.py <tag> <iters_mean> [out.json]
Reads profiles/<tag>_pmc_summary.json (tools/summarise_profile.py) and writes per-wave, per-iteration instruction counts
and the shares of a wave's resident cycles (profiles/r03_bench_sq_summary.json is the headline kernel's)."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, iters = sys.argv[1], float(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles", "r03_bench_sq_summary.json")
s = json.load(open(os.path.join(root, "profiles", f"{tag}_pmc_summary.json")))
m = {k: v["mean"] for k, v in s.items() if k.startswith("SQ_")}
waves, wc = m["SQ_WAVES"], m["SQ_WAVE_CYCLES"]
per = lambda k: m[k] / waves / iters
d = {
    "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc, two passes of 8 counters each, tools/profile_r03e.sh; means over the bench's launches)",
    "kernel": s["_kernel"],
    "iterations_mean_per_problem": iters,
    "raw_mean_per_launch": m,
    "per_wave_per_iteration": {
        "valu_instructions": per("SQ_INSTS_VALU"), "salu_instructions": per("SQ_INSTS_SALU"), "lds_instructions": per("SQ_INSTS_LDS"),
        "vmem_instructions": per("SQ_INSTS_VMEM"), "mfma_instructions": per("SQ_INSTS_MFMA"), "wave_cycles_quads": wc / waves / iters,
    },
    "share_of_wave_resident_cycles": {
        "executing_any_instruction": m["SQ_ACTIVE_INST_ANY"] / wc, "executing_valu": m["SQ_ACTIVE_INST_VALU"] / wc,
        "executing_scalar": m["SQ_ACTIVE_INST_SCA"] / wc, "executing_lds": m["SQ_ACTIVE_INST_LDS"] / wc,
        "waiting_on_a_counter_s_waitcnt": m["SQ_WAIT_ANY"] / wc, "waiting_for_issue": m["SQ_WAIT_INST_ANY"] / wc,
        "waiting_for_issue_lds": m["SQ_WAIT_INST_LDS"] / wc,
    },
    "mfma_pipe_busy_share_of_sq_busy_cycles_x4": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_BUSY_CYCLES"]),
    "reading": sys.argv[4] if len(sys.argv) > 4 else "two waves per SIMD: the vector ALU of a SIMD is busy ~2 x executing_valu of the time; no single wait dominates",
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d["per_wave_per_iteration"]), json.dumps(d["share_of_wave_resident_cycles"]))
