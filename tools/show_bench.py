"""One-line digest of a bench.py JSON line: python tools/show_bench.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("%s %.0f QP/s  ms/step %.3f  lone launch %.3f ms (%.0f QP/s kernel-only)  fp64 frac %.4f (dense count %.4f)  iters mean %.0f max %d  solved %d/%d" % (
    d.get("n_gpus"), d["value"], d["ms_per_step"], r["launch_ms"], d["kernel_only"]["value"], r["frac"], r["frac_dense_count"],
    r["iterations_mean"], r["iterations_max"], d["solver"]["solved"], d["solver"]["problems"]))
