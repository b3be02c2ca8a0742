import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"].get("batches_in_flight_per_gpu"), round(d["value"]), "QP/s  ms/step %.3f  kernel_avg_ms %.3f  iters mean %.0f max %d  solved %d/%d" % (
    d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["valu"]["iterations_mean"], d["valu"]["iterations_max"], d["solver"]["solved"], d["solver"]["problems"]))
