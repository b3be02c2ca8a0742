"""Print the GPU timeline (kernels and copies, start / duration in ms relative to the step's first event) of the LAST
pipelined step in a rocprofv3 --kernel-trace --memory-copy-trace csv directory.  Dev tool."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:48], r.get("Grid_Size", ""), r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy"))[:24], r.get("Bytes", ""), r.get("Stream_Id", "")))
ev.sort()
if not ev:
    print("no events"); sys.exit(0)
# steps are separated by gaps > 1 ms with nothing running: find the last burst that contains >= 10 solver kernels
bursts, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] > end + 400_000:
        bursts.append(cur); cur = []
    cur.append(e); end = max(end or 0, e[1])
bursts.append(cur)
cands = [b for b in bursts if sum(1 for e in b if e[2] == "K" and "admm" in e[3]) >= 4]
b = cands[-2] if len(cands) >= 2 else cands[-1]
t0 = b[0][0]
print("burst: %d events, %.3f ms" % (len(b), (max(e[1] for e in b) - t0) / 1e6))
agg = {}
for s, e, k, n, g, q in b:
    if k == "K" or (e - s) > 50_000:
        print("%9.3f %8.3f %s %-48s grid %-8s q %s" % ((s - t0) / 1e6, (e - s) / 1e6, k, n, g, q))
    agg.setdefault((k, n), [0, 0]); agg[(k, n)][0] += 1; agg[(k, n)][1] += e - s
print("-- totals")
for (k, n), (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%s %-48s x%-4d %9.3f ms" % (k, n, c, t / 1e6))
