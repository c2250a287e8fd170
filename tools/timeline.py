"""Where the non-decode kernels of one Inference_Step sit in time (from a rocprofv3 --kernel-trace csv): name, start and end in
microseconds since the step's first kernel, for everything that is not one of the four decode-step kernels."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
# the last Inference_Step: walk back from the end to the last gt_conv_wino / postnet, then to the previous encoder start
dec = ("gt_dec_front", "gt_lstm_x_kernel", "gt_proj_lean")
last_end = rows[-1][1]
# find the start of the last step: the last kernel whose name has 'gt_rng_fill' or first conv after a gap > 200 us
seeds = [i for i, r in enumerate(rows) if "gt_set_seed_kernel" in r[2]]
i0 = max(0, seeds[-1] - 12) if seeds else 0
while i0 < len(rows) and "copyBuffer" not in rows[i0][2] and "gt_" not in rows[i0][2]:
    i0 += 1
t0 = rows[i0][0]
prev_dec = 0
for s, e, n in rows[i0:]:
    if any(d in n for d in dec):
        prev_dec += 1
        continue
    if prev_dec:
        print("   ... %d decode-step kernels ..." % prev_dec)
        prev_dec = 0
    print("%9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:90]))
print("step end %.1f us" % ((last_end - t0) / 1e3))
