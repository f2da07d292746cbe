"""Condenses a rocprofv3 kernel trace (csv) into the sequence of one train step: python scripts/trace_step.py <kernel_trace.csv> <out.txt>
The step is delimited by the optimizer kernel (clip_adamw_slots_kernel); per dispatch: queue, start offset (us), duration (us), gap to the
previous dispatch of the same queue (us), short kernel name."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = lambda r, *names: next(r[n] for n in names if n in r)
ev = sorted(((int(k(r, "Start_Timestamp", "start_timestamp")), int(k(r, "End_Timestamp", "end_timestamp")), k(r, "Queue_Id", "queue_id"),
              k(r, "Kernel_Name", "kernel_name")) for r in rows), key=lambda e: e[0])
opt = [i for i, e in enumerate(ev) if "clip_adamw" in e[3]]
lo, hi = opt[-2] + 1, opt[-1] + 1
step = ev[lo:hi]
t0 = step[0][0]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void |urse::|at::native::", "", n))[:70]
last = {}
with open(sys.argv[2], "w") as f:
    f.write("step: %d dispatches, %.2f ms\n" % (len(step), (step[-1][1] - t0) / 1e6))
    busy, gaps = {}, {}
    for s, e, q, n in step:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = max(e, last.get(q, 0))
        busy[q] = busy.get(q, 0) + (e - s) / 1e3
        if gap > 0: gaps[q] = gaps.get(q, 0) + gap
        f.write("q%s %10.1f %9.1f %7.1f  %s\n" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(n)))
    for q in busy:
        f.write("queue %s: busy %.2f ms, gaps %.2f ms\n" % (q, busy[q] / 1e3, gaps.get(q, 0) / 1e3))
    cnt = {}
    for s, e, q, n in step:
        c = cnt.setdefault(short(n), [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
    for n, (c, t) in sorted(cnt.items(), key=lambda x: -x[1][1]):
        f.write("%5d x %9.1f us  %s\n" % (c, t, n))
