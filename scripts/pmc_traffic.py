"""Turn two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; each with --kernel-trace only, csv output) of
`bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-metrics` into profiles/rNN_pmc_hbm_traffic_vXX.json:
HBM bytes per launch per kernel.  FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled for gfx950 as
MI355X_MICROARCH.md (HBM) prescribes for wide streaming reads.
usage: pmc_traffic.py <dir of the FETCH pass> <dir of the WRITE pass> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter):
    per = defaultdict(lambda: [0, 0.0])        # kernel -> [dispatches, sum KiB]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under %s" % d
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"][:60]
            per[k][1] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                per[k][0] += 1
    return per


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    res = {}
    for k in fe:
        n = max(1, fe[k][0])
        res[k] = {"launches": n, "fetch_GB": round(fe[k][1] * 1024 * 2 / n / 1e9, 3),
                  "write_GB": round(wr.get(k, [1, 0.0])[1] * 1024 / max(1, wr.get(k, [1, 0.0])[0]) / 1e9, 3)}
    json.dump(res, open(out, "w"), indent=0)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_GB"] + kv[1]["write_GB"]) * kv[1]["launches"])[:12]:
        print("%-62s x%-4d fetch %.3f GB  write %.3f GB" % (k, v["launches"], v["fetch_GB"], v["write_GB"]))


if __name__ == "__main__":
    main()
