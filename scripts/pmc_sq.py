"""Summarise rocprofv3 --pmc passes (scripts/gpu_pmc_sq.sh) per kernel: mean counter value per launch.
usage: pmc_sq.py <dir holding the pass sub-directories> <out.json> [kernel-name substrings ...]"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    root, out = sys.argv[1:3]
    want = sys.argv[3:] or ["lstm_", "gemm_", "stft", "gn_", "mrl1"]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    meta = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(w in k for w in want):
                continue
            k = k[:100]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
            meta[k] = {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                       "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
    res = {}
    for k, cs in acc.items():
        res[k] = dict(meta[k])
        for c, (s, ids) in cs.items():
            res[k]["launches"] = len(ids)
            res[k][c] = s / max(1, len(ids))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k].get("SQ_BUSY_CYCLES", 0) * res[k].get("launches", 1))[:14]:
        v = res[k]
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        print("%-70s x%-3d mfma_busy/busy %.3f  wait_any %.2f  wait_inst %.2f  active %.2f  ldsconf/ldsact %.3f" % (
            k[:70], v.get("launches", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, v.get("SQ_BUSY_CYCLES", 1)) ,
            v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 1))))


if __name__ == "__main__":
    main()
