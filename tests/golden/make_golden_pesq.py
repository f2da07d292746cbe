"""Regression vectors of the PESQ oracle (oracle/pesq_ref.py) on the seeded pairs of tests/pesq_cases.py.  NOT a pin to the
reference (pesq==0.0.4 is absent from the image: see the oracle's header): they keep the oracle from drifting and let the CPU
suite check it in seconds.  Writes tests/golden/pesq_oracle.npz."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pesq_ref  # noqa: E402
from tests import pesq_cases  # noqa: E402

out = {"mos": [], "raw": [], "trace": [], "mos_f32": [], "trace_f32": []}
for i in range(len(pesq_cases.CASES)):
    fs, mode, ref, deg = pesq_cases.make_case(i)
    mos, tr = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True)
    out["mos"].append(float(mos))
    out["raw"].append(float(tr.get("raw", np.nan)))
    out["trace"].append(json.dumps({k: tr.get(k) for k in pesq_cases.TRACE_KEYS}, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))
    # the same pair with every stored buffer rounded to float32 as the ITU code's C floats are (pesq_ref.q)
    mos32, tr32 = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True, precision="f32")
    out["mos_f32"].append(float(mos32))
    out["trace_f32"].append(json.dumps({k: tr32.get(k) for k in pesq_cases.TRACE_KEYS}, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))
    assert out["trace_f32"][-1] == out["trace"][-1], i          # storage precision moves no integer decision on these pairs
    print(i, pesq_cases.CASES[i], "mos %.4f (f32 storage %+.1e)" % (mos, mos32 - mos), out["trace"][-1])
np.savez(os.path.join(HERE, "pesq_oracle.npz"), mos=np.array(out["mos"]), raw=np.array(out["raw"]), trace=np.array(out["trace"]),
         mos_f32=np.array(out["mos_f32"]), trace_f32=np.array(out["trace_f32"]))
