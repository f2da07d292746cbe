"""Parity figures the GPU tests observe, kept as a JSON artefact: `record(name, **figures)` merges them into
gpurun_out/parity/c2_parity.json (on the GPU box that directory travels back with gpurun).  The builder copies the file to
profiles/rNN_c2_parity.json, and bench.py quotes the newest committed copy in its `parity` key - so the figures in the bench
line are what a test run measured, not constants typed into bench.py."""
import json
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity", "c2_parity.json")


def record(name, **figures):
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    try:
        with open(PATH) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    figures["recorded_unix"] = int(time.time())
    data[name] = figures
    try:
        import torch
        data["_device"] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu"
    except Exception:
        pass
    with open(PATH, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
