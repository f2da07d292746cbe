#!/usr/bin/env python
"""Batched intrusive-metric evaluation: CLI / file formats of
``evaluation_metrics/calculate_intrusive_se_metrics.py:114-208`` (``--ref_scp --inf_scp --output_dir --nj
--chunksize``; writes ``{METRIC}.scp`` lines ``"uid value"`` and ``RESULTS.txt`` lines ``"METRIC: mean:.4f"`` of the
nan-mean; ``METRICS = ("PESQ", "ESTOI")`` :15, a pair without utterances scores PESQ nan :160-162).  Instead of a process
pool of per-pair CPU workers (:127-132) pairs are grouped by (fs, length) and each group is scored in batched GPU passes
(one workgroup per pair for PESQ).  SDR (``sdr_metric`` :90-109, defined but unused there) is available with
``--metrics PESQ ESTOI SDR``.  Several GPUs: pairs are split ``i % world`` with no device collective (``--rank / --world``,
default from RANK / WORLD_SIZE); each rank writes ``{METRIC}.scp.rank{r}`` and rank 0 merges once all parts exist.
"""
import argparse
import os
import time
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from . import metrics
from .dataset import read_audio

METRICS = ("PESQ", "ESTOI")       # calculate_intrusive_se_metrics.py:15


def score_pairs(pairs, device="cuda", max_batch=2048, metric_names=METRICS):
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
            # PESQ on this stream, ESTOI / SDR beside it on a second one (metrics.score_batch); 2,048 pairs per PESQ launch
            # amortise the tail of its slowest pair
            res = {m: v.cpu().numpy() for m, v in metrics.score_batch(r, e, fs, metric_names).items()}
            for j, c in enumerate(chunk):
                out[c[0]] = {m: float(v[j]) for m, v in res.items()}
    return out


def _merge(outdir, names, world, timeout_s=3600.0):
    t0 = time.time()
    parts = [[outdir / ("%s.scp.rank%d" % (m, r)) for r in range(world)] for m in names]
    while not all(p.exists() for row in parts for p in row):
        if time.time() - t0 > timeout_s:
            raise TimeoutError("waiting for the other ranks' score files in %s" % outdir)
        time.sleep(0.5)
    rows = {m: [] for m in names}
    for m, row in zip(names, parts):
        for p in row:
            for line in p.read_text().splitlines():
                idx, uid, val = line.split()
                rows[m].append((int(idx), uid, float(val)))
        rows[m].sort()
    return rows


def main(args):
    rank = int(os.environ.get("RANK", "0")) if args.rank is None else args.rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.world is None else args.world
    names = tuple(args.metrics)
    refs = {}
    with open(args.ref_scp, "r") as f:
        for line in f:
            uid, audio_path = line.strip().split()
            refs[uid] = audio_path
    mine = []
    with open(args.inf_scp, "r") as f:
        for i, line in enumerate(f):
            uid, audio_path = line.strip().split()
            if i % world != rank:                        # pairs split i % world, no collective
                continue
            ref, fs = read_audio(refs[uid])
            inf, fs2 = read_audio(audio_path)
            assert fs == fs2, (fs, fs2)
            mine.append((i, uid, ref[0], inf[0], fs))
    device = args.device
    if device == "cuda" and world > 1:
        device = "cuda:%d" % (int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count()))
    scores = score_pairs([(uid, r, e, fs) for _, uid, r, e, fs in mine], device, metric_names=names)
    outdir = Path(args.output_dir)
    outdir.mkdir(parents=True, exist_ok=True)
    if world > 1:
        for m in names:
            tmp = outdir / ("%s.scp.rank%d.tmp" % (m, rank))
            with tmp.open("w") as f:
                for i, uid, *_ in mine:
                    f.write("%d %s %r\n" % (i, uid, scores[uid][m]))
            tmp.rename(outdir / ("%s.scp.rank%d" % (m, rank)))
        if rank != 0:
            return
        rows = _merge(outdir, names, world)
    else:
        rows = {m: [(i, uid, scores[uid][m]) for i, uid, *_ in mine] for m in names}
    for m in names:
        with (outdir / ("%s.scp" % m)).open("w") as f:
            for _, uid, v in rows[m]:
                f.write("%s %s\n" % (uid, v))
    with (outdir / "RESULTS.txt").open("w") as f:
        for m in names:
            f.write("%s: %.4f\n" % (m, np.nanmean([v for _, _, v in rows[m]])))
    print("Overall results have been written in %s" % (outdir / "RESULTS.txt"), flush=True)


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--ref_scp", type=str, required=True)
    p.add_argument("--inf_scp", type=str, required=True)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--nj", type=int, default=8, help="kept for CLI compatibility (pairs are batched on the GPU)")
    p.add_argument("--chunksize", type=int, default=1000)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--metrics", nargs="+", default=list(METRICS), choices=["PESQ", "ESTOI", "SDR"])
    p.add_argument("--rank", type=int, default=None, help="this process's share of the pairs (default: RANK)")
    p.add_argument("--world", type=int, default=None, help="number of processes sharing the pairs (default: WORLD_SIZE)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())
