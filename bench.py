#!/usr/bin/env python
"""bench.py -- hot-path throughput on MI355X (see DESIGN.md "Measurement").

python bench.py --gpus N --steps K --warmup W     (N>1 is launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  A "step" is one pass of the hot path over one batch of synthetic
4 s @ 48 kHz utterances that already sit in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16


def synth_batch(B, L, fs, seed, device):
    """SURVEY 8(d) generator: low-passed noise x syllabic envelope, 0.4 s near-silence at both ends, peak 0.9;
    noisy = clean + white noise at U(-5, 20) dB SNR, jointly peak-normalised."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = torch.randn(B, L, generator=g)
    # one-pole low-pass a = 0.95 via FFT-domain response (deterministic, cheap on host)
    k = torch.fft.rfftfreq(L)
    Hf = 1.0 / (1.0 - 0.95 * torch.exp(-2j * torch.pi * k))
    clean = torch.fft.irfft(torch.fft.rfft(n) * Hf, n=L)
    clean = clean / clean.std(dim=1, keepdim=True)
    t = torch.arange(L) / fs
    phi = torch.rand(B, 1, generator=g) * 6.2831853
    env = 0.55 + 0.45 * torch.sin(2 * torch.pi * 4.0 * t[None] + phi)
    edge = int(0.4 * fs)
    gate = torch.ones(L)
    gate[:edge] = 1e-3
    gate[L - edge:] = 1e-3
    clean = clean * env * gate
    clean = 0.9 * clean / clean.abs().amax(dim=1, keepdim=True)
    snr = torch.rand(B, 1, generator=g) * 25.0 - 5.0
    noise = torch.randn(B, L, generator=g)
    p_c = (clean ** 2).mean(dim=1, keepdim=True)
    p_n = (noise ** 2).mean(dim=1, keepdim=True)
    noise = noise * torch.sqrt(p_c / (p_n * 10 ** (snr / 10)))
    noisy = clean + noise
    sc = 0.9 / torch.maximum(noisy.abs().amax(dim=1, keepdim=True), clean.abs().amax(dim=1, keepdim=True))
    return (clean * sc).to(device), (noisy * sc).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from urgent2026_challenge_track1_amd import ops
    fs, L, B = 48000, 192000, args.batch
    n_fft, hop = 960, 480
    clean, noisy = synth_batch(B, L, fs, 2024 + rank, dev)

    def step():
        spec = ops.stft_forward(noisy, n_fft, hop)
        wav = ops.istft_forward(spec, n_fft, hop, L)
        return wav

    for _ in range(args.warmup):
        step()
    # kernel-level timing of the STFT launch with events on the launch stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()
        spec = ops.stft_forward(noisy, n_fft, hop)
        evs[i][1].record()
        ops.istft_forward(spec, n_fft, hop, L)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    k_ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    T, Fb = L // hop + 1, n_fft // 2 + 1
    alg_bytes = B * (L * 4 + T * Fb * 8)
    ach = alg_bytes / (k_ms * 1e-3) / 1e9

    out = {
        "metric": "utterances/sec (4 s @ 48 kHz), PARTIAL path: STFT+iSTFT only (train step under construction)",
        "value": world * B * args.steps / dt, "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "B%d x 4 s @ 48 kHz, n_fft 960 hop 480, stft->istft" % B, "per_gpu_batch": B},
        "roofline": {"bound": "hbm", "kernel": "stft_kernel<0>", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import stft_ref
        x = noisy[:8].cpu()
        torch.set_num_threads(os.cpu_count())
        stft_ref.stft(x, n_fft, hop)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            X, _ = stft_ref.stft(x, n_fft, hop)
            stft_ref.istft(X, n_fft, hop, L)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 8 * reps / cdt, "unit": "utt/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "oracle torch.stft+istft, 8 utt x %d reps" % reps}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
