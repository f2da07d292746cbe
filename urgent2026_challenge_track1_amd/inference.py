#!/usr/bin/env python
"""Inference entry point: ``baseline_code/inference.py:26-112`` surface (``--input_scp --output_dir --ckpt_path
--device``): per-file full-length enhancement, peak-normalise to 0.9 (:60), write ``wav/{uid}.wav`` and ``inf.scp``."""
import argparse
import os

import torch

from . import ops
from ._lib import call, stream_ptr
from .config import Config
from .d_model import SEModel
from .dataset import read_audio, write_audio


def load_from_checkpoint(path, map_location="cuda"):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    cfg = ck.get("hyper_parameters", {}).get("cfg", None)
    if cfg is None:
        cfg = Config(model_configs={"num_channel": 196, "num_layer": 6})
    elif not isinstance(cfg, Config):       # a Lightning checkpoint of the reference pickles its own Config class
        cfg = Config(**{k: v for k, v in vars(cfg).items()})
    model = SEModel(cfg)
    sd = ck["state_dict"] if "state_dict" in ck else ck
    model.se_model.load_state_dict({k[len("se_model."):] if k.startswith("se_model.") else k: v for k, v in sd.items()})
    return model.to(map_location)


def enhance_file(model, wav_np, sr, device):
    wav = torch.as_tensor(wav_np).float().to(device).view(1, -1)
    length = torch.tensor([wav.shape[-1]])
    with torch.no_grad():
        enhanced, _ = model.se_model(wav, length, sr)
        enhanced = enhanced.contiguous()
        scratch = torch.empty(1, dtype=torch.int32, device=enhanced.device)
        call("peak_normalize", enhanced, enhanced.numel(), 0.9, scratch, stream_ptr())
    return enhanced


def main(args):
    model = load_from_checkpoint(args.ckpt_path, args.device)
    model.eval()
    input_audios = {}
    with open(args.input_scp) as f:
        for line in f:
            utt, wav = line.strip().split()
            input_audios[utt] = wav
    os.makedirs(args.output_dir + "/wav", exist_ok=True)
    with open(args.output_dir + "/inf.scp", "w") as f:
        for uid, wav_path in input_audios.items():
            wav, sr = read_audio(wav_path)
            enhanced = enhance_file(model, wav, sr, args.device)
            write_audio(args.output_dir + "/wav/%s.wav" % uid, enhanced.cpu().numpy().flatten(), sr)
            print("%s %s/wav/%s.wav" % (uid, args.output_dir, uid), file=f)
    print("done")


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_scp", type=str, required=True)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--ckpt_path", type=str, required=True)
    p.add_argument("--device", type=str, default="cuda")
    return p


if __name__ == "__main__":
    main(parser().parse_args())
