"""Generates the committed golden vectors (run in the build container: `python tests/golden/make_golden.py`).

 * oracle_small.npz   - seeded inputs and oracle outputs (STFT, BSRNN_SE forward/backward, MR-L1, SI-SNR) at tiny
                        shapes; pins the oracle against silent drift and gives the GPU tests fixed targets.
 * ref_config.npz     - behaviour of the REFERENCE's own baseline_code/config.py (imported from /root/reference):
                        yaml-overrides-CLI precedence and train_tag rule, as data.
 * ref_flow.npz       - (written by make_golden_flow.py) outputs of the reference's own bsrnn_flowse.py / odes.py /
                        sampling via the SURVEY 8(c) shim.
Fixtures are data only (inputs / expected outputs); no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import bsrnn_ref, losses_ref, stft_ref  # noqa: E402

SMALL = dict(N=16, L=1, fs=16000, nsamp=2400, seed=1234)


def small_model(seed=SMALL["seed"]):
    torch.manual_seed(seed)
    m = bsrnn_ref.BSRNN_SE(SMALL["N"], SMALL["L"])
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n or ".0.weight" in n[-12:] and "mlp_" in n or ".0.bias" in n[-10:] and "mlp_" in n:
                p.add_(0.1 * torch.randn_like(p))
    return m


def small_inputs():
    g = torch.Generator().manual_seed(SMALL["seed"] + 1)
    clean = 0.3 * torch.randn(2, SMALL["nsamp"], generator=g)
    noisy = clean + 0.1 * torch.randn(2, SMALL["nsamp"], generator=g)
    return clean, noisy, torch.tensor([SMALL["nsamp"], SMALL["nsamp"] - 500])


def main():
    out = {}
    clean, noisy, lens = small_inputs()
    m = small_model()
    wav, spec = m(noisy, lens, SMALL["fs"])
    loss = losses_ref.mr_l1_loss(clean, wav)
    loss.mean().backward()
    out["clean"], out["noisy"], out["lens"] = clean.numpy(), noisy.numpy(), lens.numpy()
    out["wav"], out["spec"] = wav.detach().numpy(), torch.view_as_real(spec.detach()).numpy()
    out["loss"] = loss.detach().numpy()
    out["sisnr"] = losses_ref.si_snr_loss(clean, wav.detach()).numpy()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    out["param_checksum"] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
    for name in ("bsrnn.bsrnn.band_split.fc.0.weight", "bsrnn.bsrnn.rnn_time.0.weight_hh_l0_reverse",
                 "bsrnn.bsrnn.norm_freq.0.weight", "bsrnn.bsrnn.mask_decoder.mlp_mask.3.3.bias"):
        out["grad:" + name] = dict(m.named_parameters())[name].grad.numpy()
    X, _ = stft_ref.stft(noisy, 320, 160, "hann", lens)
    out["stft320"] = torch.view_as_real(X).numpy()
    out["istft320"] = stft_ref.istft(X, 320, 160, SMALL["nsamp"]).numpy()
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **out)

    ref_cfg = "/root/reference/baseline_code/config.py"
    if os.path.exists(ref_cfg):
        import importlib.util
        spec_ = importlib.util.spec_from_file_location("ref_config", ref_cfg)
        rc = importlib.util.module_from_spec(spec_)
        spec_.loader.exec_module(rc)
        c = rc.Config(learning_rate=5e-4, batch_size=7, config_file="/root/reference/conf/models/BSRNN_baseline.yaml")
        c.read_yaml()
        keys = sorted(k for k, v in vars(c).items() if isinstance(v, (int, float, str, bool)))
        np.savez(os.path.join(HERE, "ref_config.npz"), keys=np.array(keys), values=np.array([str(getattr(c, k)) for k in keys]),
                 model_configs=np.array(str(sorted(c.model_configs.items()))),
                 defaults_keys=np.array(sorted(vars(rc.Config()).keys())))
    print("golden written")


if __name__ == "__main__":
    main()
