"""Golden vectors produced BY THE REFERENCE'S OWN CODE (run in the build container only):
``/root/reference/baseline_code/models/bsrnn_flowse.py`` (with the SURVEY 8(c) espnet shim: only
``choose_norm``/``choose_norm1d`` -> ``nn.GroupNorm(1, C)`` are given behaviour), ``models/odes.py`` and
``sampling/`` are imported, run on seeded inputs, checked against oracle/flow_ref.py, and the inputs / outputs are
stored in tests/golden/ref_flow.npz.  Only data is stored, never reference source.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import flow_ref  # noqa: E402

CFG = dict(input_dim=769, num_channel=16, num_layer=2, B=2, T=9, seed=77)


def load_reference():
    names = ["espnet2", "espnet2.enh", "espnet2.enh.encoder", "espnet2.enh.encoder.stft_encoder", "espnet2.enh.decoder",
             "espnet2.enh.decoder.stft_decoder", "espnet2.enh.separator", "espnet2.enh.separator.bsrnn_separator",
             "espnet2.enh.diffusion", "espnet2.enh.diffusion.sdes", "espnet2.enh.diffusion.score_based_diffusion",
             "espnet2.enh.diffusion.abs_diffusion", "espnet2.enh.layers", "espnet2.enh.layers.bsrnn"]
    for n in names:
        sys.modules.setdefault(n, types.ModuleType(n))
    dummy = type("Dummy", (nn.Module,), {})
    sys.modules["espnet2.enh.encoder.stft_encoder"].STFTEncoder = dummy
    sys.modules["espnet2.enh.decoder.stft_decoder"].STFTDecoder = dummy
    sys.modules["espnet2.enh.separator.bsrnn_separator"].BSRNNSeparator = dummy
    for k in ("OUVESDE", "OUVPSDE", "SDE"):
        setattr(sys.modules["espnet2.enh.diffusion.sdes"], k, dummy)
    sys.modules["espnet2.enh.diffusion.score_based_diffusion"].ScoreModel = dummy
    sys.modules["espnet2.enh.diffusion.abs_diffusion"].AbsDiffusion = dummy
    lay = sys.modules["espnet2.enh.layers.bsrnn"]
    lay.MaskDecoder = dummy
    lay.choose_norm = lambda t, c, *a, **k: nn.GroupNorm(1, c)
    lay.choose_norm1d = lambda t, c, *a, **k: nn.GroupNorm(1, c)

    def by_path(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    net = by_path("ref_bsrnn_flowse", REF + "/baseline_code/models/bsrnn_flowse.py")
    odes = by_path("ref_odes", REF + "/baseline_code/models/odes.py")
    sys.path.insert(0, REF)
    import baseline_code.sampling as sampling
    return net, odes, sampling


def main():
    net, odes, sampling = load_reference()
    torch.manual_seed(CFG["seed"])
    ref = net.BSRNN(input_dim=CFG["input_dim"], num_channel=CFG["num_channel"], num_layer=CFG["num_layer"],
                    target_fs=48000, causal=False)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n or ("mlp_" in n and ".0." in n):
                p.add_(0.1 * torch.randn_like(p))
    mine = flow_ref.BSRNNFlow(CFG["input_dim"], CFG["num_channel"], CFG["num_layer"])
    mine.load_state_dict(ref.state_dict(), strict=True)
    g = torch.Generator().manual_seed(CFG["seed"] + 1)
    B, Fb, T = CFG["B"], CFG["input_dim"], CFG["T"]
    x = torch.randn(B, 1, Fb, T, dtype=torch.complex64, generator=g) * 0.3
    y = torch.randn(B, 1, Fb, T, dtype=torch.complex64, generator=g) * 0.3
    t = torch.tensor([0.8, 0.31])
    with torch.no_grad():
        out_ref = ref(torch.cat([x, y], 1), t, fs=48000)
        out_mine = mine(torch.cat([x, y], 1), t)
    assert torch.equal(out_ref, out_mine), (out_ref - out_mine).abs().max()

    # flow-matching ODE + white-box Euler solver of the reference, driven by the reference network
    ode = odes.FLOWMATCHING(sigma_min=0.05, sigma_max=0.5)
    vf = lambda xx, tt, yy: -ref(torch.cat([xx, yy], dim=1), tt, fs=48000)
    torch.manual_seed(5)
    solver = sampling.get_white_box_solver("euler", ode, vf, y, T_rev=1.0, t_eps=0.03, N=4)
    torch.manual_seed(6)
    z_used = torch.randn_like(y)          # prior_sampling draws randn_like(y) first thing
    torch.manual_seed(6)
    sample_ref, ns = solver()
    ode_m = flow_ref.FlowMatching(0.05, 0.5)
    sample_mine = flow_ref.euler_sample(lambda xx, tt, yy: -mine(torch.cat([xx, yy], 1), tt), ode_m, y, z_used, 1.0, 0.03, 4)
    assert ns == 4 and torch.allclose(sample_ref, sample_mine, atol=1e-6), (sample_ref - sample_mine).abs().max()
    mean_r, std_r = ode.marginal_prob(x, t, y)
    mean_m, std_m = ode_m.marginal_prob(x, t, y)
    assert torch.equal(mean_r, mean_m) and torch.equal(std_r, std_m)
    assert ode.der_std(t) == ode_m.der_std(t) and torch.equal(ode.der_mean(x, t, y), ode_m.der_mean(x, t, y))

    # parameter count of the full-size flow DNN (103,245,488)
    full = net.BSRNN(input_dim=769, num_channel=384, num_layer=6, target_fs=48000, causal=False)
    n_full = sum(p.numel() for p in full.parameters())
    assert n_full == 103245488, n_full

    sd = ref.state_dict()
    np.savez_compressed(
        os.path.join(HERE, "ref_flow.npz"), x=torch.view_as_real(x).numpy(), y=torch.view_as_real(y).numpy(), t=t.numpy(),
        out=torch.view_as_real(out_ref).numpy(), z=torch.view_as_real(z_used).numpy(),
        sample=torch.view_as_real(sample_ref).numpy(), mean=torch.view_as_real(mean_r).numpy(), std=std_r.numpy(),
        n_params_full=np.array(n_full), keys=np.array(list(sd.keys())),
        **{"w:" + k: v.numpy() for k, v in sd.items()})
    print("ref_flow.npz written; reference == oracle (bitwise forward), Euler trajectory matches")


if __name__ == "__main__":
    main()
