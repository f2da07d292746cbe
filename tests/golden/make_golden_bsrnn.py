"""Pin of the DISCRIMINATIVE oracle (oracle/bsrnn_ref.py) to reference-held code (run in the build container only).

espnet2's ``BSRNNSeparator`` is not installed, but the reference carries its structural twin in-tree:
``baseline_code/models/bsrnn_flowse.py:16-86`` (``BandSplit`` with the 481-bin table of the discriminative model) and
``:288-307`` (the dual-path loop).  This script imports that file (espnet shim of make_golden_flow.py: only
``choose_norm`` / ``choose_norm1d`` -> ``nn.GroupNorm(1, C)`` get behaviour), and

1. runs the reference ``BandSplit(481, channels=N)`` on spectra of 481 / 221 / 161 / 81 bins (48 / 22.05 / 16 / 8 kHz,
   with and without the ``fs`` argument) and asserts ``oracle.bsrnn_ref.BandSplit`` BIT-EQUAL;
2. runs the reference ``BSRNN(input_dim=481)`` end to end, captures the tensor entering the loop (output of
   ``condition_fc``) and the tensor leaving it (input of ``grad_decoder``) with forward hooks, and asserts
   ``oracle.bsrnn_ref.BSRNN.dual_path`` equal on the same weights:
   a. with the time embedding modules replaced by a zero embedding (``out + 0`` is exact): BIT-EQUAL;
   b. with the reference's real ``GaussianFourierProjection`` and one ``t`` for the whole batch, folded into the
      oracle's GroupNorm bias (``beta + t_emb``: GN(x) + t_emb == GN_{beta + t_emb}(x) up to one rounding): <= 2e-6.

Inputs, weights and the reference's outputs go to tests/golden/ref_bsrnn.npz (data only, no reference source); the CPU
test tests/test_oracle.py::test_bsrnn_oracle_equals_reference_twin and the GPU test
tests/test_bsrnn_gpu.py::test_dual_path_and_band_split_match_reference_twin_vectors read it.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden_flow import load_reference  # noqa: E402
from oracle import bsrnn_ref  # noqa: E402

N, L, B, T = 16, 2, 2, 11
LOOP_MODULES = ("norm_time", "rnn_time", "fc_time", "norm_freq", "rnn_freq", "fc_freq")


class _ZeroEmbedding(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, t):
        return torch.zeros(t.shape[0], self.n)


def _loop_state(ref):
    return {k: v for k, v in ref.state_dict().items() if k.split(".")[0] in LOOP_MODULES}


def _drive(ref, dnn_input, t):
    """reference BSRNN.forward with hooks on the loop's entry and exit."""
    grabbed = {}
    h1 = ref.condition_fc.register_forward_hook(lambda m, i, o: grabbed.__setitem__("z", o.permute(0, 3, 1, 2).clone()))
    h2 = ref.grad_decoder.register_forward_pre_hook(lambda m, i: grabbed.__setitem__("skip", i[0].clone()))
    with torch.no_grad():
        ref(dnn_input, t, fs=48000)
    h1.remove()
    h2.remove()
    return grabbed["z"], grabbed["skip"]


def main():
    net, _, _ = load_reference()
    out = {}

    # ---- 1. BandSplit -------------------------------------------------------------------------------------------
    torch.manual_seed(311)
    ref_bs = net.BandSplit(481, target_fs=48000, channels=N)
    with torch.no_grad():
        for p in ref_bs.norm.parameters():
            p.add_(0.1 * torch.randn_like(p))
    mine_bs = bsrnn_ref.BandSplit(481, 48000, N)
    mine_bs.load_state_dict(ref_bs.state_dict(), strict=True)
    g = torch.Generator().manual_seed(312)
    for fs, F in ((48000, 481), (22050, 221), (16000, 161), (8000, 81)):
        x = 0.5 * torch.randn(B, 7, F, 2, generator=g)
        with torch.no_grad():
            z_none = ref_bs(x, fs=None)
            z_fs = ref_bs(x, fs=fs)
            z_mine = mine_bs(x)
        # 22.05 kHz: the fs rule of the reference stops one band earlier than the bin-count rule (subband_freqs[i] >= fs / 2
        # at a band edge below the last bin); models/bsrnn.py never passes fs to the separator, so fs=None is THE path
        assert torch.equal(z_none, z_mine), (fs, (z_none - z_mine).abs().max())
        assert z_fs.shape[-1] <= z_none.shape[-1] and torch.equal(z_fs, z_none[..., :z_fs.shape[-1]])
        assert z_mine.shape[-1] == bsrnn_ref.num_bands_for(F, bsrnn_ref.SUBBANDS_481)
        out["bs_x_%d" % fs] = x.numpy()
        out["bs_z_%d" % fs] = z_none.numpy()
        out["bs_kfs_%d" % fs] = np.array(z_fs.shape[-1])
    for k, v in ref_bs.state_dict().items():
        out["bsw:" + k] = v.numpy()

    # ---- 2. dual-path loop ------------------------------------------------------------------------------------------
    torch.manual_seed(313)
    ref = net.BSRNN(input_dim=481, num_channel=N, num_layer=L, target_fs=48000, causal=False)
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if n_.split(".")[0] in ("norm_time", "norm_freq"):
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(314)
    x = 0.3 * torch.randn(B, 1, 481, T, dtype=torch.complex64, generator=g)
    y = 0.3 * torch.randn(B, 1, 481, T, dtype=torch.complex64, generator=g)
    dnn_input = torch.cat([x, y], 1)
    t = torch.full((B,), 0.37)

    mine = bsrnn_ref.BSRNN(481, N, L, 48000, False, 1)
    mine.load_state_dict(_loop_state(ref), strict=False)

    # a. zero time embedding: the loop alone, bit for bit
    real_tcond = ref.t_cond
    ref.t_cond = nn.ModuleList([_ZeroEmbedding(N) for _ in range(L)])
    z, skip0 = _drive(ref, dnn_input, t)
    with torch.no_grad():
        skip0_mine = mine.dual_path(z)
    assert torch.equal(skip0, skip0_mine), (skip0 - skip0_mine).abs().max()

    # b. the reference's own embedding, folded into the oracle's norm bias
    ref.t_cond = real_tcond
    z_b, skip_t = _drive(ref, dnn_input, t)
    assert torch.equal(z, z_b)
    folded = bsrnn_ref.BSRNN(481, N, L, 48000, False, 1)
    folded.load_state_dict(_loop_state(ref), strict=False)
    with torch.no_grad():
        for i in range(L):
            folded.norm_time[i].bias.add_(ref.t_cond[i](t)[0])
        skip_t_mine = folded.dual_path(z)
    err = (skip_t - skip_t_mine).abs().max().item() / skip_t.abs().max().item()
    assert err <= 2e-6, err

    out.update(z=z.numpy(), skip_zero_temb=skip0.numpy(), skip_folded_temb=skip_t.numpy())
    for k, v in _loop_state(ref).items():
        out["w:" + k] = v.numpy()
    for k, v in _loop_state(folded).items():
        if k.startswith("norm_time") and k.endswith("bias"):
            out["wfold:" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "ref_bsrnn.npz"), **out)
    print("ref_bsrnn.npz written: BandSplit bit-equal at 4 rates; dual-path loop bit-equal (zero t_emb), %.1e (folded t_emb)" % err)


if __name__ == "__main__":
    main()
