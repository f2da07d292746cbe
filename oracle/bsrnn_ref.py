"""Oracle: ``BSRNN_SE`` (STFTEncoder -> BSRNNSeparator -> STFTDecoder) on CPU torch.

TEST INFRASTRUCTURE - never imported by the product path.

Follows ``baseline_code/models/bsrnn.py:9-41`` (wrapper), and for the separator
the espnet2 ``BSRNNSeparator`` / ``BSRNN`` / ``BandSplit`` / ``MaskDecoder``
restated in SURVEY A.2; the dual-path loop and BandSplit are the same code as
the in-tree twin ``baseline_code/models/bsrnn_flowse.py:16-86,288-307`` (minus
the ``t_emb`` lines 293-294).  Parameter/attribute names match espnet so that
``state_dict`` keys are ``bsrnn.bsrnn.band_split.norm.{i}.weight`` etc.
Architecture pin: parameter counts of ``conf/models/BSRNN_baseline.yaml:30-31``
(checked in tests/test_oracle.py).  Numerical pin (round 3): ``BandSplit`` and ``BSRNN.dual_path``
are asserted BIT-EQUAL to the reference's own ``bsrnn_flowse.BandSplit`` / ``BSRNN`` loop run in the
build container (tests/golden/make_golden_bsrnn.py -> ref_bsrnn.npz).  ``MaskDecoder`` and the
``m*x + r`` tail restate espnet2 (absent; the in-tree twin's GradDecoder is a different head): that
part stays unpinned against espnet itself.

``emulate_bf16=True`` restates the rounding points of the bf16 MFMA product
path (operands of every dense contraction rounded to bf16, f32 accumulate, gate
pre-activations of the input projection stored in bf16, hidden state carried in
bf16) so that GPU bf16 results can be checked tightly; the f32 mode is the
reference arithmetic.
"""
from itertools import accumulate

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import stft_ref

SUBBANDS_481 = tuple([5] + [4] * 19 + [10] * 6 + [40] * 7 + [60])      # bsrnn_flowse.py:29
SUBBANDS_769 = tuple([5] + [4] * 26 + [10] * 10 + [50] * 10 + [60])    # bsrnn_flowse.py:36


def _r(x, on):
    """round-trip through bf16 when emulating the MFMA path."""
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def num_bands_for(F_bins, subbands):
    """K rule of BandSplit.forward with fs=None (bsrnn_flowse.py:63-86): stop once hz_band >= F."""
    hz = 0
    for i, sb in enumerate(subbands):
        hz += sb
        if hz >= F_bins:
            return i + 1
    return len(subbands)


class BandSplit(nn.Module):
    def __init__(self, input_dim=481, target_fs=48000, channels=128):
        super().__init__()
        if input_dim == 481 and target_fs == 48000:
            self.subbands = SUBBANDS_481
        elif input_dim == 769 and target_fs == 48000:
            self.subbands = SUBBANDS_769
        else:
            raise NotImplementedError
        self.norm = nn.ModuleList([nn.GroupNorm(1, 2 * sb) for sb in self.subbands])
        self.fc = nn.ModuleList([nn.Conv1d(2 * sb, channels, 1) for sb in self.subbands])

    def forward(self, x, emulate_bf16=False):
        # x: [B, T, F, 2] -> [B, N, T, K]
        outs = []
        hz = 0
        for i, sb in enumerate(self.subbands):
            xb = x[:, :, hz:hz + sb, :]
            if sb > xb.size(2):
                xb = F.pad(xb, (0, 0, 0, sb - xb.size(2)))
            xb = xb.reshape(xb.size(0), xb.size(1), -1)
            o = self.norm[i](xb.transpose(1, 2))
            w, b = self.fc[i].weight, self.fc[i].bias
            o = F.conv1d(_r(o, emulate_bf16), _r(w, emulate_bf16), b)
            outs.append(o.unsqueeze(-1))
            hz += sb
            if hz >= x.size(2):
                break
        return torch.cat(outs, dim=-1)


class MaskDecoder(nn.Module):
    def __init__(self, freq_dim, subbands, channels=128, num_spk=1):
        super().__init__()
        self.subbands, self.freq_dim, self.num_spk = subbands, freq_dim, num_spk
        mk = lambda sb: nn.Sequential(nn.GroupNorm(1, channels), nn.Conv1d(channels, 4 * channels, 1), nn.Tanh(),
                                      nn.Conv1d(4 * channels, int(sb * 4 * num_spk), 1), nn.GLU(dim=1))
        self.mlp_mask = nn.ModuleList([mk(sb) for sb in subbands])
        self.mlp_residual = nn.ModuleList([mk(sb) for sb in subbands])

    @staticmethod
    def _mlp(seq, xb, e):
        o = seq[0](xb)
        o = torch.tanh(F.conv1d(_r(o, e), _r(seq[1].weight, e), seq[1].bias))
        o = F.conv1d(_r(o, e), _r(seq[3].weight, e), seq[3].bias)
        return F.glu(o, dim=1)

    def forward(self, x, emulate_bf16=False):
        ms, rs = [], []
        for i in range(len(self.subbands)):
            if i >= x.size(-1):
                break
            xb = x[:, :, :, i]
            o = self._mlp(self.mlp_mask[i], xb, emulate_bf16).transpose(1, 2).contiguous()
            ms.append(o.reshape(o.size(0), o.size(1), self.num_spk, -1, 2))
            o = self._mlp(self.mlp_residual[i], xb, emulate_bf16).transpose(1, 2).contiguous()
            rs.append(o.reshape(o.size(0), o.size(1), self.num_spk, -1, 2))
        m = torch.cat(ms, dim=3)
        r = torch.cat(rs, dim=3)
        m = F.pad(m, (0, 0, 0, int(self.freq_dim - m.size(-2))))
        r = F.pad(r, (0, 0, 0, int(self.freq_dim - r.size(-2))))
        return m.moveaxis(1, 2), r.moveaxis(1, 2)


def lstm_bidir(lstm, x, emulate_bf16=False):
    """nn.LSTM(batch_first, bidirectional) forward; manual loop when emulating bf16."""
    if not emulate_bf16:
        return lstm(x)[0]
    S, T, _ = x.shape
    H = lstm.hidden_size
    outs = []
    for sfx, rev in (("", False), ("_reverse", True)):
        wih = getattr(lstm, "weight_ih_l0" + sfx)
        whh = getattr(lstm, "weight_hh_l0" + sfx)
        b = getattr(lstm, "bias_ih_l0" + sfx) + getattr(lstm, "bias_hh_l0" + sfx)
        gx = _r(F.linear(_r(x, True), _r(wih, True), b), True)       # [S, T, 4H] stored bf16
        h = x.new_zeros(S, H)
        c = x.new_zeros(S, H)
        hs = [None] * T
        whh_r = _r(whh, True)
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            g = gx[:, t] + F.linear(h, whh_r)
            i_, f_, g_, o_ = g.chunk(4, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            h = _r(torch.sigmoid(o_) * torch.tanh(c), True)
            hs[t] = h
        outs.append(torch.stack(hs, dim=1))
    return torch.cat(outs, dim=-1)


class BSRNN(nn.Module):
    def __init__(self, input_dim=481, num_channel=16, num_layer=6, target_fs=48000, causal=True, num_spk=1):
        super().__init__()
        assert not causal and num_spk == 1
        self.num_layer = num_layer
        self.band_split = BandSplit(input_dim, target_fs=target_fs, channels=num_channel)
        N, hd = num_channel, 2 * num_channel
        self.norm_time = nn.ModuleList([nn.GroupNorm(1, N) for _ in range(num_layer)])
        self.rnn_time = nn.ModuleList([nn.LSTM(N, hd, batch_first=True, bidirectional=True) for _ in range(num_layer)])
        self.fc_time = nn.ModuleList([nn.Linear(2 * hd, N) for _ in range(num_layer)])
        self.norm_freq = nn.ModuleList([nn.GroupNorm(1, N) for _ in range(num_layer)])
        self.rnn_freq = nn.ModuleList([nn.LSTM(N, hd, batch_first=True, bidirectional=True) for _ in range(num_layer)])
        self.fc_freq = nn.ModuleList([nn.Linear(2 * hd, N) for _ in range(num_layer)])
        self.mask_decoder = MaskDecoder(input_dim, self.band_split.subbands, channels=N, num_spk=num_spk)

    def dual_path(self, z, emulate_bf16=False):
        """the L x {time path, band path} loop on z [B, N, T, K] (bsrnn_flowse.py:288-307 without the t_emb lines 293-294);
        pinned to the reference's own loop by tests/golden/make_golden_bsrnn.py."""
        e = emulate_bf16
        B, N, T, K = z.shape
        skip = z
        for i in range(self.num_layer):
            out = self.norm_time[i](skip)
            out = out.transpose(1, 3).reshape(B * K, T, N)
            out = lstm_bidir(self.rnn_time[i], out, e)
            out = F.linear(_r(out, e), _r(self.fc_time[i].weight, e), self.fc_time[i].bias)
            out = out.reshape(B, K, T, N).transpose(1, 3)
            skip = skip + out
            out = self.norm_freq[i](skip)
            out = out.permute(0, 2, 3, 1).contiguous().reshape(B * T, K, N)
            out = lstm_bidir(self.rnn_freq[i], out, e)
            out = F.linear(_r(out, e), _r(self.fc_freq[i].weight, e), self.fc_freq[i].bias)
            out = out.reshape(B, T, K, N).permute(0, 3, 1, 2).contiguous()
            skip = skip + out
        return skip

    def forward(self, x, emulate_bf16=False):
        e = emulate_bf16
        skip = self.dual_path(self.band_split(x, e), e)
        m, r = self.mask_decoder(skip, e)
        m = torch.view_as_complex(m.contiguous())
        r = torch.view_as_complex(r.contiguous())
        xc = torch.view_as_complex(x.contiguous())
        m = m[..., :xc.size(-1)]
        r = r[..., :xc.size(-1)]
        return torch.view_as_real(m * xc.unsqueeze(1) + r)


class BSRNNSeparator(nn.Module):
    def __init__(self, input_dim, num_spk=1, num_channels=16, num_layers=6, target_fs=48000, causal=True):
        super().__init__()
        self.bsrnn = BSRNN(input_dim, num_channels, num_layers, target_fs, causal, num_spk)

    def forward(self, spec, emulate_bf16=False):
        feature = torch.stack([spec.real, spec.imag], dim=-1)
        masked = self.bsrnn(feature, emulate_bf16)            # [B, 1, T, F, 2]
        return torch.complex(masked[..., 0], masked[..., 1])[:, 0]


class BSRNN_SE(nn.Module):
    """baseline_code/models/bsrnn.py:9-41."""

    def __init__(self, num_channel=192, num_layer=6):
        super().__init__()
        self.n_fft, self.hop, self.default_fs = 960, 480, 48000
        self.bsrnn = BSRNNSeparator(self.n_fft // 2 + 1, 1, num_channel, num_layer, 48000, False)

    def forward(self, speech_mix, speech_lengths, fs, emulate_bf16=False):
        n_fft, hop = stft_ref.reconfig_for_fs(self.n_fft, self.hop, fs, self.default_fs)
        spec, _ = stft_ref.stft(speech_mix, n_fft, hop, "hann", speech_lengths)
        enh = self.bsrnn(spec, emulate_bf16)
        wav = stft_ref.istft(enh, n_fft, hop, int(torch.as_tensor(speech_lengths).max()))
        return wav, enh
