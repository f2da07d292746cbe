"""Oracle: BSRNN-Flow DNN, flow-matching ODE, Euler sampler and the FlowSEModel step on CPU torch.

TEST INFRASTRUCTURE - never imported by the product path.

Restates ``baseline_code/models/bsrnn_flowse.py`` (``BandSplit`` :16-86, ``GaussianFourierProjection`` :90-99,
``GradDecoder`` :103-168, ``BSRNN`` :171-318), ``baseline_code/models/odes.py:52-98`` (``FLOWMATCHING``),
``baseline_code/sampling/__init__.py:30-65`` + ``sampling/odesolvers.py:72-81`` (white-box Euler solver) and
``baseline_code/flow_model.py`` (``speech_to_feature`` :134-139, ``forward_step`` :149-187, ``_loss`` :122-132,
``enhance`` :189-200, ``forward`` :203-209, EMA via torch_ema :53,84).

PINNED: ``tests/golden/make_golden_flow.py`` imports the reference's OWN ``bsrnn_flowse.py`` (espnet shim of SURVEY
8c), ``odes.py`` and ``sampling`` in the build container, loads identical weights into this restatement and asserts
equality before writing ``tests/golden/ref_flow.npz``; tests/test_oracle.py re-checks this file against it.
torch_ema (absent) is restated from its published update rule: unpinned.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import stft_ref
from .bsrnn_ref import SUBBANDS_481, SUBBANDS_769, num_bands_for


class BandSplit(nn.Module):
    def __init__(self, input_dim, channels):
        super().__init__()
        self.subbands = SUBBANDS_481 if input_dim == 481 else SUBBANDS_769
        assert sum(self.subbands) == input_dim
        self.norm = nn.ModuleList([nn.GroupNorm(1, 2 * sb) for sb in self.subbands])
        self.fc = nn.ModuleList([nn.Conv1d(2 * sb, channels, 1) for sb in self.subbands])

    def forward(self, x):
        outs, hz = [], 0
        for i, sb in enumerate(self.subbands):
            xb = x[:, :, hz:hz + sb, :]
            if sb > xb.size(2):
                xb = F.pad(xb, (0, 0, 0, sb - xb.size(2)))
            xb = xb.reshape(xb.size(0), xb.size(1), -1)
            outs.append(self.fc[i](self.norm[i](xb.transpose(1, 2))).unsqueeze(-1))
            hz += sb
            if hz >= x.size(2):
                break
        return torch.cat(outs, dim=-1)


class GaussianFourierProjection(nn.Module):
    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, x):
        x_proj = x[:, None] * self.W[None, :] * 2 * torch.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


class GradDecoder(nn.Module):
    def __init__(self, freq_dim, subbands, channels, sub_channel=16):
        super().__init__()
        self.subbands, self.freq_dim, self.sub_channel = subbands, freq_dim, sub_channel
        self.conv_after_mask = nn.Sequential(nn.Conv2d(sub_channel, 4, 5, 1, 2), nn.GLU(dim=1))
        self.conv_after_residual = nn.Sequential(nn.Conv2d(sub_channel, 4, 5, 1, 2), nn.GLU(dim=1))
        mk = lambda sb: nn.Sequential(nn.GroupNorm(1, channels), nn.Conv1d(channels, sb * sub_channel, 1), nn.Tanh())
        self.mlp_mask = nn.ModuleList([mk(sb) for sb in subbands])
        self.mlp_residual = nn.ModuleList([mk(sb) for sb in subbands])

    def forward(self, x):
        B, N, T, K = x.shape
        ms, rs = [], []
        for i, sb in enumerate(self.subbands):
            if i >= K:
                break
            xb = x[:, :, :, i]
            ms.append(self.mlp_mask[i](xb).view(B, self.sub_channel, sb, T))
            rs.append(self.mlp_residual[i](xb).view(B, self.sub_channel, sb, T))
        m = self.conv_after_mask(torch.cat(ms, dim=2))
        r = self.conv_after_residual(torch.cat(rs, dim=2))
        m = F.pad(m, (0, 0, 0, int(self.freq_dim - m.size(-2))))
        r = F.pad(r, (0, 0, 0, int(self.freq_dim - r.size(-2))))
        return m.moveaxis(1, 3).contiguous(), r.moveaxis(1, 3).contiguous()


class BSRNNFlow(nn.Module):
    """bsrnn_flowse.BSRNN(input_dim, num_channel, num_layer, target_fs=48000, causal=False)."""

    def __init__(self, input_dim=769, num_channel=384, num_layer=6):
        super().__init__()
        N, hd = num_channel, 2 * num_channel
        self.num_layer = num_layer
        self.band_split_y = BandSplit(input_dim, N)
        self.band_split_x = BandSplit(input_dim, N)
        self.condition_fc = nn.Linear(2 * N, N)
        self.norm_time, self.rnn_time, self.fc_time = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.norm_freq, self.rnn_freq, self.fc_freq = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.t_cond = nn.ModuleList()
        for _ in range(num_layer):
            self.t_cond.append(GaussianFourierProjection(N // 2, scale=1))
            self.norm_time.append(nn.GroupNorm(1, N))
            self.rnn_time.append(nn.LSTM(N, hd, batch_first=True, bidirectional=True))
            self.fc_time.append(nn.Linear(2 * hd, N))
            self.norm_freq.append(nn.GroupNorm(1, N))
            self.rnn_freq.append(nn.LSTM(N, hd, batch_first=True, bidirectional=True))
            self.fc_freq.append(nn.Linear(4 * N, N))
        self.grad_decoder = GradDecoder(input_dim, self.band_split_x.subbands, N)

    def forward(self, dnn_input, t):
        x = dnn_input[:, 0].permute(0, 2, 1)
        y = dnn_input[:, 1].permute(0, 2, 1)
        x = torch.stack([x.real, x.imag], dim=-1)
        y = torch.stack([y.real, y.imag], dim=-1)
        xx, yy = self.band_split_x(x), self.band_split_y(y)
        zz = torch.cat([xx, yy], dim=1).permute(0, 2, 3, 1)
        z = self.condition_fc(zz).permute(0, 3, 1, 2)
        B, N, T, K = z.shape
        skip = z
        for i in range(self.num_layer):
            out = self.norm_time[i](skip)
            out = out + self.t_cond[i](t)[..., None, None]
            out = out.transpose(1, 3).reshape(B * K, T, N)
            out, _ = self.rnn_time[i](out)
            out = self.fc_time[i](out).reshape(B, K, T, N).transpose(1, 3)
            skip = skip + out
            out = self.norm_freq[i](skip)
            out = out.permute(0, 2, 3, 1).contiguous().reshape(B * T, K, N)
            out, _ = self.rnn_freq[i](out)
            out = self.fc_freq[i](out).reshape(B, T, K, N).permute(0, 3, 1, 2).contiguous()
            skip = skip + out
        m, r = self.grad_decoder(skip)
        x_t = dnn_input[:, 0]
        Fb = x_t.shape[1]
        m = torch.view_as_complex(m)[:, 0:Fb, :]
        r = torch.view_as_complex(r)[:, 0:Fb, :]
        return (m * x_t + r).unsqueeze(1)


class FlowMatching:
    """odes.py:52-98."""

    def __init__(self, sigma_min=0.0, sigma_max=0.5):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def _std(self, t):
        return (1 - t) * self.sigma_min + t * self.sigma_max

    def marginal_prob(self, x0, t, y):
        return (1 - t)[:, None, None, None] * x0 + t[:, None, None, None] * y, self._std(t)

    def prior_sampling(self, y, z):
        std = self._std(torch.ones((y.shape[0],)))
        return y + z * std[:, None, None, None]

    def der_mean(self, x0, t, y):
        return y - x0

    def der_std(self, t):
        return self.sigma_max - self.sigma_min


def euler_timesteps(T_rev, t_eps, N):
    ts = torch.linspace(T_rev, t_eps, N)
    steps = [float(ts[i] - ts[i + 1]) if i != N - 1 else float(ts[-1]) for i in range(N)]
    return ts, steps


def euler_sample(vf_fn, ode, Y, z, T_rev=1.0, t_eps=0.03, N=15):
    """sampling/__init__.py:30-65 with 'euler' (odesolvers.py:72-81): x <- x - step * VF(x, t, Y)."""
    xt = ode.prior_sampling(Y, z)
    ts, steps = euler_timesteps(T_rev, t_eps, N)
    for i in range(N):
        vec_t = torch.ones(Y.shape[0]) * ts[i]
        xt = xt + vf_fn(xt, vec_t, Y) * (-steps[i])
    return xt


class FlowSE(nn.Module):
    """FlowSEModel arithmetic (flow_model.py) with explicit noise / time inputs so that it is reproducible."""

    def __init__(self, n_fft=1536, hop_length=384, bsrnn_hidden=384, num_layer=6, sigma_min=0.05, sigma_max=0.5,
                 t_eps=0.03, T_rev=1.0, spec_abs_exponent=0.667, spec_factor=0.065):
        super().__init__()
        self.n_fft, self.hop, self.e, self.factor = n_fft, hop_length, spec_abs_exponent, spec_factor
        self.ode = FlowMatching(sigma_min, sigma_max)
        self.t_eps, self.T_rev = t_eps, T_rev
        self.dnn = BSRNNFlow(n_fft // 2 + 1, bsrnn_hidden, num_layer)

    def speech_to_feature(self, speech, fs, lens):
        n_fft, hop = stft_ref.reconfig_for_fs(self.n_fft, self.hop, fs, 48000)
        X, _ = stft_ref.stft(speech, n_fft, hop, "hann", lens)
        X = stft_ref.spec_transform(X, "exponent", self.factor, self.e)
        return X.permute(0, 2, 1).unsqueeze(1)                      # [B,1,F,T]

    def feature_to_speech(self, feat, fs, lens):
        n_fft, hop = stft_ref.reconfig_for_fs(self.n_fft, self.hop, fs, 48000)
        X = stft_ref.spec_back(feat.squeeze(1).permute(0, 2, 1), "exponent", self.factor, self.e)
        return stft_ref.istft(X, n_fft, hop, int(torch.as_tensor(lens).max()))

    def forward(self, x, t, y):
        return -self.dnn(torch.cat([x, y], dim=1), t)

    def loss_from(self, x0, y, t, z):
        """forward_step :159-172 given the features, the time draw t and the complex noise z."""
        mean, std = self.ode.marginal_prob(x0, t, y)
        xt = mean + std[:, None, None, None] * z
        condVF = self.ode.der_std(t) * z + self.ode.der_mean(x0, t, y)
        vf = self(xt, t, y)
        err = vf - condVF
        return torch.mean(0.5 * torch.sum(torch.square(err.abs()).reshape(err.shape[0], -1), dim=-1))

    def enhance_from(self, Y, z, N=15):
        with torch.no_grad():
            return euler_sample(self.forward, self.ode, Y, z, self.T_rev, self.t_eps, N)


def ema_update(shadow, params, decay, num_updates):
    """torch_ema.ExponentialMovingAverage.update (use_num_updates=True): returns the new num_updates."""
    num_updates += 1
    d = min(decay, (1 + num_updates) / (10 + num_updates))
    with torch.no_grad():
        for s, p in zip(shadow, params):
            s.sub_((1.0 - d) * (s - p))
    return num_updates
