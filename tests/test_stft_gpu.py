"""GPU parity: framed STFT / iSTFT kernels vs the CPU oracle (oracle/stft_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import stft_ref

pytestmark = pytest.mark.gpu

CASES = [  # (n_fft, hop, L, window)
    (960, 480, 24000, "hann"),     # 48 kHz model STFT
    (320, 160, 16000, "hann"),     # 16 kHz
    (441, 220, 11025, "hann"),     # 22.05 kHz (3^2 * 7^2, generic radix)
    (1536, 384, 19200, "hann"),    # flow model
    (256, 128, 5000, "rect"), (512, 256, 5000, "rect"), (768, 384, 5000, "rect"), (1024, 512, 5000, "rect"),
    (160, 80, 4001, "hann"), (640, 320, 9999, "hann"), (882, 441, 22050, "hann"), (480, 240, 12000, "hann"),
    (960, 480, 7700, "rect"),      # the register-FFT kernels' rectangular-window branch, a length that ends inside a workgroup's chunk
]


@pytest.mark.parametrize("n_fft,hop,L,window", CASES)
def test_stft_matches_oracle(lib, n_fft, hop, L, window):
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator().manual_seed(n_fft + L)
    x = torch.randn(3, L, generator=g)
    lens = torch.tensor([L, L - hop * 3 - 7, L // 2])
    ref, _ = stft_ref.stft(x.double(), n_fft, hop, window, lens)
    got = ops.stft_forward(x.cuda(), n_fft, hop, ops.WIN_HANN if window == "hann" else ops.WIN_RECT, lens).cpu()
    assert got.shape == ref.shape
    scale = ref.abs().max().item()
    err = (got.to(torch.complex128) - ref).abs().max().item()
    assert err <= 2e-6 * scale + 1e-6, (err, scale)   # f32 FFT of length <= 1536: ~1e-7 relative per bin
    # masked frames are exactly zero
    olens = (lens + 2 * (n_fft // 2) - n_fft) // hop + 1
    for b in range(3):
        assert torch.all(got[b, olens[b]:] == 0)


@pytest.mark.parametrize("n_fft,hop,L,window", CASES)
def test_istft_matches_oracle_and_roundtrip(lib, n_fft, hop, L, window):
    from urgent2026_challenge_track1_amd import ops
    if window == "rect" and n_fft == 768:
        pass
    g = torch.Generator().manual_seed(n_fft * 3 + L)
    T, Fb = L // hop + 1, n_fft // 2 + 1
    X = torch.randn(2, T, Fb, dtype=torch.complex64, generator=g)
    ref = stft_ref.istft(X.to(torch.complex128), n_fft, hop, L, window)
    w = ops.WIN_HANN if window == "hann" else ops.WIN_RECT
    got = ops.istft_forward(X.cuda(), n_fft, hop, L, w).cpu()
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() <= 5e-6 * scale
    # round trip: istft(stft(x)) == x
    x = torch.randn(2, L, generator=g)
    y = ops.istft_forward(ops.stft_forward(x.cuda(), n_fft, hop, w), n_fft, hop, L, w).cpu()
    assert (y - x).abs().max().item() <= 2e-5


@pytest.mark.parametrize("n_fft,hop,L,window", CASES[:5] + CASES[-1:])
def test_istft_backward_is_adjoint(lib, n_fft, hop, L, window):
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator().manual_seed(7)
    T, Fb = L // hop + 1, n_fft // 2 + 1
    X = torch.randn(2, T, Fb, dtype=torch.complex128, generator=g, requires_grad=True)
    gw = torch.randn(2, L, generator=g, dtype=torch.float64)
    stft_ref.istft(X, n_fft, hop, L, window).backward(gw)
    ref = X.grad
    w = ops.WIN_HANN if window == "hann" else ops.WIN_RECT
    Xg = torch.view_as_real(X.detach().to(torch.complex64)).cuda().requires_grad_(True)
    ops.istft_forward(Xg, n_fft, hop, L, w).backward(gw.float().cuda())
    got = torch.view_as_complex(Xg.grad.cpu())
    scale = ref.abs().max().item()
    assert (got.to(torch.complex128) - ref).abs().max().item() <= 5e-6 * scale


def test_stft_linearity_at_full_size(lib):
    """size-independent property at the BASELINE C2 shape (B32 x 4 s @ 48 kHz)."""
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(32, 192000, generator=g).cuda()
    y = torch.randn(32, 192000, generator=g).cuda()
    ops.launch_counts(reset=True)
    a = ops.stft_forward(x, 960, 480)
    b = ops.stft_forward(y, 960, 480)
    c = ops.stft_forward(2.0 * x - 3.0 * y, 960, 480)
    assert a.shape == (32, 401, 481)
    assert (c - (2.0 * a - 3.0 * b)).abs().max().item() <= 1e-3 * c.abs().max().item()
    # Parseval on an interior frame-free quantity: round trip at full size
    r = ops.istft_forward(a, 960, 480, 192000)
    assert (r - x).abs().max().item() <= 5e-5
    n = ops.launch_counts()
    assert n["stft960"] == 3 and n["istft960"] == 1 and n["stft_generic"] == 0 and n["istft_generic"] == 0, n   # the register-FFT kernels ran
