"""GPU parity: GroupNorm forward / backward kernels against torch autograd on the same normalisation
(statistics over (T, W) per (batch, group), per-channel affine repeating every N columns)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, gamma, beta, B, T, Kg, W, N, gstride, eps):
    xr = x.double().reshape(B, T, Kg, W)
    mean = xr.mean(dim=(1, 3), keepdim=True)
    var = xr.var(dim=(1, 3), unbiased=False, keepdim=True)
    xh = (xr - mean) / torch.sqrt(var + eps)
    g = torch.stack([gamma[k * gstride:k * gstride + N] for k in range(Kg)]).double()      # [Kg, N]
    b = torch.stack([beta[k * gstride:k * gstride + N] for k in range(Kg)]).double()
    y = xh.reshape(B, T, Kg, W // N, N) * g[None, None, :, None, :] + b[None, None, :, None, :]
    return y.reshape(B, T, Kg, W)


@pytest.mark.parametrize("B,T,Kg,W,N,gstride", [(2, 37, 1, 5 * 196, 196, 0), (3, 21, 4, 196, 196, 196), (2, 19, 3, 2 * 20, 20, 20),
                                                 (2, 11, 2, 3 * 12, 12, 12)])
def test_groupnorm_fwd_bwd_matches_autograd(lib, B, T, Kg, W, N, gstride):
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, T, Kg, W, generator=g)
    ng = max(1, Kg if gstride else 1) * max(N, gstride)
    gamma = (1 + 0.3 * torch.randn(ng, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(ng, generator=g)).requires_grad_()
    dy = torch.randn(B, T, Kg, W, generator=g)
    xr = x.clone().requires_grad_()
    y_ref = _ref(xr, gamma, beta, B, T, Kg, W, N, gstride, 1e-5)
    y_ref.backward(dy.double())
    xc, gc, bc = x.cuda(), gamma.detach().cuda(), beta.detach().cuda()
    y, stats = ops.groupnorm_fwd(xc, gc, bc, B, T, Kg, W, N, N, gstride, torch.float32, 1e-5)
    assert (y.cpu().double().reshape(B, T, Kg, W) - y_ref.detach()).abs().max().item() <= 2e-5
    dgam, dbet = torch.zeros_like(gc), torch.zeros_like(bc)
    dx = ops.groupnorm_bwd(xc, dy.cuda(), stats, gc, None, dgam, dbet, B, T, Kg, W, N, gstride, 1e-5)
    assert (dx.cpu().double() - xr.grad.double()).abs().max().item() <= 5e-5 * max(1.0, xr.grad.abs().max().item())
    assert (dgam.cpu().double() - gamma.grad.double()).abs().max().item() <= 1e-4 * max(1.0, gamma.grad.abs().max().item())
    assert (dbet.cpu().double() - beta.grad.double()).abs().max().item() <= 1e-4 * max(1.0, beta.grad.abs().max().item())
