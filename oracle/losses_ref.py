"""Oracle: espnet2 ``MultiResL1SpecLoss`` / ``SISNRLoss`` and the ``SEModel`` step.

TEST INFRASTRUCTURE - never imported by the product path.

Call sites: ``baseline_code/d_model.py:24-25`` (construction: windows
[256,512,768,1024], eps 1e-6, normalize_variance, td weight .5),
``d_model.py:61-89`` (forward_step), ``d_model.py:102-113`` +
``train_se.py:78`` (AdamW + StepLR + clip 0.5).  Loss arithmetic restated from
espnet==202412 ``espnet2/enh/loss/criterions/time_domain.py``
and ``fast_bss_eval.si_sdr_loss``; parity unpinned by the reference.
"""
import torch

from . import stft_ref


def mr_l1_loss(target, estimate, window_sz=(256, 512, 768, 1024), eps=1e-6,
               time_domain_weight=0.5, normalize_variance=True):
    """MultiResL1SpecLoss.forward, reduction='sum' -> [B]."""
    target = target.float()
    estimate = estimate.float()
    if normalize_variance:
        target = target / torch.std(target, dim=1, keepdim=True)
        estimate = estimate / torch.std(estimate, dim=1, keepdim=True)
    alpha = torch.sum(estimate * target, -1, keepdim=True) / (torch.sum(estimate ** 2, -1, keepdim=True) + eps)
    td = torch.sum((estimate * alpha - target).abs(), dim=-1)
    if len(window_sz) == 0:
        return td
    spec = torch.zeros_like(td)
    for w in window_sz:
        tm = stft_ref.stft(target, w, w // 2, window=None)[0].abs()
        em = stft_ref.stft(estimate * alpha, w, w // 2, window=None)[0].abs()
        spec = spec + torch.sum((em - tm).abs(), dim=(1, 2))
    return td * time_domain_weight + (1 - time_domain_weight) * spec / len(window_sz)


def si_snr_loss(ref, inf):
    """SISNRLoss(zero_mean=True, clamp_db=None) == fast_bss_eval.si_sdr_loss -> [B] (= -SI-SDR dB)."""
    ref = ref - ref.mean(dim=-1, keepdim=True)
    inf = inf - inf.mean(dim=-1, keepdim=True)
    ref = ref / torch.clamp(torch.linalg.norm(ref, dim=-1, keepdim=True), min=1e-6)
    inf = inf / torch.clamp(torch.linalg.norm(inf, dim=-1, keepdim=True), min=1e-6)
    coh = torch.sum(ref * inf, dim=-1) ** 2
    return 10.0 * torch.log10((1 - coh) / coh)


def forward_step(model, clean, noisy, fs, lengths, emulate_bf16=False):
    """SEModel.forward_step (d_model.py:61-89) -> (loss, -sisnr, se_speech)."""
    B = clean.shape[0]
    clean = clean.view(B, -1).float()
    noisy = noisy.view(B, -1).float()
    se = model(noisy, lengths, fs, emulate_bf16)[0]
    loss = mr_l1_loss(clean, se).mean()
    with torch.no_grad():
        sisnr = si_snr_loss(clean, se).mean()
    return loss, -sisnr, se


def make_optimizer(params, lr=1e-3, eps=1e-8, weight_decay=1e-6):
    """d_model.py:104-109."""
    return torch.optim.AdamW(params, lr=lr, eps=eps, weight_decay=weight_decay)


def train_step(model, opt, clean, noisy, fs, lengths, clip=0.5, emulate_bf16=False):
    """One optimisation step as Lightning runs it: closure (fwd+bwd) -> clip_grad_norm_(0.5) -> AdamW.step."""
    opt.zero_grad(set_to_none=True)
    loss, sisnr, _ = forward_step(model, clean, noisy, fs, lengths, emulate_bf16)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    return loss.detach(), sisnr, gn
