"""CPU experiment (not a test; run by hand: ``python -m tests.exp_f16_emulation``): which operand formats does the forward need to
reach north_star's 1e-3 on the enhanced waveform?  The oracle's rounding-point emulation (oracle/bsrnn_ref.py ``_r``) is re-run with
the rounding function of each CALL SITE chosen separately (bf16 = 8 significant bits, f16 = 11), at N = 196, L = 6, 4 s @ 48 kHz
(401 steps on the time path), against the f32 oracle.  Output: profiles/r05_exp_f16_emulation_v1.log.

Sites: "rec"  = recurrent operands (W_hh, carried h), "proj" = input projection operands (x_n, W_ih) and whether the pre-activation is
stored in 16 bits between the two products ("gx16": the two-kernel form) or kept in f32 (fused kernels), "fc" = the Linear after each
BLSTM (its input is the stored h), "bs" = band split, "md" = mask decoder."""
import sys
import time

import torch
import torch.nn.functional as F

from oracle import bsrnn_ref, losses_ref, stft_ref

RND = {
    "f32": lambda x: x,
    "bf16": lambda x: x.to(torch.bfloat16).to(torch.float32),
    "f16": lambda x: x.to(torch.float16).to(torch.float32),
}


def lstm_bidir(lstm, x, rec, proj, gx16, hstore):
    S, T, _ = x.shape
    H = lstm.hidden_size
    outs = []
    for sfx, rev in (("", False), ("_reverse", True)):
        wih = getattr(lstm, "weight_ih_l0" + sfx)
        whh = getattr(lstm, "weight_hh_l0" + sfx)
        b = getattr(lstm, "bias_ih_l0" + sfx) + getattr(lstm, "bias_hh_l0" + sfx)
        gx = RND[gx16](F.linear(RND[proj](x), RND[proj](wih), b))
        h = x.new_zeros(S, H)
        c = x.new_zeros(S, H)
        hs = [None] * T
        whh_r = RND[rec](whh)
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            g = gx[:, t] + F.linear(h, whh_r)
            i_, f_, g_, o_ = g.chunk(4, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            hf = torch.sigmoid(o_) * torch.tanh(c)
            h = RND[rec](hf)
            hs[t] = RND[hstore](hf) if hstore != rec else h
        outs.append(torch.stack(hs, dim=1))
    return torch.cat(outs, dim=-1)


def forward(net, noisy, lens, fs, cfg):
    m = net.bsrnn.bsrnn
    n_fft, hop = stft_ref.reconfig_for_fs(net.n_fft, net.hop, fs, net.default_fs)
    spec, _ = stft_ref.stft(noisy, n_fft, hop, "hann", lens)
    x = torch.stack([spec.real, spec.imag], dim=-1)
    r_bs, r_md, r_fc = RND[cfg["bs"]], RND[cfg["md"]], RND[cfg["fc"]]
    # band split
    outs, hz = [], 0
    for i, sb in enumerate(m.band_split.subbands):
        xb = x[:, :, hz:hz + sb, :]
        if sb > xb.size(2):
            xb = F.pad(xb, (0, 0, 0, sb - xb.size(2)))
        xb = xb.reshape(xb.size(0), xb.size(1), -1)
        o = m.band_split.norm[i](xb.transpose(1, 2))
        o = F.conv1d(r_bs(o), r_bs(m.band_split.fc[i].weight), m.band_split.fc[i].bias)
        outs.append(o.unsqueeze(-1))
        hz += sb
        if hz >= x.size(2):
            break
    skip = torch.cat(outs, dim=-1)
    B, N, T, K = skip.shape
    for i in range(m.num_layer):
        out = m.norm_time[i](skip).transpose(1, 3).reshape(B * K, T, N)
        out = lstm_bidir(m.rnn_time[i], out, cfg["rec"], cfg["proj"], cfg["gx_time"], cfg["hstore"])
        out = F.linear(r_fc(out), r_fc(m.fc_time[i].weight), m.fc_time[i].bias)
        skip = skip + out.reshape(B, K, T, N).transpose(1, 3)
        out = m.norm_freq[i](skip).permute(0, 2, 3, 1).contiguous().reshape(B * T, K, N)
        out = lstm_bidir(m.rnn_freq[i], out, cfg["rec"], cfg["proj"], cfg["gx_band"], cfg["hstore"])
        out = F.linear(r_fc(out), r_fc(m.fc_freq[i].weight), m.fc_freq[i].bias)
        skip = skip + out.reshape(B, T, K, N).permute(0, 3, 1, 2).contiguous()
    ms, rs = [], []
    md = m.mask_decoder
    for i in range(len(md.subbands)):
        if i >= skip.size(-1):
            break
        xb = skip[:, :, :, i]
        for seq, acc in ((md.mlp_mask[i], ms), (md.mlp_residual[i], rs)):
            o = seq[0](xb)
            o = torch.tanh(F.conv1d(r_md(o), r_md(seq[1].weight), seq[1].bias))
            o = F.glu(F.conv1d(r_md(o), r_md(seq[3].weight), seq[3].bias), dim=1).transpose(1, 2).contiguous()
            acc.append(o.reshape(o.size(0), o.size(1), 1, -1, 2))
    mm = F.pad(torch.cat(ms, dim=3), (0, 0, 0, int(md.freq_dim - sum(t.size(3) for t in ms)))).moveaxis(1, 2)
    rr = F.pad(torch.cat(rs, dim=3), (0, 0, 0, int(md.freq_dim - sum(t.size(3) for t in rs)))).moveaxis(1, 2)
    mc = torch.view_as_complex(mm.contiguous())[..., :x.size(2)]
    rc = torch.view_as_complex(rr.contiguous())[..., :x.size(2)]
    enh = (mc * torch.view_as_complex(x.contiguous()).unsqueeze(1) + rc)[:, 0]
    return stft_ref.istft(enh, n_fft, hop, int(lens.max()))


def cfg(all_="bf16", **kw):
    c = dict(rec=all_, proj=all_, gx_time=all_, gx_band="f32", hstore=all_, fc=all_, bs=all_, md=all_)
    c.update(kw)
    return c


CASES = [
    ("shipped bf16 (time: two-kernel gx in bf16, band: fused)", cfg("bf16")),
    ("all f16, same structure", cfg("f16")),
    ("all f16, time-path projection fused too (gx f32)", cfg("f16", gx_time="f32")),
    ("all bf16, time-path projection fused too (gx f32)", cfg("bf16", gx_time="f32")),
    ("f16 recurrences + projections, h stored bf16 for fc (one h copy), fc/bs/md bf16", cfg("bf16", rec="f16", proj="f16", gx_time="f16")),
    ("f16 recurrences + projections + fc on f16 h; bs/md bf16", cfg("f16", bs="bf16", md="bf16")),
    ("f16 recurrences + projections, h stored bf16, fc on that bf16 h with f16... (fc bf16), bs/md f16", cfg("f16", hstore="bf16", fc="bf16")),
    ("only recurrent operands f16 (W_hh, h), everything else bf16", cfg("bf16", rec="f16", hstore="f16", fc="f16")),
    ("only recurrent operands bf16, everything else f16", cfg("f16", rec="bf16", hstore="bf16", fc="bf16")),
]


def main():
    torch.manual_seed(21)
    N, L, B, FS = 196, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 48000
    Ls = 4 * FS
    net = bsrnn_ref.BSRNN_SE(N, L)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(5)
    clean = 0.3 * torch.randn(B, Ls, generator=g)
    noisy = clean + 0.1 * torch.randn(B, Ls, generator=g)
    lens = torch.full((B,), Ls, dtype=torch.int32)
    with torch.no_grad():
        wav_r, _ = net(noisy, lens, FS, False)
        loss_r = float(losses_ref.mr_l1_loss(clean, wav_r).mean())
        print("f32 oracle: loss %.6g, wav peak %.4g  (B=%d x 4 s @ 48 kHz, N=%d, L=%d)" % (loss_r, float(wav_r.abs().max()), B, N, L), flush=True)
        for name, c in CASES:
            t0 = time.time()
            wav = forward(net, noisy, lens, FS, c)
            l2 = float((wav - wav_r).norm() / wav_r.norm())
            mx = float((wav - wav_r).abs().max() / wav_r.abs().max())
            el = abs(float(losses_ref.mr_l1_loss(clean, wav).mean()) - loss_r) / abs(loss_r)
            print("%-100s  wav rel. L2 %.2e  max/peak %.2e  loss %.2e   (%.0f s)" % (name, l2, mx, el, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
