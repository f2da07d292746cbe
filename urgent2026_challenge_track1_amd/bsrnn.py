"""BSRNN_SE on MI355X: the reference's nn.Module surface over the hand-written HIP kernels.

Mirrors ``baseline_code/models/bsrnn.py:9-41`` (``BSRNN_SE(num_channel, num_layer)`` with
sub-modules ``encoder`` / ``decoder`` / ``bsrnn`` and ``forward(speech_mix, speech_lengths, fs)
-> (enhanced_wav, enhanced_feature)``) and the espnet2 ``BSRNNSeparator`` parameter tree it wraps
(``bsrnn.bsrnn.{band_split,norm_time,rnn_time,fc_time,norm_freq,rnn_freq,fc_freq,mask_decoder}``),
so reference checkpoints load by name.  The torch modules below are parameter CONTAINERS only:
their ``forward`` is never called.  All parameters are views into one flat f32 buffer (and one
flat gradient buffer) so the fused optimizer and the bucketed RCCL all-reduce see contiguous
memory; activations are channel-last ``[B, T, K, N]``.

Every arithmetic step is a kernel of liburse_hip.so; torch only allocates, views and records
the autograd graph (custom Functions whose backward writes parameter gradients straight into
the flat gradient buffer).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import call, stream_ptr

SUBBANDS_481 = tuple([5] + [4] * 19 + [10] * 6 + [40] * 7 + [60])   # reference bsrnn_flowse.py:29
SUBBANDS_769 = tuple([5] + [4] * 26 + [10] * 10 + [50] * 10 + [60])  # reference bsrnn_flowse.py:36
_DIAG_SKIP_WGRADS = os.environ.get("URSE_DIAG_SKIP_WGRADS", "0") == "1"
GN_EPS = 1e-5


def _ptr(t, off_elems=0):
    return t.data_ptr() + off_elems * t.element_size()


class _PackPlan:
    """A set of (cast + pad [+ transpose]) copies from the flat parameter buffer, run as ONE launch."""

    def __init__(self, transpose):
        self.transpose = transpose
        self.segs = []
        self.size = 0
        self._table = None
        self._out = {}

    def add(self, in_off, in_rows, in_cols, out_rows, out_cols):
        off = self.size
        self.segs.append([in_off, in_rows, in_cols, in_cols, off, out_rows, out_cols, out_cols])
        self.size += (out_rows * out_cols + 31) // 32 * 32
        return (off, out_rows, out_cols)

    def reserve(self, n):
        off = self.size
        self.size += (n + 31) // 32 * 32
        return off

    def add_at(self, in_off, in_rows, in_cols, out_off, out_rows, out_cols, out_ld):
        """copy a block to an explicit place of the output buffer (row pitch out_ld)."""
        self.segs.append([in_off, in_rows, in_cols, in_cols, out_off, out_rows, out_cols, out_ld])

    def run(self, flat, dtype):
        if not self.segs:
            return None
        if self._table is None or self._table.device != flat.device:
            self._table = torch.tensor(self.segs, dtype=torch.int64, device=flat.device)
        key = (dtype, flat.device)
        if key not in self._out:
            self._out[key] = torch.empty(self.size, dtype=dtype, device=flat.device)
        out = self._out[key]
        big = max(s[5] * s[6] for s in self.segs)
        call("pack_segments", flat, out, self._table, len(self.segs), max(1, min(64, (big + 2047) // 2048)),
             int(self.transpose), ops._dt(out), stream_ptr())
        return out


def _view(buf, h):
    off, r, c = h
    return buf[off:off + r * c].view(r, c)


def _empty_padded(K, M, ld, n, dtype, device):
    """[K, M, ld] buffer whose first n columns a GEMM will write: zeros only in the padding (the next GEMM reads it as K padding;
    a full zero fill of the two mask-decoder hidden tensors was 1.4 GB per pass)."""
    t = torch.empty(K, M, ld, dtype=dtype, device=device)
    if ld > n:
        t[:, :, n:].zero_()
    return t


def nt_grouped(rows, device, in_dt, out_dt, act=0):
    """one launch over per-band GEMM records {A, B, C, bias, resid, lda, ldb, ldc, M, N, K, ldr}; the library sees the
    host copy too and picks the kernel / grid (include/urse.h: urse_gemm_nt_grouped_h)."""
    host = np.asarray(rows, dtype=np.int64)
    assert host.ndim == 2 and host.shape[1] == 12
    call("gemm_nt_grouped_h", ops.upload(torch.from_numpy(host), device), host.ctypes.data, host.shape[0], in_dt, out_dt, act,
         stream_ptr())


class BSRNNCore(nn.Module):
    """espnet2 ``BSRNN`` (band split -> L x dual-path BLSTM -> mask decoder -> m*x + r), HIP-backed."""

    def __init__(self, input_dim=481, num_channel=16, num_layer=6, target_fs=48000, causal=False, num_spk=1,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        if causal or num_spk != 1:
            raise NotImplementedError("the reference uses causal=False, num_spk=1 (models/bsrnn.py:27-34)")
        if target_fs != 48000 or input_dim not in (481, 769):
            raise NotImplementedError("band tables defined for input_dim 481 / 769 @ 48 kHz (bsrnn_flowse.py:23-40)")
        if num_channel % 4:
            raise ValueError("num_channel must be a multiple of 4")
        self.subbands = SUBBANDS_481 if input_dim == 481 else SUBBANDS_769
        self.input_dim, self.N, self.H, self.num_layer = input_dim, num_channel, 2 * num_channel, num_layer
        # compute_dtype: operand format of the FORWARD contractions (bf16 | f16 | f32).  f16 = IEEE half, 11 significant bits at bf16's bytes and
        # MFMA rate: the forward then meets north_star's 1e-3 on the enhanced waveform (bf16: 4e-3).  Gradients need bf16's range, so under f16 every
        # backward operand is bf16 (self.bwd_dtype): tensors that are both a forward operand and a backward operand (normalised inputs, hidden
        # states, the mask decoder's tanh layer) are written in both formats by the kernel that produces them, when a backward will follow.
        self.compute_dtype = compute_dtype
        N, H = self.N, self.H
        self._make_front_back()
        self.norm_time = nn.ModuleList([nn.GroupNorm(1, N) for _ in range(num_layer)])
        self.rnn_time = nn.ModuleList([nn.LSTM(N, H, batch_first=True, bidirectional=True) for _ in range(num_layer)])
        self.fc_time = nn.ModuleList([nn.Linear(2 * H, N) for _ in range(num_layer)])
        self.norm_freq = nn.ModuleList([nn.GroupNorm(1, N) for _ in range(num_layer)])
        self.rnn_freq = nn.ModuleList([nn.LSTM(N, H, batch_first=True, bidirectional=True) for _ in range(num_layer)])
        self.fc_freq = nn.ModuleList([nn.Linear(2 * H, N) for _ in range(num_layer)])
        self._flat = None
        self._flat_grad = None
        self._off = {}
        self._plans = None
        self._tables = {}
        self._packed = None
        self._packed_version = -1
        self.param_version = 0          # bumped by the optimizer after every update
        self.grad_ready_hook = None     # callable(tag) fired when a parameter group's grads are final
        self._deferred, self._inflight, self._side = [], None, None     # weight-gradient GEMMs parked for the side stream
        self._grad_pack = None          # (data_ptr of a gradient stream tensor, its bf16 K-padded copy)
        self._gn_stats = None           # (a dual-path output, its GroupNorm statistics from the fc GEMM's epilogue)

    # ------------------------------------------------------------------------------------------
    # parameter containers (espnet names) of the parts that differ between the discriminative and the flow DNN
    # ------------------------------------------------------------------------------------------
    def _make_band_split(self):
        bs = nn.Module()
        bs.norm = nn.ModuleList([nn.GroupNorm(1, 2 * sb) for sb in self.subbands])
        bs.fc = nn.ModuleList([nn.Conv1d(2 * sb, self.N, 1) for sb in self.subbands])
        return bs

    def _make_front_back(self):
        N = self.N
        self.band_split = self._make_band_split()
        md = nn.Module()
        mk = lambda sb: nn.Sequential(nn.GroupNorm(1, N), nn.Conv1d(N, 4 * N, 1), nn.Tanh(),
                                      nn.Conv1d(4 * N, 4 * sb, 1), nn.GLU(dim=1))
        md.mlp_mask = nn.ModuleList([mk(sb) for sb in self.subbands])
        md.mlp_residual = nn.ModuleList([mk(sb) for sb in self.subbands])
        self.mask_decoder = md

    @staticmethod
    def _bs_params(prefix, bs):
        return [(prefix + ".gamma", [m.weight for m in bs.norm]), (prefix + ".beta", [m.bias for m in bs.norm]),
                (prefix + ".w", [m.weight for m in bs.fc]), (prefix + ".b", [m.bias for m in bs.fc])]

    def _front_params(self):
        return self._bs_params("bs", self.band_split)

    def _back_params(self):
        order = []
        for tag, mlps in (("m", self.mask_decoder.mlp_mask), ("r", self.mask_decoder.mlp_residual)):
            p = "md%s." % tag
            order += [(p + "gamma", [s[0].weight for s in mlps]), (p + "beta", [s[0].bias for s in mlps]),
                      (p + "w1", [s[1].weight for s in mlps]), (p + "b1", [s[1].bias for s in mlps]),
                      (p + "w2", [s[3].weight for s in mlps]), (p + "b2", [s[3].bias for s in mlps])]
        return order

    back_tag = "md"
    band_groups = ("bs", "mdm", "mdr")      # parameter groups that hold one tensor per band

    # ------------------------------------------------------------------------------------------
    # flat parameter / gradient buffers
    # ------------------------------------------------------------------------------------------
    def _ordered_params(self):
        """Flat layout: groups that the kernels read as one matrix are made contiguous."""
        L = self.num_layer
        order = list(self._front_params())
        for l in range(L):
            for path, norm, rnn, fc in (("t", self.norm_time[l], self.rnn_time[l], self.fc_time[l]),
                                        ("f", self.norm_freq[l], self.rnn_freq[l], self.fc_freq[l])):
                p = "l%d%s." % (l, path)
                order += [(p + "gamma", [norm.weight]), (p + "beta", [norm.bias]),
                          (p + "wih", [rnn.weight_ih_l0, rnn.weight_ih_l0_reverse]),
                          (p + "bih", [rnn.bias_ih_l0, rnn.bias_ih_l0_reverse]),
                          (p + "bhh", [rnn.bias_hh_l0, rnn.bias_hh_l0_reverse]),
                          (p + "whh", [rnn.weight_hh_l0, rnn.weight_hh_l0_reverse]),
                          (p + "wfc", [fc.weight]), (p + "bfc", [fc.bias])]
        return order + list(self._back_params())

    def _ensure_flat(self):
        first = next(self.parameters())
        if self._flat is not None and self._flat.device == first.device and \
                first.data_ptr() == self._flat.data_ptr() + 4 * self._first_off:
            return
        order = self._ordered_params()
        total = 0
        offs = {}
        plist = []
        for name, ps in order:
            total = (total + 3) // 4 * 4          # 16-byte aligned group starts
            offs[name] = total
            for p in ps:
                plist.append((p, total))
                total += p.numel()
        assert len(plist) == len([q for q in self.parameters() if q.requires_grad])
        # slot map of the optimizer: 0 = parameters every batch uses, 1 + k = parameters of band k (without a gradient
        # when fs puts fewer than k + 1 bands in the spectrum: torch.optim.AdamW skips those, and so does ours)
        nb = len(self.subbands)
        slot = np.zeros(total, dtype=np.uint8)
        for name, ps in order:
            if len(ps) == nb and name.split(".")[0] in self.band_groups:
                o = offs[name]
                for k, q in enumerate(ps):
                    slot[o:o + q.numel()] = 1 + k
                    o += q.numel()
        self.n_slot = 1 + nb
        self._slot_map = torch.from_numpy(slot).to(first.device)
        flat = torch.zeros(total, dtype=torch.float32, device=first.device)
        # the gradients are followed by one "used" flag per slot: part of the last all-reduce bucket, so that a band
        # counts as used when ANY rank used it (what DDP's find_unused_parameters does), zeroed with the gradients
        self._n_params = total
        grad = torch.zeros(total + (self.n_slot + 3) // 4 * 4, dtype=torch.float32, device=first.device)
        with torch.no_grad():
            for p, o in plist:
                flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
                p.data = flat[o:o + p.numel()].view(p.shape)
                p.grad = grad[o:o + p.numel()].view(p.shape)
        self._flat, self._flat_grad, self._off = flat, grad, offs
        self._first_off = [o for p, o in plist if p is first][0]
        self._plans = None
        self._packed_version = -1
        self._tables = {}

    @property
    def flat_params(self):
        self._ensure_flat()
        return self._flat

    @property
    def flat_grads(self):
        """gradients of every parameter, then the per-slot "used" flags (see _ensure_flat)."""
        self._ensure_flat()
        return self._flat_grad

    @property
    def used_flags(self):
        self._ensure_flat()
        return self._flat_grad[self._n_params:self._n_params + self.n_slot]

    @property
    def slot_map(self):
        self._ensure_flat()
        return self._slot_map

    def mark_used_bands(self, K):
        """training forward at K bands: slots 0..K get a gradient this step, the bands above do not."""
        call("fill_used_flags", self.used_flags, self.n_slot, 1 + K, stream_ptr())

    def grad_groups(self):
        """[(tag, offset, numel)] in the order backward completes them (for bucketed all-reduce)."""
        self._ensure_flat()
        names = list(self._off.keys())
        ends = [self._off[n] for n in names[1:]] + [self._flat_grad.numel()]     # (the last group carries the used flags)
        spans = {n: (self._off[n], e - self._off[n]) for n, e in zip(names, ends)}
        groups = []

        def span(ns):
            lo = min(spans[n][0] for n in ns)
            hi = max(spans[n][0] + spans[n][1] for n in ns)
            return lo, hi - lo
        is_layer = lambda n: n[0] == "l" and n[1].isdigit()
        first = min(i for i, n in enumerate(names) if is_layer(n))
        last = max(i for i, n in enumerate(names) if is_layer(n))
        groups.append((self.back_tag,) + span(names[last + 1:]))
        for l in reversed(range(self.num_layer)):
            groups.append(("l%df" % l,) + span([n for n in names if n.startswith("l%df." % l)]))
            groups.append(("l%dt" % l,) + span([n for n in names if n.startswith("l%dt." % l)]))
        groups.append(("bs",) + span(names[:first]))
        return groups

    def _g(self, name, numel=None, off=0):
        o = self._off[name] + off
        n = numel if numel is not None else 0
        return self._flat_grad[o:o + n] if numel is not None else self._flat_grad[o:]

    def _p(self, name, numel, off=0):
        o = self._off[name] + off
        return self._flat[o:o + numel]

    def _ready(self, tag):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(tag)

    # ------------------------------------------------------------------------------------------
    # static tables / packed weights
    # ------------------------------------------------------------------------------------------
    def _build_plans(self, dtype):
        N, H = self.N, self.H
        Np, Hp = ops.kpad(N, dtype), ops.kpad(ops.pad_to(H, 16), dtype)
        ld2H, ld4N = ops.kpad(2 * H, dtype), ops.kpad(4 * N, dtype)
        pn, pt = _PackPlan(False), _PackPlan(True)
        h = {}
        o = self._off
        self._dims = dict(Np=Np, Hp=Hp, ld2H=ld2H, ld4N=ld4N)
        self._plan_front(pn, pt, h, dtype)
        for l in range(self.num_layer):
            for path in "tf":
                p = "l%d%s." % (l, path)
                h[p + "wfc"] = pn.add(o[p + "wfc"], N, 2 * H, N, ld2H)
                h[p + "wfcT"] = pt.add(o[p + "wfc"], N, 2 * H, 2 * H, Np)
        self._plan_back(pn, pt, h, dtype)
        self._plans = (dtype, pn, pt, h)

    def _plan_band_split(self, prefix, pn, pt, h, dtype):
        N, Np, o = self.N, self._dims["Np"], self._off
        w_off = 0
        for k, sb in enumerate(self.subbands):
            xpad = ops.kpad(2 * sb, dtype)
            h[prefix + ".w", k] = pn.add(o[prefix + ".w"] + w_off, N, 2 * sb, N, xpad)
            h[prefix + ".wT", k] = pt.add(o[prefix + ".w"] + w_off, N, 2 * sb, 2 * sb, Np)
            w_off += N * 2 * sb

    def _plan_front(self, pn, pt, h, dtype):
        self._plan_band_split("bs", pn, pt, h, dtype)

    def _plan_back(self, pn, pt, h, dtype):
        N, Np, ld4N, o = self.N, self._dims["Np"], self._dims["ld4N"], self._off
        for tag in "mr":
            p = "md%s." % tag
            w2_off = 0
            for k, sb in enumerate(self.subbands):
                ppad = ops.kpad(4 * sb, dtype)
                h[p + "w1", k] = pn.add(o[p + "w1"] + k * 4 * N * N, 4 * N, N, 4 * N, Np)
                h[p + "w1T", k] = pt.add(o[p + "w1"] + k * 4 * N * N, 4 * N, N, N, ld4N)
                h[p + "w2", k] = pn.add(o[p + "w2"] + w2_off, 4 * sb, 4 * N, 4 * sb, ld4N)
                h[p + "w2T", k] = pt.add(o[p + "w2"] + w2_off, 4 * sb, 4 * N, 4 * N, ppad)
                w2_off += 4 * sb * 4 * N

    @property
    def bwd_dtype(self):
        return ops.bwd_dtype(self.compute_dtype)

    def _prepare(self):
        """(re)pack the GEMM operands from the f32 master weights; once per parameter version."""
        self._ensure_flat()
        dtype = self.compute_dtype
        if self._plans is None or self._plans[0] != dtype:
            self._build_plans(dtype)
            self._packed_version = -1
        if self._packed_version == self.param_version:
            return
        _, pn, pt, h = self._plans
        bn, bt = pn.run(self._flat, dtype), pt.run(self._flat, self.bwd_dtype)     # (the transposed copies are the dgrad GEMMs' weights)
        H = self.H
        pk = {}
        for key, hd in h.items():
            name = key[0] if isinstance(key, tuple) else key
            pk[key] = _view(bt if name.endswith("T") else bn, hd)
        N = self.N
        if not hasattr(self, "_lstm_bufs") or self._lstm_bufs.get("key") != (dtype, self._flat.device):
            self._lstm_bufs = {"key": (dtype, self._flat.device)}
        names = ["l%d%s." % (l, path) for l in range(self.num_layer) for path in "tf"]
        srcs = {p: (self._p(p + "wih", 8 * H * N), self._p(p + "whh", 8 * H * H), self._p(p + "bih", 8 * H), self._p(p + "bhh", 8 * H))
                for p in names}
        multi = (ops.PACK_MULTI and self._flat.is_cuda and all(p in self._lstm_bufs for p in names) and
                 all(set(k for k in ("whhq", "whhb", "wx", "whhb_rw", "whhTq", "wihq") if self._lstm_bufs[p].get(k) is not None) ==
                     self._lstm_layouts(p[-2]) <= {"whhq", "whhb", "wx", "wihq"} for p in names))
        if multi:
            # the buffers exist (every step after the first): one launch per layout for all 12 LSTMs instead of three to five per LSTM
            self._lstm_bufs["table"] = ops.lstm_pack_multi([srcs[p] + (self._lstm_bufs[p],) for p in names], N, H, dtype,
                                                           table=self._lstm_bufs.get("table"))
        for l in range(self.num_layer):
            for path in "tf":
                p = "l%d%s." % (l, path)
                lp = self._lstm_bufs[p] if multi else ops.lstm_pack(*srcs[p], N, H, dtype, out=self._lstm_bufs.get(p),
                                                                    layouts=self._lstm_layouts(path))
                self._lstm_bufs[p] = lp
                pk[p + "wih"], pk[p + "wihT"], pk[p + "bias"] = lp["wih"], lp["wihT"], lp["bias"]
                pk[p + "whh"], pk[p + "whhT"] = lp["whh"], lp["whhT"]
                pk[p + "whhq"], pk[p + "whhTq"] = lp.get("whhq"), lp.get("whhTq")
                pk[p + "wihq"] = lp.get("wihq")
                pk[p + "whhb"] = lp.get("whhb")
                pk[p + "whhb_rw"] = lp.get("whhb_rw")
                pk[p + "wx"] = lp.get("wx")
        self._packed = pk
        self._packed_version = self.param_version

    def _lstm_layouts(self, path):
        """optional weight layouts this half layer's dispatch can reach (ops.model_lstm_layouts, narrowed by the path where the kernels are
        path-specific at this hidden size: the fused row-wave forward and its unfused fallbacks serve many short sequences, i.e. the band path)."""
        lay = ops.model_lstm_layouts()
        if self.compute_dtype == torch.float16:
            lay &= {"whhq", "wihq", "wx"}          # what ops.lstm_pack produces for f16 operands (the set never matched, 12 LSTMs were re-packed one by one)
        if self.H == 392 and path == "t":
            lay -= {"wx", "whhb_rw"}
        d = self._dims
        if (self.H == 392 and path == "f" and not ops.BAND_CLUSTERX) or not ops.lstm_clusterx_supported(self.N, d["Np"], self.H, d["Hp"]):
            lay -= {"wihq"}          # (the fused cluster forward: the time path, and - round 6, in rounds - the band path; other shapes have no such kernel)
        return lay

    def _band_tables(self, F, dtype, device):
        key = (F, dtype, device)
        if key in self._tables:
            return self._tables[key]
        rows, f2k = [], [-1] * F
        f0 = xoff = poff = 0
        K = 0
        for k, sb in enumerate(self.subbands):
            xpad, ppad = ops.kpad(2 * sb, dtype), ops.kpad(4 * sb, dtype)
            rows.append([f0, sb, xoff, xpad, 2 * f0, poff, ppad, 0])
            for f in range(f0, min(F, f0 + sb)):
                f2k[f] = k
            f0 += sb
            xoff += xpad
            poff += ppad
            K = k + 1
            if f0 >= F:
                break
        tb = dict(K=K, rows=rows, ldx=xoff, P=poff,
                  bands=torch.tensor(rows, dtype=torch.int32, device=device),
                  f2k=torch.tensor(f2k, dtype=torch.int32, device=device))
        self._tables[key] = tb
        return tb

    # ------------------------------------------------------------------------------------------
    # band split
    # ------------------------------------------------------------------------------------------
    def bandsplit_fwd(self, spec, prefix="bs", out=None, width=None, col0=0, save=True):
        """band split of spec [B,T,F,2]; default -> z f32 [B,T,K,N]; with `out` (a [B,T,K,width] tensor of the compute
        dtype) the N channels are written at column offset col0 (flow: x / y halves of the condition_fc input)."""
        B, T, F, _ = spec.shape
        dt, dev, N = self.compute_dtype, spec.device, self.N
        tb = self._band_tables(F, dt, dev)
        K, pk = tb["K"], self._packed
        n_gb = 2 * sum(self.subbands)
        xnb = torch.empty(B * T, tb["ldx"], dtype=dt, device=dev)
        xnb_b = torch.empty_like(xnb, dtype=torch.bfloat16) if (save and dt == torch.float16) else None    # bf16 copy for the weight gradient
        stats = torch.empty(B * K * 2, dtype=torch.float64, device=dev)
        call("bandsplit_norm_fwd", spec, tb["bands"], self._p(prefix + ".gamma", n_gb), self._p(prefix + ".beta", n_gb),
             xnb, stats, B, T, F, K, tb["ldx"], GN_EPS, ops._dt(xnb), xnb_b, stream_ptr())
        if out is None:
            out, width = torch.empty(B, T, K, N, dtype=torch.float32, device=dev), N
        M = B * T
        rows = []
        for k in range(K):
            r = tb["rows"][k]
            w = pk[prefix + ".w", k]
            rows.append([_ptr(xnb, r[2]), _ptr(w), _ptr(out, k * width + col0),
                         _ptr(self._flat, self._off[prefix + ".b"] + k * N), 0,
                         tb["ldx"], w.shape[1], K * width, M, N, r[3], 0])
        nt_grouped(rows, dev, ops._dt(xnb), ops._dt(out))
        return out, ((xnb_b if xnb_b is not None else xnb), stats, tb)

    def bandsplit_bwd(self, spec, saved, dz, prefix="bs", dzT=None, width=None, col0=0, ready=True):
        """dz f32 [B,T,K,N], or (flow) dzT = [B*T*K, width] of the compute dtype holding the N gradient channels at
        column offset col0 (width-col0 >= kpad(N), zero padded)."""
        xnb, stats, tb = saved
        B, T, F, _ = spec.shape
        dt, dev, N = self.bwd_dtype, spec.device, self.N
        K, pk, Np = tb["K"], self._packed, self._dims["Np"]
        M = B * T
        tail_overlap = ops.TN_OVERLAP and ops.TN_OVERLAP_TAIL and spec.is_cuda and bool(self._deferred)
        if tail_overlap:
            # the last half layer's weight gradients (2.4 ms alone) have no recurrence left to hide under: they run on the second queue beside
            # this function's small kernels and are joined at its end
            self._run_deferred_wgrads(dev, 256)
        else:
            self._flush_deferred_wgrads()        # the dual-path layers are behind us: nothing left to hide them under
        if dzT is None:
            dzT, width = ops.pack2d(dz.reshape(M * K, N), M * K, Np, dt), Np
        dxnb = torch.empty(M, tb["ldx"], dtype=torch.float32, device=dev)
        rows, tn_rows = [], []
        w_off = 0
        for k in range(K):
            r = tb["rows"][k]
            sb = r[1]
            a = dzT.view(M, K * width)[:, k * width + col0:k * width + col0 + Np]
            gw = self._g(prefix + ".w", N * 2 * sb, w_off).view(N, 2 * sb)
            tn_rows.append(ops.tn_desc(a, xnb[:, r[2]:r[2] + r[3]], gw, colsum=self._g(prefix + ".b", N, k * N), Mo=N,
                                       No=2 * sb))
            wT = pk[prefix + ".wT", k]
            rows.append([_ptr(dzT, k * width + col0), _ptr(wT), _ptr(dxnb, r[2]), 0, 0, K * width, Np, tb["ldx"], M,
                         2 * sb, Np, 0])
            w_off += N * 2 * sb
        ops.gemm_tn_grouped(tn_rows, dt, dev)     # all per-band weight / bias gradients in one launch
        nt_grouped(rows, dev, ops._dt(dzT), ops.F32)
        n_gb = 2 * sum(self.subbands)
        call("bandsplit_norm_bwd", spec, dxnb, tb["bands"], stats, self._g(prefix + ".gamma", n_gb),
             self._g(prefix + ".beta", n_gb), B, T, F, K, tb["ldx"], GN_EPS, stream_ptr())
        if tail_overlap:
            self._flush_deferred_wgrads()        # joins the second queue (and signals the joined gradients' tags)
        if ready:
            self._ready("bs")

    # ------------------------------------------------------------------------------------------
    # dual-path half layer: GN -> BLSTM -> Linear -> +skip   (time: sequences along T, band: along K)
    # ------------------------------------------------------------------------------------------
    def _seqmap(self, path, B, T, K):
        if path == "t":
            return dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
        return dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)

    def dualpath_fwd(self, skip, l, path, save, temb=None):
        B, T, K, N = skip.shape
        H, dt, pk, d = self.H, self.compute_dtype, self._packed, self._dims
        p = "l%d%s." % (l, path)
        M = B * T * K
        # the statistics of `skip` came out of the GEMM that produced it (the previous half layer's fc + residual), if one did
        pre = self._gn_stats[1] if (self._gn_stats is not None and self._gn_stats[0].data_ptr() == skip.data_ptr() and
                                    self._gn_stats[0].shape == skip.shape) else None
        self._gn_stats = None
        sm = self._seqmap(path, B, T, K)
        # f16 forward with a backward to follow: x_n and h are operands of the weight-gradient GEMMs, whose other operand (the gate / output
        # gradients) is bf16.  Where the library has the mixed-operand kernels for this half layer's shapes (round 6: f16 -> bf16 in registers
        # behind the fragment read) the backward reads the f16 tensors themselves; otherwise the producing kernels write them once more in bf16.
        two = save and dt == torch.float16
        if two and ops.tn_act_f16_supported(M, N, 2 * H, 0, True) and \
                ops.tn_act_f16_supported(M, 4 * H, N, H, True, sm["stride"], sm["seq_len"]):
            two = False
        xn, stats, xn_b = ops.groupnorm_fwd(skip, self._p(p + "gamma", N), self._p(p + "beta", N), B, T, 1, K * N, N,
                                            d["Np"], 0, dt, GN_EPS, add=temb, stats=pre, bf16_copy=two or None)
        cx_ok = (ops.USE_CLUSTERX_LSTM and ops.USE_CLUSTER_LSTM and dt in ops.HALF_TYPES and pk.get(p + "wihq") is not None and
                 pk.get(p + "whhq") is not None and H not in ops.CLUSTER2_H and not (path == "f" and ops.BAND_PATH_NO_CLUSTER) and
                 ops.lstm_clusterx_supported(N, d["Np"], H, d["Hp"]))
        # the band path in ROUNDS through the fused cluster forward (round 6), where the plan's rounds x steps price below the row-wave kernel
        band_cx = cx_ok and path == "f" and ops.BAND_CLUSTERX and ops.band_clusterx_pays(H, d["Hp"], sm["n_seq"], sm["seq_len"])
        fused = (not band_cx and ops.USE_RWX_LSTM and ops.USE_RW_LSTM and dt in ops.HALF_TYPES and pk.get(p + "wx") is not None and
                 sm["n_seq"] >= ops.RW_MIN_SEQ and not (ops.USE_CLUSTER_LSTM and not (path == "f" and ops.BAND_PATH_NO_CLUSTER) and
                                                      ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is not None))
        # (fused: the input projection runs inside the recurrence kernel - no gate GEMM, no [M, 8H] pre-activation matrix)
        # (the time path above the clusters' capacity - more than 33 utterances per GPU at 48 kHz - in rounds too, instead of gate GEMM + streaming forward)
        time_cx = (cx_ok and path == "t" and ops.TIME_CLUSTERX_ROUNDS and ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is None and
                   ops.lstm_clusterx_plan(H, d["Hp"], sm["n_seq"]) is not None)
        fused_c = not fused and cx_ok and (band_cx or time_cx or ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is not None)
        gx = None if (fused or fused_c) else ops.gemm_nt(xn, pk[p + "wih"], pk[p + "bias"])
        hout_b = None
        if fused_c:
            # the cluster forward with the projection fused (the time path at C2)
            r = ops.lstm_fwd_clusterx(xn, pk[p + "wihq"], pk[p + "whhq"], pk[p + "bias"], N, H, d["Hp"], save=save, bf16_copy=two, **sm)
            gx, hout, c, self._cluster_err = r[:4]
            hout_b = r[4] if two else None
        elif fused:
            r = ops.lstm_fwd_rwx(xn, pk[p + "wx"], pk[p + "bias"], N, H, d["Hp"], save=save, bf16_copy=two, **sm)
            gx, hout, c = r[:3]
            hout_b = r[3] if two else None
        elif dt == torch.float16:
            # f16 operands: the cluster forward where its plan fits (the time path at C2; H = 768, the flow DNN, forward only), else the streaming kernel
            if ops.USE_CLUSTER_LSTM and pk.get(p + "whhq") is not None and H in ops.CLUSTER2_H and not two and \
                    ops.lstm_cluster2_chunks(H, d["Hp"], **sm) is not None:
                r = ops.lstm_fwd_cluster2(gx, pk[p + "whhq"], H, d["Hp"], save=save, **sm)
                hout, c, self._cluster_err = r
            elif ops.USE_CLUSTER_LSTM and pk.get(p + "whhq") is not None and H not in ops.CLUSTER2_H and not (path == "f" and ops.BAND_PATH_NO_CLUSTER) and \
                    ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is not None:
                r = ops.lstm_fwd_cluster(gx, pk[p + "whhq"], H, d["Hp"], save=save, bf16_copy=two, **sm)
                hout, c, self._cluster_err = r[:3]
            else:
                r = ops.lstm_fwd(gx, pk[p + "whh"], H, d["Hp"], save=save, bf16_copy=two, **sm)
                hout, c = r[:2]
            hout_b = r[-1] if two else None
            gx = gx.view(torch.bfloat16)          # the kernels wrote the gate activations back in bf16 (the BPTT's operand format)
        elif ops.USE_CLUSTER_LSTM and pk.get(p + "whhq") is not None and H in ops.CLUSTER2_H and \
                ops.lstm_cluster2_chunks(H, d["Hp"], **sm) is not None:
            hout, c, err = ops.lstm_fwd_cluster2(gx, pk[p + "whhq"], H, d["Hp"], save=save, **sm)
            self._cluster_err = err
        elif ops.USE_CLUSTER_LSTM and pk.get(p + "whhq") is not None and not (path == "f" and ops.BAND_PATH_NO_CLUSTER) and \
                ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is not None:
            hout, c, err = ops.lstm_fwd_cluster(gx, pk[p + "whhq"], H, d["Hp"], save=save, **sm)
            self._cluster_err = err
        elif ops.USE_RW_LSTM and pk.get(p + "whhb") is not None and sm["n_seq"] >= ops.RW_MIN_SEQ and ops.lstm_rw_supported(H, d["Hp"]):
            if ops.RW_PAIRED and pk.get(p + "whhb_rw") is not None:
                hout, c = ops.lstm_fwd_rw(gx, pk[p + "whhb_rw"], H, d["Hp"], save=save, paired=True, **sm)
            else:
                hout, c = ops.lstm_fwd_rw(gx, pk[p + "whhb"], H, d["Hp"], save=save, **sm)
        elif ops.USE_WIDE_LSTM and pk.get(p + "whhb") is not None and sm["n_seq"] >= ops.WIDE_MIN_SEQ:
            hout, c = ops.lstm_fwd_wide(gx, pk[p + "whhb"], H, d["Hp"], save=save, **sm)
        else:
            hout, c = ops.lstm_fwd(gx, pk[p + "whh"], H, d["Hp"], save=save, **sm)
        out = torch.empty_like(skip)
        if ops.FUSE_GN_STATS and dt in ops.HALF_TYPES and N % 4 == 0 and not (path == "f" and l == self.num_layer - 1):
            # the next half layer normalises `out` over each batch element (T * K rows): its sums ride on this GEMM's epilogue
            _, st = ops.gemm_nt(hout, pk[p + "wfc"], self._p(p + "bfc", N), resid=skip.view(M, N), out=out.view(M, N), gn_rows=T * K)
            self._gn_stats = (out, st)      # (holding `out` keeps its storage from being handed to another tensor)
        else:
            ops.gemm_nt(hout, pk[p + "wfc"], self._p(p + "bfc", N), resid=skip.view(M, N), out=out.view(M, N))
        return out, ((stats, xn_b, gx, c, (hout_b if hout_b is not None else hout)) if save else None)

    def dualpath_bwd(self, skip, saved, l, path, dout):
        stats, xn, gates, c, hout = saved
        B, T, K, N = skip.shape
        H, dt, pk, d = self.H, self.bwd_dtype, self._packed, self._dims       # (f16 forward: every operand here is bf16, see __init__)
        p = "l%d%s." % (l, path)
        M = B * T * K
        dout2 = dout.reshape(M, N)
        if self._grad_pack is not None and self._grad_pack[0] == dout.data_ptr() and self._grad_pack[1].shape == (M, d["Np"]):
            doT = self._grad_pack[1]              # written by the GroupNorm backward that produced `dout`
        else:
            doT = ops.pack2d(dout2, M, d["Np"], dt)
        self._grad_pack = None
        dh = torch.empty(M, d["ld2H"], dtype=dt, device=skip.device)
        ops.gemm_nt(doT, pk[p + "wfcT"], out=dh, N=2 * H)
        sm = self._seqmap(path, B, T, K)
        overlap = ops.TN_OVERLAP and skip.is_cuda
        use_nsplit = ops.use_nsplit_bwd(H, d["Hp"], dt, path, sm, pk.get(p + "whhTq") is not None)
        if overlap and (path == "t" or ops.TN_OVERLAP_BAND):
            # the time path's BPTT occupies 136 of the 256 CUs for ~7 ms (and the band path's last round of workgroups
            # leaves most CUs idle): the weight-gradient GEMMs deferred by the previous half layers run beside it on a
            # second stream (they only feed the optimizer / all-reduce)
            self._run_deferred_wgrads(skip.device, ops.wgrad_shadow_wgs(path, use_nsplit), None if path == "t" else ops.TN_BAND_PARTS)
        if ops.USE_CLUSTER_LSTM_BWD and pk.get(p + "whhTq") is not None and \
                ops.lstm_cluster_plan(H, d["Hp"], sm["n_seq"]) is not None:
            dg, self._cluster_err = ops.lstm_bwd_cluster(dh, gates, c, pk[p + "whhTq"], H, d["Hp"], **sm)
        elif (ops.USE_SPLIT_LSTM_BWD or H >= ops.SPLIT_BWD_MIN_H) and dt == torch.bfloat16 and \
                ops.lstm_split_chunks(H, **sm) is not None:
            dg, self._cluster_err = ops.lstm_bwd_split(dh, gates, c, pk[p + "whhT"], H, **sm)
        elif use_nsplit:
            dg, self._cluster_err = ops.lstm_bwd_nsplit(dh, gates, c, pk[p + "whhT"], H, **sm)
        else:
            dg = ops.lstm_bwd(dh, gates, c, pk[p + "whhT"], H, rows16=ops.BWD_ROWS16.get(path, 0), **sm)   # dgates, gate-interleaved columns
        if overlap and path == "t":
            self._join_deferred_wgrads(keep=ops.TN_JOIN_LAG)
        st, L = sm["stride"], sm["seq_len"]
        tag = "l%d%s" % (l, path)

        # three launches per half layer, deferred separately (the band path's BPTT gets only the first ops.TN_BAND_PARTS of
        # them for company, see _run_deferred_wgrads); the half layer's gradients are final with the last one
        def wg_fc(target_wgs=0):
            if _DIAG_SKIP_WGRADS:                       # timing diagnostic only (gradients wrong): bound on what the TN GEMMs cost
                return
            ops.gemm_tn(doT, hout, self._g(p + "wfc", N * 2 * H).view(N, 2 * H), colsum=self._g(p + "bfc", N), Mo=N, No=2 * H,
                        target_wgs=target_wgs)

        def wg_dir(dr, sh, inv, last):
            def run(target_wgs=0):
                if _DIAG_SKIP_WGRADS:
                    return
                gb = self._g(p + "bih", 8 * H)
                gwih = self._g(p + "wih", 8 * H * N).view(8 * H, N)
                # per direction ONE pass over the [M, 4H] dgates yields dW_ih (+ bias gradient) and dW_hh
                ops.gemm_tn_dual(dg[:, dr * 4 * H:(dr + 1) * 4 * H], xn, gwih[dr * 4 * H:(dr + 1) * 4 * H],
                                 gb[dr * 4 * H:(dr + 1) * 4 * H], hout[:, dr * H:(dr + 1) * H],
                                 self._g(p + "whh", 4 * H * H, dr * 4 * H * H).view(4 * H, H), 4 * H, N, H, sh, st, L, inv,
                                 perm_h=H, target_wgs=target_wgs)
                if last:
                    call("axpby", gb, self._g(p + "bhh", 8 * H), 1.0, 1.0, 8 * H, stream_ptr())
            return run

        parts = [(wg_fc, None), (wg_dir(0, -st, 0, False), None), (wg_dir(1, st, L - 1, True), tag)]
        if overlap:
            self._deferred.extend(parts)               # the closures keep doT / hout / dg / xn alive until they have run
        else:
            for fn, _ in parts:
                fn()
        gn_sums = None
        if ops.FUSE_GN_BWD and ops.gemm_nt_gnbwd_supported(M, N, dg.shape[1], T * K, dt):
            # the reduce pass of the GroupNorm backward (sums of dxn against the normalised skip) rides on the dgrad GEMM's epilogue
            dxn, gn_sums = ops.gemm_nt_gnbwd(dg, pk[p + "wihT"], N, skip, stats, self._p(p + "gamma", N), self._g(p + "gamma", N),
                                             self._g(p + "beta", N), T * K, GN_EPS)
        else:
            dxn = ops.gemm_nt(dg, pk[p + "wihT"], out_dtype=torch.float32, N=N)
        if dt == torch.bfloat16 and N % 4 == 0:
            dskip, packed = ops.groupnorm_bwd(skip, dxn, stats, self._p(p + "gamma", N), dout, self._g(p + "gamma", N),
                                              self._g(p + "beta", N), B, T, 1, K * N, N, 0, GN_EPS, pack_ld=d["Np"], sums=gn_sums)
            self._grad_pack = (dskip.data_ptr(), packed)
        else:
            dskip = ops.groupnorm_bwd(skip, dxn, stats, self._p(p + "gamma", N), dout, self._g(p + "gamma", N),
                                      self._g(p + "beta", N), B, T, 1, K * N, N, 0, GN_EPS)
        if not overlap:
            self._ready(tag)
        return dskip

    # deferred weight-gradient GEMMs (see dualpath_bwd) --------------------------------------------------------------
    def _run_deferred_wgrads(self, device, target_wgs, limit=None):
        """launch what is deferred so far (the first `limit` launches of it) on the side stream, gated on the compute
        stream's current position."""
        if not self._deferred or limit == 0:
            return
        now, later = (self._deferred, []) if limit is None else (self._deferred[:limit], self._deferred[limit:])
        if self._side is None:
            self._side = ops.low_priority_stream(device)
        start = torch.cuda.Event()
        start.record(torch.cuda.current_stream())
        self._side.wait_event(start)
        with torch.cuda.stream(self._side):
            for fn, _ in now:
                fn(target_wgs)                                # they share the chip with a BPTT kernel
            done = torch.cuda.Event()
            done.record(self._side)
        self._inflight = (self._inflight or []) + [(done, now)]
        self._deferred = later
        ops.CO_RESIDENT_WGS = target_wgs          # a cooperative recurrence launched from here on shares the chip with these

    def _join_deferred_wgrads(self, keep=0):
        """wait (on the compute stream) for the side stream's batches, except the `keep` most recent ones."""
        if self._inflight is None:
            return
        n = len(self._inflight) - keep
        if n <= 0:
            return
        batches, rest = self._inflight[:n], self._inflight[n:]
        self._inflight = rest or None
        if not rest:
            ops.CO_RESIDENT_WGS = 0                      # (the compute stream waits below: nothing of the second queue is resident after)
        for done, items in batches:
            torch.cuda.current_stream().wait_event(done)
            for _, tag in items:                         # gradients final: tell the reducer (and drop the closures)
                if tag is not None:
                    self._ready(tag)

    def _flush_deferred_wgrads(self):
        """end of backward: whatever is still deferred runs on the compute stream."""
        self._join_deferred_wgrads()
        items, self._deferred = self._deferred, []
        for fn, tag in items:
            fn()
            if tag is not None:
                self._ready(tag)

    # ------------------------------------------------------------------------------------------
    # mask decoder + complex mask apply
    # ------------------------------------------------------------------------------------------
    def maskdec_fwd(self, skip, spec, save):
        B, T, K, N = skip.shape
        F = spec.shape[2]
        dt, dev, pk, d = self.compute_dtype, skip.device, self._packed, self._dims
        tb = self._band_tables(F, dt, dev)
        assert tb["K"] == K
        M, Np, ld4N, P = B * T, d["Np"], d["ld4N"], tb["P"]
        Kf = len(self.subbands)
        xns, sts, hids, pres, keep = [], [], [], [], []
        rows1, rows2 = [], []
        for tag in "mr":
            p = "md%s." % tag
            two = save and dt == torch.float16            # f16 forward, backward to follow: x_n and the tanh layer also in bf16
            if two:
                xn, st, xn_b = ops.groupnorm_fwd(skip, self._p(p + "gamma", Kf * N), self._p(p + "beta", Kf * N), B, T, K, N, N,
                                                 Np, N, dt, GN_EPS, bf16_copy=True)
            else:
                xn, st = ops.groupnorm_fwd(skip, self._p(p + "gamma", Kf * N), self._p(p + "beta", Kf * N), B, T, K, N, N,
                                           Np, N, dt, GN_EPS)
                xn_b = xn
            hid = _empty_padded(K, M, ld4N, 4 * N, dt, dev)     # the GEMM writes the 4N columns; only the K padding must be zero
            hid_b = _empty_padded(K, M, ld4N, 4 * N, torch.bfloat16, dev) if two else hid
            pre = torch.empty(M, P, dtype=torch.float32, device=dev)
            b2_off = 0
            for k in range(K):
                sb = self.subbands[k]
                r = tb["rows"][k]
                w1, w2 = pk[p + "w1", k], pk[p + "w2", k]
                rows1.append([_ptr(xn, k * Np), _ptr(w1), _ptr(hid, k * M * ld4N),
                              _ptr(self._flat, self._off[p + "b1"] + k * 4 * N), _ptr(hid_b, k * M * ld4N) if two else 0, K * Np, Np, ld4N, M, 4 * N, Np,
                              ld4N if two else 0])       # (f16 operands, act 1: the aux slot is the bf16 copy of the output, include/urse.h)
                rows2.append([_ptr(hid, k * M * ld4N), _ptr(w2), _ptr(pre, r[5]),
                              _ptr(self._flat, self._off[p + "b2"] + b2_off), 0, ld4N, ld4N, P, M, 4 * sb, ld4N, 0])
                b2_off += 4 * sb
            xns.append(xn_b); sts.append(st); hids.append(hid_b); pres.append(pre)
            keep += [xn, hid]                   # (f16: the forward operands, alive until both launches below are queued)
        fdt = ops.dtype_code(dt)
        nt_grouped(rows1, dev, fdt, fdt, act=1)
        nt_grouped(rows2, dev, fdt, ops.F32)
        del keep
        out = torch.empty(B, T, F, 2, dtype=torch.float32, device=dev)
        call("glu_mask_apply_fwd", pres[0], pres[1], spec, out, tb["bands"], tb["f2k"], M, F, P, stream_ptr())
        return out, ((xns, sts, hids, pres, tb) if save else None)

    def maskdec_bwd(self, skip, spec, saved, dout):
        xns, sts, hids, pres, tb = saved
        B, T, K, N = skip.shape
        F = spec.shape[2]
        dt, dev, pk, d = self.bwd_dtype, skip.device, self._packed, self._dims
        M, Np, ld4N, P = B * T, d["Np"], d["ld4N"], tb["P"]
        Kf = len(self.subbands)
        dpre = [torch.zeros(M, P, dtype=dt, device=dev) for _ in range(2)]
        call("glu_mask_apply_bwd", pres[0], pres[1], spec, dout, dpre[0], dpre[1], tb["bands"], tb["f2k"], M, F, P,
             ops._dt(dpre[0]), stream_ptr())
        dhp = [_empty_padded(K, M, ld4N, 4 * N, dt, dev) for _ in range(2)]
        dxn = [torch.empty(M * K, N, dtype=torch.float32, device=dev) for _ in range(2)]
        rows_a, rows_b = [], []
        for i, tag in enumerate("mr"):
            p = "md%s." % tag
            for k in range(K):
                r = tb["rows"][k]
                rows_a.append([_ptr(dpre[i], r[5]), _ptr(pk[p + "w2T", k]), _ptr(dhp[i], k * M * ld4N), 0,
                               _ptr(hids[i], k * M * ld4N), P, r[6], ld4N, M, 4 * N, r[6], ld4N])
                rows_b.append([_ptr(dhp[i], k * M * ld4N), _ptr(pk[p + "w1T", k]), _ptr(dxn[i], k * N), 0, 0, ld4N,
                               ld4N, K * N, M, N, ld4N, 0])
        nt_grouped(rows_a, dev, ops._dt(dpre[0]), ops._dt(dhp[0]), act=2)
        tn_rows = []
        for i, tag in enumerate("mr"):
            p = "md%s." % tag
            w2_off = b2_off = 0
            for k in range(K):
                sb = self.subbands[k]
                r = tb["rows"][k]
                tn_rows.append(ops.tn_desc(dpre[i][:, r[5]:r[5] + r[6]], hids[i][k],
                                           self._g(p + "w2", 16 * sb * N, w2_off).view(4 * sb, 4 * N),
                                           colsum=self._g(p + "b2", 4 * sb, b2_off), Mo=4 * sb, No=4 * N))
                xk = xns[i].view(M, K * Np)[:, k * Np:(k + 1) * Np]
                tn_rows.append(ops.tn_desc(dhp[i][k], xk, self._g(p + "w1", 4 * N * N, k * 4 * N * N).view(4 * N, N),
                                           colsum=self._g(p + "b1", 4 * N, k * 4 * N), Mo=4 * N, No=N))
                w2_off += 16 * sb * N
                b2_off += 4 * sb
        defer_md = ops.TN_OVERLAP and ops.DEFER_MASKDEC_WGRADS and skip.is_cuda
        if defer_md:
            # the decoder's weight gradients feed nothing but the optimizer: they join the second queue (and start beside the first BPTT, when that
            # queue is still empty) instead of taking 0.86 ms of the compute stream (step time: neutral within the run-to-run spread,
            # profiles/r04_ab_order_bias_v1.log); the closure keeps their operands alive
            keep = (dpre, hids, dhp, xns)

            def wg_md(target_wgs=0, _rows=tn_rows, _keep=keep):
                ops.gemm_tn_grouped(_rows, dt, dev)
            self._deferred.append((wg_md, "md"))
        else:
            ops.gemm_tn_grouped(tn_rows, dt, dev)     # 4 x K weight / bias gradients in one launch
        nt_grouped(rows_b, dev, ops._dt(dhp[0]), ops.F32)
        dskip = None
        for i, tag in enumerate("mr"):
            p = "md%s." % tag
            last = i == 1 and dt == torch.bfloat16 and N % 4 == 0
            dskip = ops.groupnorm_bwd(skip, dxn[i].view(B, T, K, N), sts[i], self._p(p + "gamma", Kf * N), dskip,
                                      self._g(p + "gamma", Kf * N), self._g(p + "beta", Kf * N), B, T, K, N, N, N,
                                      GN_EPS, pack_ld=Np if last else 0)
            if last:                              # the last dual-path half layer consumes this gradient next
                dskip, packed = dskip
                self._grad_pack = (dskip.data_ptr(), packed)
        if not defer_md:
            self._ready("md")
        return dskip

    # ------------------------------------------------------------------------------------------
    def forward(self, spec_ri):
        """spec_ri f32 [B, T, F, 2] -> masked spectrum f32 [B, T, F, 2] (num_spk = 1 squeezed)."""
        ops.require_cuda(spec_ri)
        self._prepare()
        if self._inflight is not None or self._deferred:
            # an aborted backward left weight-gradient GEMMs on the side stream: they still read doT / hout / dg, which
            # the closures below keep alive - wait for them before dropping the closures (the allocator would hand the
            # buffers to this forward on the compute stream otherwise)
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
        self._deferred, self._inflight, self._grad_pack = [], None, None     # nothing survives an aborted backward
        ops.CO_RESIDENT_WGS = 0       # (an aborted backward may have left the second queue's reservation set: ADVICE r3)
        ops.COMM_RESERVED_CUS = 0     # (... or RCCL's: GradBucketReducer.finish() never ran; nothing of it is in flight once a new forward starts)
        spec_ri = spec_ri.contiguous().float()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not train:
            z, _ = self.bandsplit_fwd(spec_ri, save=False)
            for l in range(self.num_layer):
                z, _ = self.dualpath_fwd(z, l, "t", False)
                z, _ = self.dualpath_fwd(z, l, "f", False)
            out = self.maskdec_fwd(z, spec_ri, False)[0]
            ops.poll_kernel_errors(spec_ri.device, sync=True)      # inference: fail now rather than return garbage
            return out
        ops.poll_kernel_errors(spec_ri.device)                     # training: deferred check of the previous steps
        anchor = self._flat.new_zeros((), requires_grad=True)
        self.mark_used_bands(self._band_tables(spec_ri.shape[2], self.compute_dtype, spec_ri.device)["K"])
        z = _BandSplitFn.apply(anchor, spec_ri, self)
        for l in range(self.num_layer):
            z = _DualPathFn.apply(z, self, l, "t")
            z = _DualPathFn.apply(z, self, l, "f")
        return _MaskDecFn.apply(z, spec_ri, self)


class _BandSplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, spec, core):
        z, saved = core.bandsplit_fwd(spec)
        ctx.core, ctx.saved, ctx.spec = core, saved, spec
        return z

    @staticmethod
    def backward(ctx, dz):
        ctx.core.bandsplit_bwd(ctx.spec, ctx.saved, dz.contiguous())
        ctx.saved = None
        return None, None, None


class _DualPathFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, core, l, path, temb=None):
        out, saved = core.dualpath_fwd(skip, l, path, True, temb)
        ctx.core, ctx.saved, ctx.skip, ctx.l, ctx.path = core, saved, skip, l, path
        return out

    @staticmethod
    def backward(ctx, dout):
        d = ctx.core.dualpath_bwd(ctx.skip, ctx.saved, ctx.l, ctx.path, dout.contiguous())
        ctx.saved = None
        return d, None, None, None, None


class _MaskDecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, spec, core):
        out, saved = core.maskdec_fwd(skip, spec, True)
        ctx.core, ctx.saved, ctx.skip, ctx.spec = core, saved, skip, spec
        return out

    @staticmethod
    def backward(ctx, dout):
        d = ctx.core.maskdec_bwd(ctx.skip, ctx.spec, ctx.saved, dout.contiguous())
        ctx.saved = None
        return d, None, None


class BSRNNSeparator(nn.Module):
    """espnet2 ``BSRNNSeparator`` surface: complex [B,T,F] -> ([complex [B,T,F]], ilens, {})."""

    def __init__(self, input_dim, num_spk=1, num_channels=16, num_layers=6, target_fs=48000, causal=True,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        self.bsrnn = BSRNNCore(input_dim, num_channels, num_layers, target_fs, causal, num_spk, compute_dtype)

    def forward(self, input, ilens=None, additional=None):
        spec_ri = torch.view_as_real(input) if input.is_complex() else input
        out = self.bsrnn(spec_ri)
        return [torch.view_as_complex(out)], ilens, {}


class _StftCfg(nn.Module):
    """parameter-free stand-in for espnet2 STFTEncoder / STFTDecoder (keeps the attribute names)."""

    def __init__(self, n_fft, hop_length, default_fs):
        super().__init__()
        self.n_fft, self.hop_length, self.default_fs = n_fft, hop_length, default_fs
        self.output_dim = n_fft // 2 + 1

    def reconfig(self, fs):
        if fs is None:
            return self.n_fft, self.hop_length
        fs = int(fs)
        return self.n_fft * fs // self.default_fs, self.hop_length * fs // self.default_fs


class BSRNN_SE(nn.Module):
    """Drop-in for ``baseline_code/models/bsrnn.py:9-41``."""

    def __init__(self, num_channel=192, num_layer=6, compute_dtype=torch.bfloat16):
        super().__init__()
        self.encoder = _StftCfg(960, 480, 48000)
        self.decoder = _StftCfg(960, 480, 48000)
        self.bsrnn = BSRNNSeparator(self.encoder.output_dim, 1, num_channel, num_layer, 48000, False, compute_dtype)

    @property
    def core(self):
        return self.bsrnn.bsrnn

    def forward(self, speech_mix, speech_lengths, fs):
        ops.require_cuda(speech_mix)
        n_fft, hop = self.encoder.reconfig(fs)
        lens = torch.as_tensor(speech_lengths)
        feature_mix = ops.stft_forward(speech_mix.float(), n_fft, hop, ops.WIN_HANN, lens)
        feature_pre, _, _ = self.bsrnn(feature_mix, None, None)
        enhanced_feature = feature_pre[0]
        n_fft_d, hop_d = self.decoder.reconfig(fs)
        enhanced_wav = ops.istft_forward(enhanced_feature, n_fft_d, hop_d, int(lens.max()))
        return enhanced_wav, enhanced_feature
