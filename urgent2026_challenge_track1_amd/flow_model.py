"""BSRNN-Flow on MI355X: the generative (flow-matching) model of the reference behind its own surface.

Mirrors ``baseline_code/models/bsrnn_flowse.py::BSRNN`` (two band splits + ``condition_fc``, 6 x dual-path BLSTM with
a Gaussian-Fourier time embedding added after ``norm_time``, ``GradDecoder`` = per-band GN + 1x1 conv + tanh ->
Conv2d(16->4, 5x5) + GLU, complex ``m * x_t + r``; :171-318), ``models/odes.py::FLOWMATCHING`` (:52-98), the white-box
Euler sampler (``sampling/__init__.py:30-65``, ``odesolvers.py:72-81``) and ``flow_model.py::FlowSEModel``
(``speech_to_feature`` :134-139, ``forward`` :203-209, ``forward_step`` :149-187, ``enhance`` :189-200, EMA :53,84,
98-112).  Parameter names follow the reference (``dnn.band_split_x...``, ``dnn.grad_decoder.conv_after_mask.0...``) so
``flow_bsrnn.ckpt`` loads by name.  Internally features are ``[B, T, F, 2]`` f32 (the reference's ``[B,1,F,T]`` complex
is converted at the surface only).
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import call, require_cuda, stream_ptr
from .bsrnn import BSRNNCore, GN_EPS, _PackPlan, _ptr, _view, _DualPathFn, nt_grouped
from .d_model import _DTYPES, StepLR

SUB_CH = 16


class _TCond(nn.Module):
    """GaussianFourierProjection(embedding_size=N/2, scale=1): fixed random frequencies (not trained)."""

    def __init__(self, half):
        super().__init__()
        self.W = nn.Parameter(torch.randn(half), requires_grad=False)


class FlowBSRNNCore(BSRNNCore):
    back_tag = "gd"
    band_groups = ("bsx", "bsy", "gdm", "gdr")

    def __init__(self, input_dim=769, num_channel=384, num_layer=6, compute_dtype=torch.bfloat16):
        # compute_dtype f16 (round 6): the FORWARD of the flow DNN with IEEE-half operands - what the Euler sampler chains 15 times; the enhanced
        # waveform then stays inside north_star's 1e-3 of the f32 oracle at bf16's speed (tests/test_c4_fullsize_gpu.py).  Training keeps bf16 / f32:
        # forward() refuses a gradient-recording pass in f16 (the flow decoder's backward has no mixed-operand form).
        super().__init__(input_dim, num_channel, num_layer, 48000, False, 1, compute_dtype)
        self._pf = None

    # ---- containers / flat layout -----------------------------------------------------------------------
    def _make_front_back(self):
        N = self.N
        self.band_split_y = self._make_band_split()
        self.band_split_x = self._make_band_split()
        self.condition_fc = nn.Linear(2 * N, N)
        self.t_cond = nn.ModuleList([_TCond(N // 2) for _ in range(self.num_layer)])
        gd = nn.Module()
        gd.conv_after_mask = nn.Sequential(nn.Conv2d(SUB_CH, 4, 5, 1, 2), nn.GLU(dim=1))
        gd.conv_after_residual = nn.Sequential(nn.Conv2d(SUB_CH, 4, 5, 1, 2), nn.GLU(dim=1))
        mk = lambda sb: nn.Sequential(nn.GroupNorm(1, N), nn.Conv1d(N, sb * SUB_CH, 1), nn.Tanh())
        gd.mlp_mask = nn.ModuleList([mk(sb) for sb in self.subbands])
        gd.mlp_residual = nn.ModuleList([mk(sb) for sb in self.subbands])
        self.grad_decoder = gd

    def _front_params(self):
        return (self._bs_params("bsx", self.band_split_x) + self._bs_params("bsy", self.band_split_y) +
                [("cfc.w", [self.condition_fc.weight]), ("cfc.b", [self.condition_fc.bias])])

    def _back_params(self):
        gd, order = self.grad_decoder, []
        for tag, mlps, conv in (("m", gd.mlp_mask, gd.conv_after_mask), ("r", gd.mlp_residual, gd.conv_after_residual)):
            p = "gd%s." % tag
            order += [(p + "gamma", [s[0].weight for s in mlps]), (p + "beta", [s[0].bias for s in mlps]),
                      (p + "w1", [s[1].weight for s in mlps]), (p + "b1", [s[1].bias for s in mlps]),
                      (p + "cw", [conv[0].weight]), (p + "cb", [conv[0].bias])]
        return order

    def _plan_front(self, pn, pt, h, dtype):
        N, Np, o = self.N, self._dims["Np"], self._off
        self._plan_band_split("bsx", pn, pt, h, dtype)
        self._plan_band_split("bsy", pn, pt, h, dtype)
        h["cfc.w"] = pn.add(o["cfc.w"], N, 2 * N, N, ops.kpad(2 * N, dtype))
        h["cfc.wT"] = pt.add(o["cfc.w"], N, 2 * N, 2 * N, Np)

    def _plan_back(self, pn, pt, h, dtype):
        """decoder 1x1 convs with output rows permuted (ch, f) -> (f, ch) so that the GEMM writes the channel-last
        [B,T,F,16] feature map the 5x5 convolution reads; biases permuted the same way (f32 plan)."""
        N, Np, o = self.N, self._dims["Np"], self._off
        self._pf, self._pf_h = _PackPlan(False), {}
        for tag in "mr":
            p = "gd%s." % tag
            w_off = b_off = 0
            for k, sb in enumerate(self.subbands):
                blk = pn.reserve(SUB_CH * sb * Np)
                bblk = self._pf.reserve(SUB_CH * sb)
                for ch in range(SUB_CH):
                    pn.add_at(o[p + "w1"] + w_off + ch * sb * N, sb, N, blk + ch * Np, sb, Np, SUB_CH * Np)
                    self._pf.add_at(o[p + "b1"] + b_off + ch * sb, sb, 1, bblk + ch, sb, 1, SUB_CH)
                h[p + "w1", k] = (blk, SUB_CH * sb, Np)
                self._pf_h[p + "b1", k] = (bblk, SUB_CH * sb, 1)
                w_off += SUB_CH * sb * N
                b_off += SUB_CH * sb

    def _prepare(self):
        stale = self._packed_version != self.param_version or self._plans is None or self._plans[0] != self.compute_dtype
        super()._prepare()
        if stale:
            bf = self._pf.run(self._flat, torch.float32)
            for key, hd in self._pf_h.items():
                self._packed[key] = _view(bf, hd).view(-1)

    # ---- forward pieces ---------------------------------------------------------------------------------------
    def front_fwd(self, x, y, save=True):
        """two band splits -> concat on channels -> condition_fc  (bsrnn_flowse.py:283-287)."""
        B, T, F, _ = x.shape
        dt, dev, N, pk = self.compute_dtype, x.device, self.N, self._packed
        K = self._band_tables(F, dt, dev)["K"]
        W2 = ops.kpad(2 * N, dt)
        cat = torch.zeros(B, T, K, W2, dtype=dt, device=dev) if W2 != 2 * N else \
            torch.empty(B, T, K, W2, dtype=dt, device=dev)
        _, sx = self.bandsplit_fwd(x, "bsx", cat, W2, 0, save=save)
        _, sy = self.bandsplit_fwd(y, "bsy", cat, W2, N, save=save)
        z = torch.empty(B, T, K, N, dtype=torch.float32, device=dev)
        ops.gemm_nt(cat.view(B * T * K, W2), pk["cfc.w"], self._p("cfc.b", N), out=z.view(B * T * K, N))
        return z, (cat, sx, sy)

    def time_embeddings(self, t):
        B, N = t.shape[0], self.N
        out = []
        for l in range(self.num_layer):
            e = torch.empty(B, N, dtype=torch.float32, device=t.device)
            call("time_embedding", t, self.t_cond[l].W, e, B, N // 2, stream_ptr())
            out.append(e)
        return out

    def graddec_fwd(self, skip, xt, sign, save):
        B, T, K, N = skip.shape
        F = xt.shape[2]
        dt, dev, pk, Np = self.compute_dtype, skip.device, self._packed, self._dims["Np"]
        tb = self._band_tables(F, dt, dev)
        Fs = sum(self.subbands[:K])
        M, Kf = B * T, len(self.subbands)
        xns, sts, Us, pres = [], [], [], []
        rows = []
        for tag in "mr":
            p = "gd%s." % tag
            xn, st = ops.groupnorm_fwd(skip, self._p(p + "gamma", Kf * N), self._p(p + "beta", Kf * N), B, T, K, N, N,
                                       Np, N, dt, GN_EPS)
            U = torch.empty(B, T, Fs, SUB_CH, dtype=torch.float32, device=dev)
            for k in range(K):
                sb, f0 = self.subbands[k], tb["rows"][k][0]
                rows.append([_ptr(xn, k * Np), _ptr(pk[p + "w1", k]), _ptr(U, f0 * SUB_CH), _ptr(pk[p + "b1", k]), 0,
                             K * Np, Np, Fs * SUB_CH, M, SUB_CH * sb, Np, 0])
            xns.append(xn); sts.append(st); Us.append(U)
        nt_grouped(rows, dev, ops._dt(xns[0]), ops.F32, act=1)
        for i, tag in enumerate("mr"):
            p = "gd%s." % tag
            pre = torch.empty(B, T, Fs, 4, dtype=torch.float32, device=dev)
            call("conv5x5_fwd", Us[i], self._p(p + "cw", 4 * SUB_CH * 25), self._p(p + "cb", 4), pre, B, T, Fs,
                 stream_ptr())
            pres.append(pre)
        out = torch.empty(B, T, F, 2, dtype=torch.float32, device=dev)
        call("glu4_apply_fwd", pres[0], pres[1], xt, out, M, F, Fs, float(sign), stream_ptr())
        return out, ((xns, sts, Us, pres, Fs) if save else None)

    # ---- backward pieces ------------------------------------------------------------------------------------------
    def front_bwd(self, x, y, saved, dz):
        cat, sx, sy = saved
        B, T, K, W2 = cat.shape
        dt, dev, N, pk, Np = self.compute_dtype, x.device, self.N, self._packed, self._dims["Np"]
        R = B * T * K
        dzT = ops.pack2d(dz.reshape(R, N), R, Np, dt)
        ops.gemm_tn(dzT, cat.view(R, W2), self._g("cfc.w", N * 2 * N).view(N, 2 * N), colsum=self._g("cfc.b", N), Mo=N,
                    No=2 * N)
        dcat = torch.zeros(R * W2 + 64, dtype=dt, device=dev)[:R * W2].view(R, W2)   # slack: band slices read kpad(N) wide
        ops.gemm_nt(dzT, pk["cfc.wT"], out=dcat, N=2 * N)
        self.bandsplit_bwd(x, sx, None, "bsx", dcat, W2, 0, ready=False)
        self.bandsplit_bwd(y, sy, None, "bsy", dcat, W2, N, ready=True)

    def graddec_bwd(self, skip, xt, saved, dout, sign):
        xns, sts, Us, pres, Fs = saved
        B, T, K, N = skip.shape
        F = xt.shape[2]
        dt, dev, pk, Np = self.compute_dtype, skip.device, self._packed, self._dims["Np"]
        tb = self._band_tables(F, dt, dev)
        M, Kf = B * T, len(self.subbands)
        dpre = [torch.empty(B, T, Fs, 4, dtype=torch.float32, device=dev) for _ in range(2)]
        call("glu4_apply_bwd", pres[0], pres[1], xt, dout, dpre[0], dpre[1], M, F, Fs, float(sign), stream_ptr())
        ldu = ops.kpad(Fs * SUB_CH, dt)
        dxn = [torch.empty(M * K, N, dtype=torch.float32, device=dev) for _ in range(2)]
        rows = []
        keep = []
        for i, tag in enumerate("mr"):
            p = "gd%s." % tag
            dU = torch.empty(B, T, Fs, SUB_CH, dtype=torch.float32, device=dev)
            call("conv5x5_bwd", Us[i], self._p(p + "cw", 4 * SUB_CH * 25), dpre[i], dU, self._g(p + "cw", 4 * SUB_CH * 25),
                 self._g(p + "cb", 4), B, T, Fs, stream_ptr())
            dUp = torch.zeros(M, ldu, dtype=dt, device=dev)
            call("tanh_bwd_pack", dU, Us[i], dUp, M, Fs * SUB_CH, ldu, ops._dt(dUp), stream_ptr())
            keep.append(dUp)
            w_off = b_off = 0
            for k in range(K):
                sb, f0 = self.subbands[k], tb["rows"][k][0]
                n_out = SUB_CH * sb
                a = dUp[:, f0 * SUB_CH:f0 * SUB_CH + n_out]
                xk = xns[i].view(M, K * Np)[:, k * Np:(k + 1) * Np]
                ops.gemm_tn(a, xk, self._g(p + "w1", n_out * N, w_off).view(n_out, N),
                            colsum=self._g(p + "b1", n_out, b_off), Mo=n_out, No=N, perm_h=-sb)
                kp = ops.kpad(n_out, dt)
                w1T = ops.pack2d(pk[p + "w1", k], N, kp, dt, transpose=True)       # [N, kpad(16 sb)]
                keep.append(w1T)
                rows.append([_ptr(dUp, f0 * SUB_CH), _ptr(w1T), _ptr(dxn[i], k * N), 0, 0, ldu, kp, K * N, M, N, kp, 0])
                w_off += n_out * N
                b_off += n_out
        nt_grouped(rows, dev, ops._dt(keep[0]), ops.F32)
        dskip = None
        for i, tag in enumerate("mr"):
            p = "gd%s." % tag
            dskip = ops.groupnorm_bwd(skip, dxn[i].view(B, T, K, N), sts[i], self._p(p + "gamma", Kf * N), dskip,
                                      self._g(p + "gamma", Kf * N), self._g(p + "beta", Kf * N), B, T, K, N, N, N,
                                      GN_EPS)
        self._ready("gd")
        return dskip

    def forward(self, x_ri, y_ri, t, sign=1.0):
        """x_ri (= x_t), y_ri: f32 [B,T,F,2]; t f32 [B] -> sign * (m * x_t + r) as f32 [B,T,F,2]."""
        require_cuda(x_ri, y_ri, t)
        self._prepare()
        x_ri, y_ri, t = x_ri.contiguous().float(), y_ri.contiguous().float(), t.contiguous().float()
        tembs = self.time_embeddings(t)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        ops.poll_kernel_errors(x_ri.device)      # deferred check (the sampler calls this 15 times: no host stall per call)
        if train:
            if self.compute_dtype == torch.float16:
                raise NotImplementedError("compute_dtype f16 is the flow DNN's INFERENCE arithmetic (sampler / enhance); train it in bf16 or f32")
            anchor = self._flat.new_zeros((), requires_grad=True)
            self.mark_used_bands(self._band_tables(x_ri.shape[2], self.compute_dtype, x_ri.device)["K"])
            z = _FlowFrontFn.apply(anchor, x_ri, y_ri, self)
            for l in range(self.num_layer):
                z = _DualPathFn.apply(z, self, l, "t", tembs[l])
                z = _DualPathFn.apply(z, self, l, "f")
            return _GradDecFn.apply(z, x_ri, self, float(sign))
        z, _ = self.front_fwd(x_ri, y_ri, save=False)
        for l in range(self.num_layer):
            z, _ = self.dualpath_fwd(z, l, "t", False, tembs[l])
            z, _ = self.dualpath_fwd(z, l, "f", False)
        return self.graddec_fwd(z, x_ri, sign, False)[0]


class _FlowFrontFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, y, core):
        z, saved = core.front_fwd(x, y)
        ctx.core, ctx.saved, ctx.x, ctx.y = core, saved, x, y
        return z

    @staticmethod
    def backward(ctx, dz):
        ctx.core.front_bwd(ctx.x, ctx.y, ctx.saved, dz.contiguous())
        ctx.saved = None
        return None, None, None, None


class _GradDecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, xt, core, sign):
        out, saved = core.graddec_fwd(skip, xt, sign, True)
        ctx.core, ctx.saved, ctx.skip, ctx.xt, ctx.sign = core, saved, skip, xt, sign
        return out

    @staticmethod
    def backward(ctx, dout):
        d = ctx.core.graddec_bwd(ctx.skip, ctx.xt, ctx.saved, dout.contiguous(), ctx.sign)
        ctx.saved = None
        return d, None, None, None


class _FlowLossFn(torch.autograd.Function):
    """mean_B( 0.5 * sum |vf - cvf|^2 )  (flow_model.py:122-132, loss_type 'mse')."""

    @staticmethod
    def forward(ctx, vf, cvf):
        B = vf.shape[0]
        per_b = vf[0].numel() // 2
        loss = torch.empty(B, dtype=torch.float64, device=vf.device)
        grad = torch.empty_like(vf)
        call("flow_loss", vf, cvf, loss, grad, B, per_b, 1.0 / B, stream_ptr())
        ctx.grad = grad
        return loss.mean().float()

    @staticmethod
    def backward(ctx, g):
        out = ctx.grad
        ctx.grad = None
        # the upstream scale (1 in the training loop) is applied on the device: no host sync inside the step
        call("scale_by_device_scalar", out.view(-1), g.reshape(1).float().contiguous(), out.numel(), stream_ptr())
        return out, None


class FlowEMA:
    """torch_ema.ExponentialMovingAverage(parameters, decay, use_num_updates=True) on the flat buffer."""

    def __init__(self, core, decay):
        self.core, self.decay, self.num_updates = core, decay, 0
        self.shadow = core.flat_params.clone()
        self.collected = None

    def _home(self):
        """the shadow lives where the flat parameters live: a model moved with ``.to(device)`` after the EMA was created (or loaded
        from a checkpoint on the host) takes its shadow along instead of handing a host pointer to the kernel."""
        flat = self.core.flat_params
        if self.shadow.device != flat.device:
            self.shadow = self.shadow.to(flat.device)
            if self.collected is not None:
                self.collected = self.collected.to(flat.device)
        return flat

    def update(self):
        flat = self._home()
        require_cuda(self.shadow)
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        call("ema_update", self.shadow, flat, float(1.0 - d), self.shadow.numel(), stream_ptr())

    def store(self):
        self.collected = self._home().clone()

    def copy_to(self):
        self._home()
        self.core.flat_params.copy_(self.shadow)
        self.core.param_version += 1

    def restore(self):
        if self.collected is not None:
            self._home()
            self.core.flat_params.copy_(self.collected)
            self.core.param_version += 1
            self.collected = None

    def state_dict(self, model=None):
        """torch_ema layout (``shadow_params`` = one tensor per parameter of ``model.parameters()``, in that order; that is
        what the reference stores under checkpoint['ema'], flow_model.py:96) plus our flat copy."""
        sd = {"decay": self.decay, "num_updates": self.num_updates, "shadow_flat": self.shadow.detach().cpu(),
              "collected_params": None}
        if model is not None:
            sd["shadow_params"] = [(p.detach() if o is None else self.shadow[o:o + p.numel()].view(p.shape)).cpu().clone()
                                   for p, o in self._slices(model)]
        return sd

    def _slices(self, model):
        """(parameter, offset of its shadow in the flat buffer | None for the frozen time-embedding frequencies)."""
        flat = self.core.flat_params
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        return [(p, (p.data_ptr() - lo) // 4 if lo <= p.data_ptr() < hi else None) for p in model.parameters()]

    def load_state_dict(self, sd, model=None):
        self.decay, self.num_updates = sd["decay"], sd["num_updates"]
        self._home()
        if "shadow_flat" in sd:
            self.shadow.copy_(sd["shadow_flat"])
        elif model is not None and "shadow_params" in sd:       # a checkpoint written by torch_ema itself
            for (p, o), s in zip(self._slices(model), sd["shadow_params"]):
                if o is not None:
                    self.shadow[o:o + p.numel()].copy_(s.reshape(-1))
        else:
            raise KeyError("EMA state without shadow parameters")


class FlowSEModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        g = lambda k, d: getattr(cfg, k, d)
        self.n_fft, self.hop = g("n_fft", 1536), g("hop_length", 384)
        self.spec_e, self.spec_factor = g("spec_abs_exponent", 0.667), g("spec_factor", 0.065)
        self.sigma_min, self.sigma_max = g("sigma_min", 0.05), g("sigma_max", 0.5)
        self.t_eps, self.T_rev = g("t_eps", 0.03), g("T_rev", 1.0)
        self.loss_type = g("loss_type", "mse")
        dtype = _DTYPES[str(g("compute_dtype", "bf16"))]
        self.dnn = FlowBSRNNCore(self.n_fft // 2 + 1, g("bsrnn_hidden", 384), g("num_layer", 6), dtype)
        self.ema_decay = g("ema_decay", 0.999)
        self.ema = None
        self._error_loading_ema = False
        self.logged = {}

    def log(self, name, value, **_):
        self.logged[name] = value

    # ---- EMA weights for evaluation (flow_model.py:98-113): eval() swaps them in, train() restores ---------------------
    def train(self, mode=True, no_ema=False):
        res = super().train(mode)
        if self.ema is not None and not self._error_loading_ema:
            if not mode and not no_ema:
                if self.ema.collected is None:      # (the reference would overwrite its stored copy on a second eval())
                    self.ema.store()
                self.ema.copy_to()
            elif self.ema.collected is not None:
                self.ema.restore()
        return res

    def eval(self, no_ema=False):
        return self.train(False, no_ema=no_ema)

    def on_save_checkpoint(self, checkpoint):
        if self.ema is None:
            self.init_ema()
        checkpoint["ema"] = self.ema.state_dict(self)

    def on_load_checkpoint(self, checkpoint):
        ema = checkpoint.get("ema", None)
        if ema is None:
            self._error_loading_ema = True
            import warnings
            warnings.warn("EMA state_dict not found in checkpoint!")
            return
        if self.ema is None:
            self.init_ema()
        self.ema.load_state_dict(ema, self)

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cuda"):
        """Lightning's ``FlowSEModel.load_from_checkpoint`` as inference.py:33 uses it."""
        from .config import Config
        ck = torch.load(path, map_location="cpu", weights_only=False)
        cfg = ck.get("hyper_parameters", {}).get("cfg", None)
        cfg = Config(**vars(cfg)) if cfg is not None and not isinstance(cfg, Config) else (cfg or Config())
        model = cls(cfg)
        sd = ck["state_dict"] if "state_dict" in ck else ck
        if not any(k.startswith("dnn.") for k in sd):
            raise KeyError("not a FlowSEModel checkpoint (no dnn.* parameters): %s" % path)
        model.load_state_dict({k: v for k, v in sd.items() if k.startswith("dnn.")})
        model = model.to(map_location)
        model.on_load_checkpoint(ck)
        return model

    # ---- features ----------------------------------------------------------------------------------------------
    def _stft_cfg(self, fs):
        if fs is None:
            return self.n_fft, self.hop
        fs = int(fs)
        return self.n_fft * fs // 48000, self.hop * fs // 48000

    def speech_to_feature_ri(self, speech, fs, speech_length):
        n_fft, hop = self._stft_cfg(fs)
        spec = ops.stft_forward(speech.float(), n_fft, hop, ops.WIN_HANN, torch.as_tensor(speech_length))
        ri = torch.view_as_real(spec).contiguous()
        out = torch.empty_like(ri)
        call("spec_transform", ri, out, ri.numel() // 2, float(self.spec_e), float(self.spec_factor), 0, stream_ptr())
        return out                                                   # [B,T,F,2]

    def feature_ri_to_speech(self, feat_ri, fs, speech_length):
        n_fft, hop = self._stft_cfg(fs)
        back = torch.empty_like(feat_ri)
        call("spec_transform", feat_ri.contiguous(), back, feat_ri.numel() // 2, float(self.spec_e),
             float(self.spec_factor), 1, stream_ptr())
        return ops.istft_forward(back, n_fft, hop, int(torch.as_tensor(speech_length).max()))

    def speech_to_feature(self, speech, fs, speech_length):
        """reference layout: complex [B,1,F,T] (flow_model.py:134-139)."""
        return torch.view_as_complex(self.speech_to_feature_ri(speech, fs, speech_length)).permute(0, 2, 1).unsqueeze(1)

    def feature_to_speech(self, feature, fs, speech_length):
        ri = torch.view_as_real(feature.squeeze(1).permute(0, 2, 1).contiguous())
        return self.feature_ri_to_speech(ri, fs, speech_length)

    # ---- vector field ----------------------------------------------------------------------------------------------
    def vector_field_ri(self, xt_ri, t, y_ri):
        """FlowSEModel.forward (:203-209): -dnn(cat[x, y], t), on [B,T,F,2] tensors."""
        return self.dnn(xt_ri, y_ri, t, sign=-1.0)

    def forward(self, x, t, y):
        to_ri = lambda c: torch.view_as_real(c.squeeze(1).permute(0, 2, 1).contiguous())
        out = self.vector_field_ri(to_ri(x), t, to_ri(y))
        return torch.view_as_complex(out).permute(0, 2, 1).unsqueeze(1)

    def prior_sample_ri(self, Y_ri, z_ri=None):
        """ode.prior_sampling (odes.py:84-91): x_T = Y + sigma(1) z, z ~ CN(0, 1)."""
        if z_ri is None:
            z_ri = torch.view_as_real(torch.randn(Y_ri.shape[:-1], dtype=torch.complex64, device=Y_ri.device))
        B = Y_ri.shape[0]
        xt = torch.empty_like(Y_ri)
        ones = torch.ones(B, device=Y_ri.device)
        call("flow_prepare", Y_ri, Y_ri, z_ri.contiguous(), ones, xt, None, B, Y_ri[0].numel() // 2,
             float(self.sigma_min), float(self.sigma_max), stream_ptr())
        return xt

    def sample_ri(self, Y_ri, N=15, z_ri=None):
        """white-box Euler solver (sampling/__init__.py:30-65): x <- x - step_i * VF(x, t_i, Y)."""
        with torch.no_grad():
            xt = self.prior_sample_ri(Y_ri, z_ri)
            ts = torch.linspace(self.T_rev, self.t_eps, N)
            B = Y_ri.shape[0]
            for i in range(N):
                step = float(ts[i] - ts[i + 1]) if i != N - 1 else float(ts[-1])
                vec_t = torch.full((B,), float(ts[i]), device=Y_ri.device)
                vf = self.vector_field_ri(xt, vec_t, Y_ri)
                call("axpy", vf, xt, -step, xt.numel(), stream_ptr())
            ops.poll_kernel_errors(Y_ri.device, sync=True)     # end of the trajectory: fail rather than return garbage
            return xt

    def enhance(self, y, fs, speech_length, N=15):
        """flow_model.py:189-200."""
        Y = self.speech_to_feature_ri(y, fs, speech_length)
        return self.feature_ri_to_speech(self.sample_ri(Y, N), fs, speech_length)

    # ---- training-side pieces that exist so far ---------------------------------------------------------------------
    def loss_from_ri(self, x0_ri, y_ri, t, z_ri):
        """forward_step :164-172 / _loss :122-132 ('mse') for given t and noise (forward value, no gradient yet)."""
        B = x0_ri.shape[0]
        per_b = x0_ri[0].numel() // 2
        xt, cvf = torch.empty_like(x0_ri), torch.empty_like(x0_ri)
        call("flow_prepare", x0_ri.contiguous(), y_ri.contiguous(), z_ri.contiguous(), t.contiguous().float(), xt, cvf, B,
             per_b, float(self.sigma_min), float(self.sigma_max), stream_ptr())
        with torch.no_grad():
            vf = self.vector_field_ri(xt, t, y_ri)
        loss = torch.empty(B, dtype=torch.float64, device=x0_ri.device)
        call("flow_loss", vf, cvf, loss, None, B, per_b, 1.0, stream_ptr())
        return loss

    def init_ema(self):
        self.ema = FlowEMA(self.dnn, self.ema_decay)
        return self.ema

    def forward_step(self, batch, t=None, z_ri=None):
        """flow_model.py:149-187.  t ~ min((1-U)(T_rev - t_eps) + t_eps, T_rev) and z ~ CN(0,1) unless given."""
        clean_speech, noisy_speech, fs, speech_length = batch
        B, C, T = clean_speech.shape
        assert C == 1
        clean_speech, noisy_speech = (w.reshape(B, T).float().contiguous() for w in (clean_speech, noisy_speech))
        for w in (clean_speech, noisy_speech):          # torch.nan_to_num(., nan=0) (:156-157)
            call("nan_to_num", w, w, w.numel(), stream_ptr())
        x0 = self.speech_to_feature_ri(clean_speech, fs, speech_length)
        y = self.speech_to_feature_ri(noisy_speech, fs, speech_length)
        dev = x0.device
        if t is None:
            t = torch.clamp((1 - torch.rand(B, device=dev)) * (self.T_rev - self.t_eps) + self.t_eps, max=self.T_rev)
        if z_ri is None:
            z_ri = torch.view_as_real(torch.randn(x0.shape[:-1], dtype=torch.complex64, device=dev))
        xt, cvf = torch.empty_like(x0), torch.empty_like(x0)
        call("flow_prepare", x0, y, z_ri.contiguous(), t.contiguous().float(), xt, cvf, B, x0[0].numel() // 2,
             float(self.sigma_min), float(self.sigma_max), stream_ptr())
        vf = self.vector_field_ri(xt, t, y)
        return _FlowLossFn.apply(vf, cvf)

    def training_step(self, batch, batch_idx=0):
        loss = self.forward_step(batch)
        self.log("train_loss", loss.detach())
        return loss

    def validation_step(self, batch, batch_idx=0):
        """flow_model.py:213-231: the flow loss on the batch, and on batch 0 a 10-step Euler enhancement scored by SI-SNR."""
        with torch.no_grad():
            loss = self.forward_step(batch)
            self.log("val_loss", loss.detach())
            if batch_idx == 0:
                clean_speech, noisy_speech, fs, speech_length = batch
                B, C, T = clean_speech.shape
                predicted = self.enhance(noisy_speech.reshape(B, T).float(), fs, speech_length, N=10)
                self.log("sisnr", -ops.si_snr_loss(clean_speech.reshape(B, T).float(), predicted).mean())
        return {"val_loss": loss.detach(), "loss": loss.detach()}

    def configure_optimizers(self):
        core = self.dnn
        opt = ops.FusedClipAdamW(core.flat_params, core.flat_grads, lr=getattr(self.cfg, "learning_rate", 1e-4),
                                 eps=getattr(self.cfg, "adam_epsilon", 1e-8),
                                 weight_decay=getattr(self.cfg, "weight_decay", 1e-6),
                                 max_norm=getattr(self.cfg, "gradient_clip", 0.5), core=core)
        return [opt], [StepLR(opt, getattr(self.cfg, "lr_step_size", 1), getattr(self.cfg, "lr_gamma", 0.85))]

    def optimizer_step(self, optimizer, reducer=None):
        """clip + AdamW, then ema.update (flow_model.py:69-84)."""
        scale = reducer.finish() if reducer is not None else 1.0
        optimizer.step(grad_scale=scale, zero_grad=True)
        self.dnn.param_version += 1
        if self.ema is None:
            self.init_ema()
        self.ema.update()
