"""SEModel: the discriminative training task, same surface as ``baseline_code/d_model.py:13-113``
(``SEModel(cfg)``, ``.se_model``, ``.forward_step(batch, stage)``, ``.training_step``,
``.validation_step``, ``.configure_optimizers``) without Lightning: the loop lives in ``train_se.py``.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .bsrnn import BSRNN_SE
from .config import Config

# "f16" = IEEE-half operands in the forward contractions (bf16's bytes and MFMA rate, 11 significant bits: the enhanced waveform then meets
# north_star's 1e-3 against the f32 reference arithmetic), bf16 operands in the backward (bsrnn.BSRNNCore.__init__)
_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "float32": torch.float32,
           "f16": torch.float16, "float16": torch.float16, "f16fwd": torch.float16}


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) on the fused optimizer (d_model.py:110-111)."""

    def __init__(self, optimizer, step_size, gamma):
        self.opt, self.step_size, self.gamma = optimizer, step_size, gamma
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.opt.lr = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lr = sd["last_epoch"], sd["base_lr"]
        self.opt.lr = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)


class SEModel(nn.Module):
    def __init__(self, cfg: Config):
        super().__init__()
        self.cfg = cfg
        if cfg.se_model != "bsrnn":
            raise TypeError(cfg.se_model)
        dtype = _DTYPES[str(getattr(cfg, "compute_dtype", "bf16"))]
        self.se_model = BSRNN_SE(**(cfg.model_configs or {}), compute_dtype=dtype)
        # MultiResL1SpecLoss(window_sz=[256,512,768,1024], eps=1e-6, normalize_variance=True, time_domain_weight=.5)
        self.mr_l1_windows, self.mr_l1_eps, self.mr_l1_td_weight = (256, 512, 768, 1024), 1.0e-6, 0.5
        self.logged = {}

    def log(self, name, value, **_):
        self.logged[name] = value

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cuda"):
        """Lightning's ``SEModel.load_from_checkpoint(ckpt, map_location=...)`` as inference.py:31 calls it: a dict with
        ``state_dict`` (``se_model.*`` names) and ``hyper_parameters['cfg']``; raises on a checkpoint of another model."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        cfg = ck.get("hyper_parameters", {}).get("cfg", None) if isinstance(ck, dict) else None
        if cfg is None:
            cfg = Config(model_configs={"num_channel": 196, "num_layer": 6})
        elif not isinstance(cfg, Config):   # a Lightning checkpoint of the reference pickles its own Config class
            cfg = Config(**dict(vars(cfg)))
        sd = ck["state_dict"] if "state_dict" in ck else ck
        if not any(k.startswith("se_model.") or k.startswith("bsrnn.") for k in sd):
            raise KeyError("not an SEModel checkpoint (no se_model.* parameters): %s" % path)
        model = cls(cfg)
        model.se_model.load_state_dict({k[len("se_model."):] if k.startswith("se_model.") else k: v for k, v in sd.items()})
        return model.to(map_location)

    def forward_step(self, batch, stage="train"):
        clean_speech, noisy_speech, fs, speech_length = batch
        B, C, T = clean_speech.shape
        assert C == 1
        clean_speech = clean_speech.view(B, T).float()
        noisy_speech = noisy_speech.view(B, T).float()
        se_speech = self.se_model(noisy_speech, speech_length, fs)[0]
        # (nan_guard: a NaN batch loss trains on zero gradients, d_model.py:75-77, decided on the device)
        loss = ops.mr_l1_loss(clean_speech, se_speech, self.mr_l1_windows, self.mr_l1_eps,
                              self.mr_l1_td_weight, nan_guard=True).mean()
        with torch.no_grad():
            sisnr_loss = ops.si_snr_loss(clean_speech, se_speech).mean()
        # device scalars: no host sync inside the step (the reference .item()s here, d_model.py:82-87)
        self.log("%s_loss" % stage, loss.detach())
        self.log("%s_sisnr" % stage, -sisnr_loss)
        self.log("%s_sisnr_%s" % (stage, int(fs)), -sisnr_loss)
        return loss

    def training_step(self, batch, batch_idx=0):
        return self.forward_step(batch)

    def validation_step(self, batch, batch_idx=0):
        with torch.no_grad():
            return {"loss": self.forward_step(batch, stage="val").detach()}

    def configure_optimizers(self):
        core = self.se_model.core
        opt = ops.FusedClipAdamW(core.flat_params, core.flat_grads, lr=self.cfg.learning_rate,
                                 eps=self.cfg.adam_epsilon, weight_decay=self.cfg.weight_decay,
                                 max_norm=self.cfg.gradient_clip, core=core)
        sched = StepLR(opt, self.cfg.lr_step_size, self.cfg.lr_gamma)
        return [opt], [sched]

    def optimizer_step(self, optimizer, reducer=None):
        """clip (train_se.py:78) + AdamW on the averaged gradients; NaN gradients skip the update in-kernel."""
        scale = reducer.finish() if reducer is not None else 1.0
        optimizer.step(grad_scale=scale, zero_grad=True)
        self.se_model.core.param_version += 1
        self.log("Grad_norm", optimizer.grad_norm() * scale)
