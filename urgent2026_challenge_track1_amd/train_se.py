#!/usr/bin/env python
"""Training entry point: same CLI / yaml / checkpoint layout as ``baseline_code/train_se.py:37-84`` with an own
loop instead of ``L.Trainer`` (one process per GPU, launched by ``python -m torch.distributed.run``; RANK /
LOCAL_RANK / WORLD_SIZE from the environment).

Kept from the reference: Config + yaml overlay (:41-43), seed (:45), ``init_from`` warm start accepting a raw or
``{'state_dict': ...}`` file (:55-60), checkpoint directory ``exp/{tag}/{name}/version_{v}/checkpoints`` and file name
``best_epoch=EE-step=SSSSSS-val_loss=X.XXX.ckpt`` with top-k by val_loss every ``val_check_interval`` steps (:17-35),
resume from the newest ``*-val_loss*.ckpt`` (:67-72), clip ``gradient_clip`` (:78), StepLR per epoch.  Checkpoints are
Lightning-shaped dicts (``state_dict`` with the ``se_model.`` prefix, ``hyper_parameters['cfg']``, ``epoch``,
``global_step``) so the reference's ``inference.py`` conventions and ours interoperate.
"""
import glob
import os
import time

import torch
import torch.distributed as dist

from . import _lib, ops
from .config import Config, config_parser
from .d_model import SEModel
from .dataset import AudioDataModule, RawMixBatch
from .ddp import GradBucketReducer
from .flow_model import FlowSEModel


def ckpt_dir(cfg):
    return "./exp/%s/%s/version_%s/checkpoints" % (cfg.train_tag, cfg.train_name, cfg.train_version)


def build_model(cfg):
    """model select of train_se.py:50-53."""
    return FlowSEModel(cfg) if getattr(cfg, "model_type", "discriminative") == "flowse" else SEModel(cfg)


def core_of(model):
    return model.dnn if isinstance(model, FlowSEModel) else model.se_model.core


def model_state(model):
    """Lightning-shaped parameter names: ``se_model.*`` (SEModel) / ``dnn.*`` (FlowSEModel)."""
    if isinstance(model, FlowSEModel):
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return {"se_model." + k: v.detach().cpu().clone() for k, v in model.se_model.state_dict().items()}


def save_checkpoint(path, model, opt, sched, epoch, step, val_loss):
    # a cooperative LSTM kernel that timed out leaves garbage gradients (the optimizer skipped that update): never
    # write a checkpoint past an unexamined error
    ops.poll_kernel_errors(core_of(model).flat_params.device, sync=True)
    ck = {"state_dict": model_state(model), "hyper_parameters": {"cfg": model.cfg}, "epoch": epoch, "global_step": step,
          "val_loss": val_loss, "optimizer_states": [{k: (v.cpu() if torch.is_tensor(v) else v)
                                                      for k, v in opt.state_dict().items()}],
          "lr_schedulers": [sched.state_dict()], "urse_version": 1}
    if hasattr(model, "on_save_checkpoint"):
        model.on_save_checkpoint(ck)               # FlowSEModel: checkpoint['ema'] (flow_model.py:95-96)
    torch.save(ck, path)


def load_model_state(model, state_dict, with_ema=True):
    """weights of a checkpoint into the model.  with_ema=False is `init_from` (reference train_se.py:55-59: a plain
    ``model.load_state_dict`` - Lightning's on_load_checkpoint hook does not run there); with_ema=True is the resume path
    (Lightning restores checkpoint['ema'] through the hook, flow_model.py:98-113).
    Deviation, stated (DESIGN.md quirk list): the reference builds ExponentialMovingAverage(self.parameters()) in
    FlowSEModel.__init__, BEFORE train_se.py loads init_from, so its shadow starts from the random initialisation; here the EMA is
    created lazily after the load and starts from the init_from weights.  The decay warm-up (1 + n) / (10 + n) forgets the
    difference within a few hundred updates; EMA validation figures of the first steps differ from the reference's."""
    ck = state_dict
    if "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
    if isinstance(model, FlowSEModel):
        model.load_state_dict({k: v for k, v in state_dict.items() if k.startswith("dnn.")})
        if with_ema and "ema" in ck:
            model.on_load_checkpoint(ck)
        return
    sd = {k[len("se_model."):] if k.startswith("se_model.") else k: v for k, v in state_dict.items()}
    model.se_model.load_state_dict(sd)


def to_device(batch, dev, skipped=None):
    """host batch -> device batch; a dynamic-mixing batch is SIMULATED here, on the device (mixing.simulate_recipes)."""
    if isinstance(batch, RawMixBatch):
        return batch.materialise(dev, skipped)
    clean, noisy, fs, lens = batch
    return clean.to(dev, non_blocking=True), noisy.to(dev, non_blocking=True), fs, lens


class DevicePrefetcher:
    """Iterates a loader one batch ahead on a side stream: the host→device copies of batch i+1 and, for a dynamic-mixing batch,
    its simulator kernels run beside the train step of batch i (whose recurrences leave most issue slots idle) instead of in front
    of step i+1.  The consumer's stream waits on the batch's event; the tensors are handed over with ``record_stream``."""

    def __init__(self, loader, dev, skipped=None, depth=None):
        """``depth`` batches are staged ahead (2: the simulation of batch i + 2 may start as soon as step i - 1 is done, so a simulation
        that the low stream priority stretches past one step does not stall the next one)."""
        depth = int(os.environ.get("URSE_PREFETCH_DEPTH", "2")) if depth is None else int(depth)
        self.loader, self.dev, self.skipped, self.depth = loader, dev, skipped, max(1, depth)
        # lowest HIP priority: the simulator's thousands of small workgroups only take CUs the train step leaves idle, so they cannot
        # keep a workgroup of a cooperative recurrence kernel (which spins on its peers) off its CU
        self.stream = ops.low_priority_stream(dev, force=True) if torch.device(dev).type == "cuda" else None

    def _stage(self, batch):
        if isinstance(batch, RawMixBatch):
            ops.PREFETCH_RESERVED_CUS = 8          # simulator kernels beside the step: the cooperative forward leaves them CUs (ops.py)
        if self.stream is None:
            return to_device(batch, self.dev, self.skipped), None
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # buffers freed by the consumer are safe to reuse
        with torch.cuda.stream(self.stream):
            out = to_device(batch, self.dev, self.skipped)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        import collections
        it = iter(self.loader)
        staged, done = collections.deque(), False

        def fill():
            nonlocal done
            while not done and len(staged) < self.depth:
                try:
                    staged.append(self._stage(next(it)))
                except StopIteration:
                    done = True
        try:
            fill()
            while staged:
                out, ev = staged.popleft()
                fill()
                if ev is not None:
                    cur = torch.cuda.current_stream(self.dev)
                    cur.wait_event(ev)
                    for t in out:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                yield out
        finally:
            # the reservation belongs to this iteration: validation / inference / metrics later in the process plan their cooperative
            # kernels on the whole chip again
            ops.PREFETCH_RESERVED_CUS = 0


def validate(model, loader, dev):
    tot, n = 0.0, 0
    model.eval()
    for i, batch in enumerate(loader):
        tot += float(model.validation_step(to_device(batch, dev), i)["loss"])
        n += 1
    model.train()
    return tot / max(n, 1)


def equalise_batch_counts(sampler, world, dev):
    """GroupedBatchSampler gives ranks different batch counts when a per-fs group does not divide evenly; every rank
    must run the same number of optimisation steps per epoch (same LR schedule, paired all-reduces, a clean end), so all
    ranks truncate to the minimum.  (The reference inherits the unevenness and hangs in DDP.)"""
    sampler.max_batches = None
    n = torch.tensor([len(sampler)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(n, op=dist.ReduceOp.MIN)
    sampler.max_batches = int(n.item())
    return sampler.max_batches


def raise_kernel_errors_on_all_ranks(dev, world):
    """the cooperative kernels' error flag, OR-ed over the ranks: a timed-out hand-off on one rank raises on ALL of them (rank 0
    alone reading it in save_checkpoint would leave the others waiting in the next all-reduce)."""
    flag = ops.kernel_error_flag(dev).to(torch.int32).clone()
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) != 0:
        raise _lib.UrseError("a cooperative LSTM kernel timed out on some rank: the last optimisation steps are invalid")


def fit(cfg, max_steps=None, log_every=50):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("URSE_DIST_BACKEND", "nccl") != "nccl":
        ndev = max(1, torch.cuda.device_count())
        # ranks per device ON THIS NODE (the global world size made a 2 x 8 job look shared): cooperative grids of two processes on one
        # GPU are not planned (ops.lstm_nsplit_plan, ops.lstm_cluster_plan, ...)
        ops.SHARED_GPU_RANKS = -(-int(os.environ.get("LOCAL_WORLD_SIZE", world)) // ndev)
        local %= ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("URSE_DIST_BACKEND", "nccl")     # nccl = RCCL; gloo lets ranks share one GPU (tests)
        if backend == "nccl":
            ops.cap_rccl_channels()         # bounds what an all-reduce kernel occupies: ops.reserved_cus() leaves it that many CUs
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(cfg.seed)

    model = build_model(cfg)
    if cfg.init_from != "none":
        load_model_state(model, torch.load(cfg.init_from, map_location="cpu", weights_only=False), with_ema=False)
        if rank == 0:
            print("Init param loaded from %s" % cfg.init_from)
    model = model.to(dev)
    core = core_of(model)
    (opt,), (sched,) = model.configure_optimizers()
    epoch0, step = 0, 0
    os.makedirs(ckpt_dir(cfg), exist_ok=True)
    ckpts = sorted(glob.glob(ckpt_dir(cfg) + "/*-val_loss*.ckpt"), key=os.path.getmtime, reverse=True)
    if cfg.resume and ckpts:
        ck = torch.load(ckpts[0], map_location="cpu", weights_only=False)
        load_model_state(model, ck)
        opt.load_state_dict({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ck["optimizer_states"][0].items()})
        sched.load_state_dict(ck["lr_schedulers"][0])
        epoch0, step = ck["epoch"], ck["global_step"]
        core.param_version += 1
        if rank == 0:
            print("Resume from %s" % ckpts[0])
    if world > 1:
        dist.broadcast(core.flat_params, 0)
        core.param_version += 1
    if isinstance(model, FlowSEModel) and model.ema is None:
        model.init_ema()                       # (after the broadcast: the EMA starts from the shared weights)
    reducer = GradBucketReducer(core) if world > 1 else None

    dm = AudioDataModule(cfg, rank, world)
    train_loader, val_loader = dm.train_dataloader(), dm.val_dataloader()
    best = []   # [(val_loss, path)]
    t0 = time.time()
    skipped = {}                                # augmentations the recipes drew but the device path cannot apply
    policy, warned = getattr(cfg, "unsupported_augmentation", "warn"), False
    for epoch in range(epoch0, cfg.num_train_epochs):
        # the reference never advances the sampler epoch (quirk C.3: on_train_epoch_start is not a DataModule hook)
        n_batches, in_epoch = equalise_batch_counts(dm.train_batch_sampler, world, dev), 0
        for batch in DevicePrefetcher(train_loader, dev, skipped):
            loss = model.training_step(batch)
            loss.backward()
            model.optimizer_step(opt, reducer)
            step += 1
            in_epoch += 1
            hit = bool(skipped)
            # the points at which this rank may leave the loop or write a checkpoint: the agreement below runs there too (a run that
            # ended by max_steps / its last epoch between two log lines finished without the NotImplementedError the policy promises)
            leaving = (max_steps is not None and step >= max_steps) or step % cfg.val_check_interval == 0 or in_epoch == n_batches
            if policy == "raise" and world > 1:
                # every rank must leave the loop in the same step: a rank that raised alone would leave its peers in the next
                # step's bucket all-reduce until the collective times out - agree on the flag first.  The agreement is a blocking
                # collective + a host read (it drains the queue: no prefetch / second-queue overlap across it), so it runs when the log line is due,
                # not every step: under DDP the raise policy stops within log_every steps of the first unsupported draw, on every rank
                # at the same step
                if step % log_every == 0 or leaving:
                    flag = torch.tensor([1.0 if hit else 0.0], device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                    hit = bool(flag.item() > 0)
                else:
                    hit = False
            if hit and not warned and policy != "count":
                warned = True
                msg = ("dynamic mixing drew an augmentation the device simulator does not apply (%s): it is skipped for that "
                       "utterance (wind noise is mixed additively); counters follow in every log line "
                       "(cfg.unsupported_augmentation = warn | raise | count)" % (sorted(skipped) or "on another rank"))
                if policy == "raise":
                    raise NotImplementedError(msg)
                print("WARNING: " + msg, flush=True)
            if rank == 0 and step % log_every == 0:
                lg = {k: float(v) for k, v in model.logged.items()}
                print("epoch %d step %d %s  (%.2f s/step)%s" % (epoch, step, lg, (time.time() - t0) / log_every,
                                                               "  not applied: %s" % dict(skipped) if skipped else ""), flush=True)
                t0 = time.time()
            if step % cfg.val_check_interval == 0:
                vl = validate(model, val_loader, dev)
                raise_kernel_errors_on_all_ranks(dev, world)      # every rank fails together, none is left in an all-reduce
                if rank == 0:
                    path = "%s/best_epoch=%02d-step=%06d-val_loss=%.3f.ckpt" % (ckpt_dir(cfg), epoch, step, vl)
                    save_checkpoint(path, model, opt, sched, epoch, step, vl)
                    best = sorted(best + [(vl, path)])
                    for _, p in best[cfg.save_top_k:]:
                        os.remove(p)
                    best = best[:cfg.save_top_k]
                    print("val_loss %.3f -> %s" % (vl, path), flush=True)
            if max_steps is not None and step >= max_steps:
                return model, step
        sched.step()
    if rank == 0 and skipped:
        print("augmentations drawn but not applied on the device path: %s" % skipped, flush=True)
    return model, step


def main(argv=None):
    args = config_parser(argv)
    cfg = Config(**vars(args))
    cfg.read_yaml()
    if int(os.environ.get("RANK", "0")) == 0:
        print(vars(cfg))
    fit(cfg)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
