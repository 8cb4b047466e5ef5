"""Step driver that replaces the subset of ``lightning.Trainer`` the reference's launcher uses
(test_scripts/test_train.py:43-77): device placement, the fit loop, backward, data-parallel gradient exchange,
global-norm clipping, optimizer + per-step LR schedule, rank-aware seeding, periodic logging.

Lightning's loop is third-party control plane; the hot path it drives (loss -> denoiser fwd/bwd -> all-reduce ->
clip -> AdamW) is the part this build implements natively.  Checkpoint / resume follow Lightning's file layout
(``Fitter.save_checkpoint`` / ``fit(ckpt_path=...)``; the ``ModelCheckpoint`` callback of the YAMLs writes them).
"""
import json
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from .gradsync import FlatGradSync
from .optim import FusedAdamW


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    return seed


class ModelCheckpoint:
    """``lightning.pytorch.callbacks.ModelCheckpoint`` subset the reference configs use (configs/demo_training.yaml:14-19):
    ``dirpath``, ``filename``, ``every_n_train_steps`` (or ``every_n_epochs``), ``save_last``.  Registered by
    :meth:`Fitter.fit` as a step hook that calls :meth:`Fitter.save_checkpoint` (Lightning checkpoint layout)."""

    def __init__(self, dirpath=None, filename=None, every_n_train_steps=None, every_n_epochs=None, save_last=False,
                 save_top_k=None, **kw):
        self.dirpath = dirpath or "checkpoints"
        self.filename = filename or "epoch={epoch}-step={step}"  # Lightning's default name
        self.every_n_train_steps = int(every_n_train_steps) if every_n_train_steps else None
        self.every_n_epochs = int(every_n_epochs) if every_n_epochs else None
        self.save_last = bool(save_last)
        self.save_top_k = save_top_k  # monitor: step / mode: max (the reference YAML) == keep the newest k files
        self.kw = kw
        self.saved = []

    def _save(self, fitter, path):
        k = self.save_top_k
        if k == 0:  # Lightning: save_top_k = 0 keeps no periodic file (save_last still writes "last"), -1 keeps all
            return
        fitter.save_checkpoint(path, callbacks={"ModelCheckpoint": {"saved": self.saved + [path]}})
        self.saved.append(path)
        if k is not None and k > 0 and fitter.global_rank == 0:
            while len(self.saved) > k:
                old = self.saved.pop(0)
                if os.path.exists(old):
                    os.remove(old)

    def _path(self, fitter, name=None):
        name = name or self.filename.format(step=fitter.global_step, epoch=fitter.current_epoch)
        return os.path.join(self.dirpath, name + ".ckpt")

    def on_step(self, fitter):
        if self.every_n_train_steps and fitter.global_step % self.every_n_train_steps == 0:
            self._save(fitter, self._path(fitter))

    def on_epoch_end(self, fitter):
        if self.every_n_epochs and (fitter.current_epoch + 1) % self.every_n_epochs == 0:
            self._save(fitter, self._path(fitter))

    def on_fit_end(self, fitter):
        if self.save_last:
            fitter.save_checkpoint(self._path(fitter, "last"))


class LearningRateMonitor:
    def __init__(self, **kw):
        self.kw = kw


class GradualWarmupScheduler:
    """``warmup_scheduler.GradualWarmupScheduler(optimizer, 1, warm_up_period, after)`` semantics
    (reference trainer.py:61-64): lr = base_lr * step / period during warm-up, then ``after``."""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        assert multiplier == 1
        self.optimizer, self.total, self.after = optimizer, total_epoch, after_scheduler
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.n = 0
        self._apply()

    def _apply(self):
        if self.n <= self.total:
            for g, b in zip(self.optimizer.param_groups, self.base):
                g["lr"] = b * (float(self.n) / self.total)

    def step(self):
        self.n += 1
        if self.n > self.total and self.after is not None:
            self.after.step()
        else:
            self._apply()

    # resume (reference trainer.py:76-92 re-links the wrapped scheduler's optimizer on load; here both live in one state)
    def state_dict(self):
        after = self.after.state_dict() if self.after is not None and hasattr(self.after, "state_dict") else None
        return {"n": self.n, "base": list(self.base), "total": self.total, "after": after,
                "lr": [g["lr"] for g in self.optimizer.param_groups]}

    def load_state_dict(self, sd):
        self.n, self.base, self.total = int(sd["n"]), list(sd["base"]), sd["total"]
        if sd.get("after") is not None and self.after is not None:
            self.after.load_state_dict(sd["after"])
        for g, lr in zip(self.optimizer.param_groups, sd["lr"]):  # the LR in force when the checkpoint was written
            g["lr"] = lr


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


class Fitter:
    def __init__(self, max_steps=-1, precision="bf16-mixed", gradient_clip_val=None, fast_dev_run=False,
                 log_every_n_steps=10, accelerator="gpu", devices=1, strategy=None, callbacks=(), logger=(),
                 max_epochs=None, **ignored):
        self.max_steps = max_steps
        self.precision = str(precision)
        self.gradient_clip_val = gradient_clip_val
        self.fast_dev_run = fast_dev_run
        self.log_every_n_steps = max(1, int(log_every_n_steps))
        self.max_epochs = max_epochs
        self.global_rank, self.world_size, self.local_rank = _dist_env()
        self.global_step = 0
        self.current_epoch = 0
        self.history = []
        self.step_hooks = []  # callables(fitter) run after every optimizer step (e.g. periodic save_checkpoint)
        self.callbacks = list(callbacks or [])
        use_gpu = torch.cuda.is_available() and str(accelerator) != "cpu"
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl" if use_gpu else "gloo")
        if use_gpu:
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
            self.device = torch.device("cpu")

    @property
    def compute_dtype(self):
        return "fp32" if self.precision.startswith("32") else "bf16"

    # ------------------------------------------------------------------ checkpoint / resume
    # Lightning's checkpoint layout (the keys `Trainer.fit(ckpt_path=...)` and the reference's loader read:
    # loader.py:24-46 extracts `state_dict` entries by prefix, e.g. "unet."): the module's state_dict carries the
    # denoiser as `unet.<diffusers-style names>`, optimizer / scheduler states are the torch ones.
    def save_checkpoint(self, path, callbacks=None):
        """Write the state of the current / last ``fit`` (rank 0 only).  Returns the dict that was saved."""
        module, opt, sched = self._fit_state

        def snap(o):  # detached CPU copy (torch's optimizer.state_dict() aliases the live state)
            if torch.is_tensor(o):
                return o.detach().cpu().clone()
            if isinstance(o, dict):
                return {k: snap(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(snap(v) for v in o)
            return o

        # "batches_in_epoch": how far into epoch `epoch` the run was (not inferred from global_step % len(loader),
        # which is wrong once an epoch was cut short by max_steps or the loader length changes)
        ckpt = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "uwudiff_amd",
                "batches_in_epoch": int(getattr(self, "batches_in_epoch", 0)),
                "callbacks": callbacks or {},
                "state_dict": snap(dict(module.state_dict())),
                "optimizer_states": [snap(opt.state_dict())],
                "lr_schedulers": [sched.state_dict()] if sched is not None and hasattr(sched, "state_dict") else []}
        if self.global_rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(ckpt, path)
        return ckpt

    def load_checkpoint(self, path):
        """Resume: parameters (+ bf16 shadow), optimizer moments / step count, scheduler, global_step."""
        module, opt, sched = self._fit_state
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = ckpt["state_dict"]
        # the denoiser keeps ONE flat parameter and exposes named views: it restores itself from the `unet.*` entries;
        # everything else (buffers such as ema_loss, other sub-modules) goes through the generic loader
        module.unet.load_state_dict({k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")})
        rest = {k: v for k, v in sd.items() if not k.startswith("unet.")}
        unexpected = module.load_state_dict(rest, strict=False).unexpected_keys
        if unexpected:
            raise RuntimeError(f"checkpoint has entries this trainer does not know: {unexpected[:5]}")
        if hasattr(module.unet, "refresh_shadow"):
            module.unet.refresh_shadow()
        opt.load_state_dict(ckpt["optimizer_states"][0])
        if sched is not None and ckpt.get("lr_schedulers") and hasattr(sched, "load_state_dict"):
            sched.load_state_dict(ckpt["lr_schedulers"][0])
        self.global_step = int(ckpt["global_step"])
        self.current_epoch = int(ckpt.get("epoch", 0))
        self._resume_batches = ckpt.get("batches_in_epoch")  # None: a checkpoint of an earlier version
        saved = (ckpt.get("callbacks") or {}).get("ModelCheckpoint", {}).get("saved")
        if saved is not None:  # top-k pruning continues over the files the interrupted run had written
            for cb in self.callbacks:
                if isinstance(cb, ModelCheckpoint):
                    cb.saved = list(saved)
        return ckpt

    def _to_device(self, batch):
        x, captions, tok, added, ca = batch
        x = x.to(self.device, non_blocking=True)
        tok = [{k: v.to(self.device, non_blocking=True) for k, v in t.items()} for t in tok]
        added = {k: v.to(self.device, non_blocking=True) for k, v in added.items()}
        return x, captions, tok, added, ca

    def fit(self, module, datamodule, ckpt_path=None):
        if hasattr(module.unet, "cfg"):
            module.unet.cfg.compute_dtype = self.compute_dtype
            if hasattr(module.unet, "P"):
                module.unet.P.bf16 = self.compute_dtype == "bf16"
        module.to(self.device)
        module._trainer = self
        datamodule.setup("fit")
        loader = datamodule.train_dataloader()
        cfg = module.configure_optimizers()
        opt = cfg["optimizer"] if isinstance(cfg, dict) else cfg
        sched = cfg["lr_scheduler"]["scheduler"] if isinstance(cfg, dict) and "lr_scheduler" in cfg else None
        fused = isinstance(opt, FusedAdamW)
        sync = FlatGradSync(self.world_size)
        if fused:  # the early (inside-backward) reduction is consumed by FusedAdamW's per-chunk waits only; a foreign
            sync.attach(module.unet)  # optimizer takes the plain whole-buffer exchange below
        elif hasattr(module.unet, "set_grad_ready_hook"):
            module.unet.set_grad_ready_hook(None)
        self._fit_state = (module, opt, sched)
        if ckpt_path is not None:
            self.load_checkpoint(ckpt_path)
        if self.world_size > 1 and hasattr(module.unet, "flat"):
            # identical replicas (Lightning DDP broadcasts rank 0's module state): do not rely on per-rank seeding order
            dist.broadcast(module.unet.flat.data, src=0)
            if hasattr(module.unet, "refresh_shadow"):
                module.unet.refresh_shadow()
        # Lightning's fast_dev_run disables checkpoint callbacks and loggers
        ckpt_cbs = [] if self.fast_dev_run else [cb for cb in self.callbacks if isinstance(cb, ModelCheckpoint)]
        params = [p for g in opt.param_groups for p in g["params"]]
        max_steps = 1 if self.fast_dev_run else self.max_steps
        t0 = time.time()
        epoch = self.current_epoch if ckpt_path is not None else 0  # a resumed run continues its epoch count (max_epochs, {epoch})
        done = False
        # resume mid-epoch like Lightning does: skip the batches the interrupted epoch had already consumed
        n_batches = max(len(loader), 1)
        resume_skip = 0
        if ckpt_path is not None:
            rb = getattr(self, "_resume_batches", None)
            resume_skip = int(rb) if rb is not None else (self.global_step % n_batches)
            if rb is None and resume_skip == 0 and self.global_step > 0:
                resume_skip = n_batches  # (old checkpoints: written after the last batch of their epoch)
            if resume_skip >= n_batches:
                epoch += 1  # the checkpoint was written after the last batch of its epoch: that epoch is complete
                resume_skip = 0
        self.current_epoch = epoch
        self.batches_in_epoch = resume_skip
        grads_zeroed = False
        one = None
        while not done:
            finished_epoch, ran = True, 0
            for bi, batch in enumerate(loader):
                if bi < resume_skip:
                    continue
                if 0 <= max_steps <= self.global_step:  # nothing left to do (resumed at max_steps)
                    done, finished_epoch = True, False
                    break
                module.global_step = self.global_step
                if not grads_zeroed:  # (a fused step leaves the gradients zeroed: FusedAdamW.step(zero_grad=True))
                    for p in params:
                        if p.grad is not None:
                            p.grad.zero_()
                out = module.training_step(self._to_device(batch), self.global_step)
                loss = out["loss"]
                if one is None or one.device != loss.device or one.dtype != loss.dtype:
                    one = torch.ones_like(loss)  # (kept: autograd's own ones_like(loss) is a fill launch per step)
                loss.backward(one)
                clip = None
                if isinstance(opt, FusedAdamW):
                    chunks = sync.all_reduce(params[0].grad)
                    if self.gradient_clip_val:
                        sync.wait_all()
                        clip = opt.grad_norm_clip(self.gradient_clip_val, pre_scale=sync.pre_scale)
                        opt.step(clip=clip, pre_scale=sync.pre_scale, zero_grad=True)
                    else:
                        opt.step(pre_scale=sync.pre_scale, chunks=chunks, before_chunk=sync.wait_chunk, zero_grad=True)
                    grads_zeroed = True
                else:  # foreign optimizer (e.g. lion): plain torch path on the flat buffer
                    for p in params:
                        if self.world_size > 1:
                            sync.all_reduce(p.grad)  # same chunked exchange; nothing was reduced early (no hook)
                            sync.wait_all()
                            p.grad.mul_(sync.pre_scale)
                    if self.gradient_clip_val:
                        torch.nn.utils.clip_grad_norm_(params, self.gradient_clip_val)
                    opt.step()
                    if hasattr(module.unet, "refresh_shadow"):
                        module.unet.refresh_shadow()
                if sched is not None:
                    sched.step()
                self.global_step += 1
                ran += 1
                self.batches_in_epoch = bi + 1
                if 0 <= max_steps <= self.global_step:  # checked at the END of the step: no empty extra epoch afterwards
                    done = True
                    finished_epoch = bi + 1 >= n_batches
                for hook in self.step_hooks:
                    hook(self)
                for cb in ckpt_cbs:
                    cb.on_step(self)
                if self.global_step % self.log_every_n_steps == 0 or self.fast_dev_run or done:
                    rec = {"step": self.global_step, "loss": float(loss.detach()), "ema_loss": float(module.ema_loss),
                           "lr": opt.param_groups[0]["lr"], "elapsed_s": round(time.time() - t0, 3)}
                    self.history.append(rec)
                    if self.global_rank == 0:
                        print(json.dumps(rec), flush=True)
                if done:
                    break
            if not finished_epoch or (ran == 0 and resume_skip == 0 and done):
                break  # cut short by max_steps: the epoch is partial -- no epoch-end hooks, no epoch count
            for cb in ckpt_cbs:
                cb.on_epoch_end(self)
            epoch += 1
            self.current_epoch = epoch
            self.batches_in_epoch = 0
            resume_skip = 0
            if self.max_epochs is not None and epoch >= self.max_epochs:
                break
            if max_steps < 0 and self.max_epochs is None:
                break  # one pass when neither bound is given
        for cb in ckpt_cbs:
            cb.on_fit_end(self)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        return self.history
