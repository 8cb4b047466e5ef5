"""CPU tests of the step driver (uwudiff_amd/engine.py): warm-up scheduler resume, the ModelCheckpoint callback, the
launcher's resume pass-through, and the data-parallel path with a NON-fused optimizer (world_size 2, gloo)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sched(opt, warm=5):
    from uwudiff_amd.engine import GradualWarmupScheduler

    after = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=1e-7)
    return GradualWarmupScheduler(opt, 1, warm, after)


def test_warmup_scheduler_state_dict_resumes_the_lr_curve():
    """`lr_schedulers` was [] with warm-up on, so a resume restarted the ramp.  The LR at every step after a
    resume (inside the ramp, and after it inside the cosine) must equal the uninterrupted run's."""
    def run(n, resume_at=None):
        p = nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1e-3)
        s = _sched(opt)
        lrs = []
        for i in range(n):
            if resume_at is not None and i == resume_at:
                sd = s.state_dict()
                sd = torch.load(_roundtrip(sd), weights_only=True)  # plain data: survives the safe loader
                p2 = nn.Parameter(torch.zeros(1))
                opt = torch.optim.SGD([p2], lr=1e-3)
                s = _sched(opt)
                s.load_state_dict(sd)
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            s.step()
        return lrs

    full = run(20)
    for k in (3, 5, 6, 12):
        assert run(20, resume_at=k) == full, k
    assert full[0] == 0.0 and abs(full[5] - 1e-3) < 1e-12 and full[19] < full[6]


def _roundtrip(obj):
    import io

    b = io.BytesIO()
    torch.save(obj, b)
    b.seek(0)
    return b


class _FakeUnet(nn.Module):
    """One flat parameter + the DiT's `set_grad_ready_hook` contract (slices reported from inside backward)."""

    def __init__(self, n=1000):
        super().__init__()
        self.flat = nn.Parameter(torch.arange(n, dtype=torch.float32) / n)
        self._hook = None
        self.flat.register_post_accumulate_grad_hook(self._fire)

    def _fire(self, p):
        if self._hook is not None:
            n = p.numel()
            self._hook(p.grad, [(n // 2, n - n // 2, None)])

    def set_grad_ready_hook(self, hook, group_layers=4):
        self._hook = hook


class _FakeTrainer(nn.Module):
    def __init__(self, opt_cls):
        super().__init__()
        self.unet = _FakeUnet()
        self.register_buffer("ema_loss", torch.tensor(0.0))
        self.opt_cls = opt_cls

    def configure_optimizers(self):
        return self.opt_cls(self.unet.parameters(), lr=1.0)

    def training_step(self, batch, idx):
        x = batch[0]
        return {"loss": (self.unet.flat * x.mean()).sum()}


class _DM:
    def __init__(self, scale):
        self.scale = scale

    def setup(self, stage):
        pass

    def train_dataloader(self):
        return [(torch.full((4,), self.scale), [""], [], {}, {})]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uwudiff_amd.engine import Fitter

    tr = _FakeTrainer(torch.optim.SGD)
    if rank == 1:  # replicas that start different must be made identical (rank 0's parameters win)
        with torch.no_grad():
            tr.unet.flat.add_(5.0)
    p0 = torch.arange(1000, dtype=torch.float32) / 1000
    fit = Fitter(max_steps=2, accelerator="cpu", log_every_n_steps=1)
    fit.fit(tr, _DM(scale=float(rank + 1)))
    # grad per rank = (rank+1) everywhere; mean over 2 ranks = 1.5; two SGD steps at lr 1
    want = p0 - 2 * 1.5
    ok = torch.allclose(tr.unet.flat.detach(), want, atol=1e-6) and tr.unet._hook is None
    q.put((rank, bool(ok), float((tr.unet.flat.detach() - want).abs().max())))
    dist.destroy_process_group()


def test_foreign_optimizer_data_parallel_world2_gloo():
    """with world > 1 and a non-FusedAdamW optimizer the block gradients were all-reduced twice (early hook
    + whole-buffer exchange) -> world x too large.  Also covers the rank-0 parameter broadcast."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_model_checkpoint_callback_writes_and_prunes(tmp_path):
    from uwudiff_amd.engine import Fitter, ModelCheckpoint

    cb = ModelCheckpoint(dirpath=str(tmp_path), filename="step={step}", every_n_train_steps=2, save_top_k=2, save_last=True)
    tr = _FakeTrainer(torch.optim.SGD)
    fit = Fitter(max_steps=7, accelerator="cpu", callbacks=[cb], max_epochs=100)
    fit.fit(tr, _DM(scale=1.0))
    files = sorted(os.listdir(tmp_path))
    assert files == ["last.ckpt", "step=4.ckpt", "step=6.ckpt"], files
    ck = torch.load(os.path.join(tmp_path, "step=6.ckpt"), weights_only=True)
    assert ck["global_step"] == 6 and "unet.flat" in ck["state_dict"]
    # fast_dev_run disables checkpoint callbacks (as Lightning does)
    cb2 = ModelCheckpoint(dirpath=str(tmp_path / "dev"), every_n_train_steps=1)
    Fitter(fast_dev_run=True, accelerator="cpu", callbacks=[cb2]).fit(_FakeTrainer(torch.optim.SGD), _DM(1.0))
    assert not os.path.exists(tmp_path / "dev")


def test_checkpoint_resume_continues_epoch_count_and_topk(tmp_path):
    """the checkpoint stores the epoch it was written in; a resumed run continues from it (max_epochs bounds the
    TOTAL, {epoch} in file names does not restart), top-k pruning remembers the files of the interrupted run, the default
    file name is Lightning's, and save_top_k = 0 / -1 mean none / all."""
    from uwudiff_amd.engine import Fitter, ModelCheckpoint

    d = tmp_path / "a"
    cb = ModelCheckpoint(dirpath=str(d), every_n_train_steps=1, save_top_k=2)
    fit = Fitter(accelerator="cpu", callbacks=[cb], max_epochs=3)  # the fake loader has one batch per epoch
    fit.fit(_FakeTrainer(torch.optim.SGD), _DM(1.0))
    assert sorted(os.listdir(d)) == ["epoch=1-step=2.ckpt", "epoch=2-step=3.ckpt"]
    ck = torch.load(d / "epoch=2-step=3.ckpt", weights_only=True)
    assert ck["epoch"] == 2 and ck["global_step"] == 3
    # resume from step 3 / epoch 2 with max_epochs = 5: exactly ... more epochs, numbered on
    cb2 = ModelCheckpoint(dirpath=str(d), every_n_train_steps=1, save_top_k=2)
    fit2 = Fitter(accelerator="cpu", callbacks=[cb2], max_epochs=5)
    fit2.fit(_FakeTrainer(torch.optim.SGD), _DM(1.0), ckpt_path=str(d / "epoch=2-step=3.ckpt"))
    assert fit2.global_step == 5 and fit2.current_epoch == 5  # epoch 2 was complete: two more epochs, numbered 3 and 4
    assert sorted(os.listdir(d)) == ["epoch=3-step=4.ckpt", "epoch=4-step=5.ckpt"]  # the two files of run 1 were pruned
    for k, want in ((0, []), (-1, ["epoch=0-step=1.ckpt", "epoch=1-step=2.ckpt", "epoch=2-step=3.ckpt"])):
        dk = tmp_path / f"k{k}"
        Fitter(accelerator="cpu", callbacks=[ModelCheckpoint(dirpath=str(dk), every_n_train_steps=1, save_top_k=k)],
               max_epochs=3).fit(_FakeTrainer(torch.optim.SGD), _DM(1.0))
        assert (sorted(os.listdir(dk)) if os.path.exists(dk) else []) == want


class _DM3(_DM):
    """three batches per epoch"""

    def train_dataloader(self):
        return [(torch.full((4,), self.scale), [""], [], {}, {}) for _ in range(3)]


def test_partial_epoch_is_not_counted_and_resume_skips_the_right_batches(tmp_path):
    """max_steps inside an epoch must not fire the epoch-end hooks nor count the epoch; the checkpoint stores how far
    into its epoch it was written; max_steps ON an epoch boundary must not run an empty extra epoch."""
    from uwudiff_amd.engine import Fitter, ModelCheckpoint

    # (1) max_steps = 4 with 3 batches per epoch: epoch 0 complete, epoch 1 cut after one batch
    d = tmp_path / "p"
    cb = ModelCheckpoint(dirpath=str(d), every_n_epochs=1, save_top_k=-1, save_last=True)
    fit = Fitter(accelerator="cpu", callbacks=[cb], max_steps=4, max_epochs=100)
    fit.fit(_FakeTrainer(torch.optim.SGD), _DM3(1.0))
    assert fit.global_step == 4 and fit.current_epoch == 1 and fit.batches_in_epoch == 1
    assert sorted(os.listdir(d)) == ["epoch=0-step=3.ckpt", "last.ckpt"]  # one epoch-end file: the partial epoch wrote none
    last = torch.load(d / "last.ckpt", weights_only=True)
    assert last["epoch"] == 1 and last["global_step"] == 4 and last["batches_in_epoch"] == 1
    # resume to max_steps = 8: skips ONE batch of epoch 1, finishes it at step 6 (epoch-end file named epoch=1), runs two of epoch 2
    cb2 = ModelCheckpoint(dirpath=str(d), every_n_epochs=1, save_top_k=-1)
    fit2 = Fitter(accelerator="cpu", callbacks=[cb2], max_steps=8, max_epochs=100)
    fit2.fit(_FakeTrainer(torch.optim.SGD), _DM3(1.0), ckpt_path=str(d / "last.ckpt"))
    assert fit2.global_step == 8 and fit2.current_epoch == 2 and fit2.batches_in_epoch == 2
    assert "epoch=1-step=6.ckpt" in os.listdir(d) and "epoch=2-step=8.ckpt" not in os.listdir(d)
    # (2) max_steps = 6 = two whole epochs: both epoch-end hooks fire once, no third (empty) epoch, no duplicate file
    d2 = tmp_path / "q"
    cb3 = ModelCheckpoint(dirpath=str(d2), every_n_epochs=1, save_top_k=2)
    fit3 = Fitter(accelerator="cpu", callbacks=[cb3], max_steps=6, max_epochs=100)
    fit3.fit(_FakeTrainer(torch.optim.SGD), _DM3(1.0))
    assert fit3.global_step == 6 and fit3.current_epoch == 2
    assert sorted(os.listdir(d2)) == ["epoch=0-step=3.ckpt", "epoch=1-step=6.ckpt"]
    # (3) a checkpoint written at an epoch's last batch resumes into the NEXT epoch
    fit4 = Fitter(accelerator="cpu", callbacks=[], max_steps=7, max_epochs=100)
    fit4.fit(_FakeTrainer(torch.optim.SGD), _DM3(1.0), ckpt_path=str(d2 / "epoch=1-step=6.ckpt"))
    assert fit4.global_step == 7 and fit4.current_epoch == 2 and fit4.batches_in_epoch == 1


def test_launcher_passes_resume_checkpoint_through():
    src = open(os.path.join(ROOT, "test_scripts", "test_train.py")).read()
    assert "ckpt_path=ckpt_path" in src and "not supported" not in src
