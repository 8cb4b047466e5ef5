"""hipGraph capture of the forward+backward of a training step (torch.cuda.CUDAGraph around the ctypes launches).
Usage: python tools/probe_graph.py [B] [model]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd.dit import DiT  # noqa: E402
from uwudiff_amd.objective import DiffusionLoss  # noqa: E402
from uwudiff_amd.optim import FusedAdamW  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    name = sys.argv[2] if len(sys.argv) > 2 else "DiT-S/2"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DiT.from_config(name, cond_dim=1280, init="random", compute_dtype="bf16").to(dev)
    loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl", subfolder="scheduler"))
    opt = FusedAdamW(model.parameters(), lr=1e-6, weight_decay=0.01)
    model.flat.grad = torch.zeros_like(model.flat.data)
    xs, cs = torch.randn(B, 4, 32, 32, device=dev), torch.randn(B, 1280, device=dev)

    def fwd_bwd():
        model.flat.grad.zero_()
        loss, _ = loss_fn(xs, model, added_cond_kwargs={"text_embeds": cs})
        loss.backward()
        return loss

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        fwd_bwd()
        opt.step()

    ms_e = timeit(eager)
    print(f"eager : {ms_e:.3f} ms/step  {B / ms_e * 1e3:.0f} img/s")

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()

    def graphed():
        g.replay()
        opt.step()

    ms_g = timeit(graphed)
    print(f"graph : {ms_g:.3f} ms/step  {B / ms_g * 1e3:.0f} img/s   loss {float(static_loss):.5f}")
    # consistency: gradients of a replay == gradients of an eager pass on the same (injected) draws is covered by the
    # parity tests; here only that replays keep producing finite, changing losses
    l1 = float(static_loss)
    graphed()
    l2 = float(static_loss)
    print("losses of two replays (fresh noise each):", l1, l2)


if __name__ == "__main__":
    main()
