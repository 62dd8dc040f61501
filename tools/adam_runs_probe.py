"""Probe: torch's fused Adam kernel over the contiguous RUNS of parameters in the flat blob (kernel, BN weight, BN bias are
neighbours; ~34 runs) instead of the 98 parameter tensors -- same kernel, same numbers.  Step time and equality of the
parameters after a few steps.  usage: gpurun -- python tools/adam_runs_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sps_amd import synthetic
from sps_amd.models.models import SPSNet

cfg = dict(bench.CFG)
cfg["TRAIN"] = {"LR": 7e-5, "WEIGHT_DECAY": 1e-4, "LR_EPOCH": 1, "LR_DECAY": 0.99}
batch = torch.from_numpy(synthetic.make_scene(scan_seed=1, n_azimuth=1750)["batch"]).cuda()


class RunAdam(torch.optim.Adam):
    def __init__(self, params, module, **kw):
        super().__init__(params, **kw)
        self._module = module
        self._fast = None

    def _prepare(self):
        plan = self._module.model._train_plan(torch.device("cuda", torch.cuda.current_device()))
        g = self.param_groups[0]
        spans = sorted((plan.span_of[id(p)] + (p,)) for p in g["params"])
        runs, cur = [], None
        for off, num, p in spans:
            if cur is not None and cur[0] + cur[1] == off:
                cur[1] += num
            else:
                cur = [off, num]
                runs.append(cur)
        flat = plan.flat
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        step = torch.zeros((), dtype=torch.float32, device=flat.device)
        for off, num, p in spans:
            self.state[p] = {"step": step, "exp_avg": m[off:off + num].view(p.shape), "exp_avg_sq": v[off:off + num].view(p.shape)}
        self._fast = (flat, m, v, step, runs, spans)

    @torch.no_grad()
    def step(self, closure=None):
        if self._fast is None:
            self._prepare()
        flat, m, v, step, runs, spans = self._fast
        g0 = spans[0][2].grad
        base = g0.data_ptr() - 4 * spans[0][0]
        ok = all(p.grad is not None and p.grad.data_ptr() == base + 4 * off and p.data_ptr() == flat.data_ptr() + 4 * off for off, num, p in spans)
        if not ok:
            raise RuntimeError("layout changed")
        n = flat.numel()
        gflat = torch.as_strided(g0, (n,), (1,), storage_offset=g0.storage_offset() - spans[0][0])
        grp = self.param_groups[0]
        step += 1
        ps = [flat[o:o + k] for o, k in runs]
        gs = [gflat[o:o + k] for o, k in runs]
        ms = [m[o:o + k] for o, k in runs]
        vs = [v[o:o + k] for o, k in runs]
        b1, b2 = grp["betas"]
        torch._fused_adam_(ps, gs, ms, vs, [], [step] * len(runs), lr=grp["lr"], beta1=b1, beta2=b2, weight_decay=grp["weight_decay"],
                           eps=grp["eps"], amsgrad=False, maximize=False, grad_scale=None, found_inf=None)


def run(kind, steps=40):
    torch.manual_seed(0)
    net = bench.synthetic_weights(SPSNet(cfg)).cuda().train()
    if kind == "torch":
        (opt,), _ = net.configure_optimizers()
    else:
        opt = RunAdam(list(net.parameters()), net, lr=7e-5, weight_decay=1e-4, fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        out = net.training_step(batch, 0)
        out["loss"].backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        step()
    host = (time.perf_counter() - t) / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    return dt, host, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()

for kind in ("torch", "runs", "torch", "runs"):
    dt, host, w = run(kind)
    print(f"{kind:6s} {dt * 1e3:.3f} ms per step, host issue {host * 1e3:.3f} ms")
    if kind == "torch":
        ref = w
    else:
        print("   max |parameter difference| after 43 steps:", float((w - ref).abs().max()))
