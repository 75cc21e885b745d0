"""The data-parallel exchange on REAL RCCL with the one GPU a gpurun box has: a process group of ONE rank on backend
'nccl', and a FlatGradSync told to behave as if it had two (`world = 2`).  Every call the multi-GPU path makes then
really happens -- the layout broadcast, the early bucket's `all_reduce(async_op=True)` from inside the backward pass (the
autograd thread), `wait()` + the suffix all-reduce in step(), the update with 1/world folded in -- only the sum has one
addend.  Checked: parameters after 2 steps agree (to the run-to-run noise of the atomics) with a run without any sync whose gradients are halved by the update's scale.
usage: python tools/rccl_one_rank.py          (prints one JSON line)"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def run(with_sync, steps=2):
    import istgcn_amd  # noqa: F401
    from istgcn_amd import harness, dp
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(
        3, 60, {'layout': 'ntu-rgb+d', 'strategy': 'spatial_3'}, True, dropout=0.0, compute_dtype=torch.bfloat16)
    model.apply(harness.weights_init)
    model.to(dev).train()
    opt = harness.make_optimizer(model, loss_scale=1.0 if with_sync else 2.0)
    sync = None
    if with_sync:
        sync = dp.FlatGradSync(model)
        sync.world = 2                        # one real rank, the two-rank code path
        opt.attach_sync(sync)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 64, 25, 2, generator=g).to(dev)
    y = torch.randint(0, 60, (8,), generator=g).to(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        if with_sync:
            loss = harness.train_step(model, opt, x, y, sync)
        else:
            # the same arithmetic without any collective: loss_scale 2 makes the update divide by 2, so the backward pass
            # must NOT be scaled up -- run it by hand
            out = model(x.float())
            loss = torch.nn.functional.cross_entropy(out, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
    torch.cuda.synchronize()
    return model, opt, float(loss.detach()), time.perf_counter() - t0


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    m1, o1, l1, t1 = run(True)
    m0, o0, l0, t0 = run(False)
    m2, _, l2, _ = run(False)

    def diff(ma, mb):
        w = 0.0
        for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
            if a.is_floating_point():
                w = max(w, float((a.float() - b.float()).abs().max()))
        return w
    # yardstick: two runs WITHOUT any collective differ by the order of the fp64 / fp32 atomics (BatchNorm sums, weight
    # gradients); the run through RCCL may differ from them by no more than a few times that
    worst, noise = diff(m1, m0), diff(m0, m2)
    ok = worst <= 4 * noise + 1e-6 and abs(l1 - l0) <= 4 * abs(l0 - l2) + 1e-2 and o1.early_launches == 1
    print(json.dumps({'backend': dist.get_backend(), 'rccl_one_rank_exchange': 'ok' if ok else 'MISMATCH',
                      'loss_sync': l1, 'loss_plain': l0, 'loss_plain_again': l2, 'max_abs_param_diff': worst, 'run_to_run_noise': noise,
                      'early_all_reduces_launched_from_backward': o1.early_launches,
                      'bucket_bytes': o1.bucket_bytes, 'early_bucket_bytes': o1._early_end * 4,
                      'seconds_sync': round(t1, 3), 'seconds_plain': round(t0, 3)}))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
