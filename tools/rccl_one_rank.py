"""The data-parallel exchange on REAL RCCL with the one GPU a gpurun box has: a process group of ONE rank on backend
'nccl', and a FlatGradSync told to behave as if it had two (`world = 2`).  Every call the multi-GPU path makes then
really happens -- the layout broadcast, the early bucket's `all_reduce(async_op=True)` from inside the backward pass (the
autograd thread), `wait()` + the suffix all-reduce in step(), the update with 1/world folded in -- only the sum has one
addend.  Checked: (a) exactly -- exact_exchange_check(): which collectives were asked for (early prefix from the backward pass,
suffix from step(), together the whole buffer), the gradients in the flat buffer bit for bit, and the update recomputed with
torch ops from the buffers (1/world applied once); (b) end to end -- parameters after 2 steps agree, to the run-to-run noise
of the kernels' atomics, with a run without any sync whose gradients are halved by the update's scale.
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


def exact_exchange_check(steps=3):
    """The exchange against an EXACT expectation (no kernel noise in the comparison): every collective the optimizer asks for
    is recorded (offset, length, asynchronous?) and, for the last step, parameters and momentum are recomputed with torch ops
    from the flat buffers as they stood before the update -- g' = G / world + wd P, M' = mu M + g', P' = P - lr (g' + mu M').
    With one real rank the SUM has one addend, so G must come out of the two collectives bit for bit as the backward pass
    left it, the early bucket must be [0, early_end) launched asynchronously from the backward pass, the suffix
    [early_end, total) from step(), and the update must have applied 1/world exactly once."""
    import istgcn_amd  # noqa: F401
    from istgcn_amd import harness, dp
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = importlib.import_module('istgcn_amd.net.st_gcn_msgcn').Model(
        3, 60, {'layout': 'ntu-rgb+d', 'strategy': 'spatial_3'}, True, dropout=0.0, compute_dtype=torch.bfloat16)
    model.apply(harness.weights_init)
    model.to(dev).train()
    opt = harness.make_optimizer(model)
    sync = dp.FlatGradSync(model)
    sync.world = 2
    opt.attach_sync(sync)
    calls = []
    inner = sync.all_reduce_flat_

    def recording(flat, async_op=False):
        calls.append(((flat.data_ptr() - opt.G.data_ptr()) // 4, flat.numel(), bool(async_op), torch.is_grad_enabled()))
        return inner(flat, async_op=async_op)
    sync.all_reduce_flat_ = recording
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 64, 25, 2, generator=g).to(dev)
    y = torch.randint(0, 60, (8,), generator=g).to(dev)
    for _ in range(steps - 1):
        harness.train_step(model, opt, x, y, sync)
    del calls[:]
    out = model(x.float())
    loss = torch.nn.functional.cross_entropy(out, y)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    n_bwd = len(calls)                                        # collectives launched from inside the backward pass
    P0, M0 = opt.P.clone(), opt.M.clone()
    grads = {id(p): p.grad.detach().clone() for p in opt._live}          # what autograd delivered, per parameter
    opt.step()
    torch.cuda.synchronize()
    grp = opt.param_groups[0]
    G = opt.G
    # (1) the flat gradient buffer holds every parameter's gradient bit for bit (gather + two one-addend all-reduces)
    same = all(torch.equal(gv.reshape(-1), grads[id(p)].float().reshape(-1)) if gv.is_contiguous() else
               torch.equal(gv, grads[id(p)].float()) for p, gv in zip(opt._live, opt._gviews))
    # (2) the update, recomputed
    gp = G * (1.0 / 2.0) + grp['weight_decay'] * P0
    M1 = grp['momentum'] * M0 + gp
    P1 = P0 - grp['lr'] * (gp + grp['momentum'] * M1)
    dP = float((opt.P - P1).abs().max() / max(1e-30, float(P1.abs().max())))
    dM = float((opt.M - M1).abs().max() / max(1e-30, float(M1.abs().max())))
    total, early_end = opt.G.numel(), opt._early_end
    want = [(0, early_end, True), (early_end, total - early_end, False)]
    got = [c[:3] for c in calls]
    ok = same and dP < 1e-6 and dM < 1e-6 and got == want and n_bwd == 1 and early_end > 0
    return ok, {'collectives (offset, elements, async)': got, 'expected': want, 'launched_from_backward': n_bwd,
                'gradients_bit_exact_in_flat_buffer': bool(same), 'update_rel_err_params': dP, 'update_rel_err_momentum': dM}


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
    ok_exact, exact = exact_exchange_check()
    ok = ok and ok_exact
    print(json.dumps({'backend': dist.get_backend(), 'rccl_one_rank_exchange': 'ok' if ok else 'MISMATCH', 'exact_check': exact,
                      'loss_sync': l1, 'loss_plain': l0, 'loss_plain_again': l2, 'max_abs_param_diff': worst, 'run_to_run_noise': noise,
                      'early_all_reduces_launched_from_backward': o1.early_launches,
                      'bucket_bytes': o1.bucket_bytes, 'early_bucket_bytes': o1._early_end * 4,
                      'seconds_sync': round(t1, 3), 'seconds_plain': round(t0, 3)}))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
