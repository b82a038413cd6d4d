"""N>1 path on CPU: two gloo ranks drive the bucketed, backward-overlapped gradient all-reduce of
sim2real_lane_segment_amd.trainer with a stand-in backward (the HIP kernels need a GPU; the collective
logic -- bucket planning, slice ranges, averaging -- does not)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fcdensenet_oracle as O
from sim2real_lane_segment_amd.engine import Engine, NetSpec
from sim2real_lane_segment_amd import _lib
from sim2real_lane_segment_amd.trainer import BucketedGradReducer, TrainStepper, plan_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, seg_ranges, n_param, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.zeros(n_param)
    order = []

    def run_segments(sb, se):  # stand-in backward: each rank fills the slices the segments complete
        for s in range(sb, se):
            b, e = seg_ranges[s]
            flat[b:e] = torch.arange(b, e, dtype=torch.float32) * 1e-3 + (rank + 1)
            order.append(s)

    red = BucketedGradReducer(flat, seg_ranges, n_buckets=4)
    red.backward_and_reduce(run_segments)
    mean = flat / world
    expect = torch.arange(n_param, dtype=torch.float32) * 1e-3 + (1 + world) / 2.0
    ok = torch.allclose(mean, expect, rtol=1e-6, atol=1e-6) and order == list(range(len(seg_ranges)))
    torch.save({"ok": bool(ok), "buckets": red.buckets}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_plan_covers_arena():
    cfg = O.fcdensenet67_config(4)
    eng = Engine(NetSpec(n_classes=4), device="cpu")
    buckets = plan_buckets(eng.seg_ranges, 4)
    assert buckets[0][0] == 0 and buckets[-1][1] == eng.n_seg
    assert buckets[0][3] == eng.n_param and buckets[-1][2] == 0
    for (s0, e0, gb0, ge0), (s1, e1, gb1, ge1) in zip(buckets[:-1], buckets[1:]):
        assert e0 == s1 and ge1 == gb0
    assert 2 <= len(buckets) <= 5
    assert len(O.state_spec(cfg)) == len(eng.metas)


def test_two_rank_bucketed_allreduce(tmp_path):
    eng = Engine(NetSpec(n_classes=4), device="cpu")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, eng.seg_ranges, eng.n_param, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False)
        assert res["ok"], f"rank {r} saw wrong reduced gradients"


# ---- TrainStepper itself on two gloo ranks -----------------------------------------------------------------
# The HIP kernels need a GPU; everything else of the DDP step does not.  A recorded-gradient engine replays, segment
# by segment and in the engine's REAL segment plan, the gradient arena the CPU oracle computed for this rank's batch
# (different batches per rank), and applies the oracle's AdamW to the flat arena.  Checked: (i) the reduced arena is
# the mean of the two single-rank arenas, (ii) parameters stay bit-identical across ranks, (iii) they equal AdamW
# applied to the mean gradient.
class _RecordedEngine:
    def __init__(self, real, recorded):
        self.seg_ranges, self.n_seg, self.n_param = real.seg_ranges, real.n_seg, real.n_param
        self.params = real.params.clone()
        self.bnrun = real.bnrun.clone()
        self.grads = torch.zeros(real.n_param)
        self.recorded = recorded
        self.order = []

    def forward(self, x, training, with_backward, drop_scales=None, seed=None):
        return torch.zeros(1), None

    def loss(self, probs, y, weighted):
        return torch.zeros(7), None, None

    def backward(self, loss_scale, sb, se):
        for s in range(sb, se):
            b, e = self.seg_ranges[s]
            self.grads[b:e] = self.recorded[b:e] * loss_scale
            self.order.append(s)

    def adamw_step(self, m, v, step, lr, betas, eps, wd, grad_scale=1.0):
        O.adamw_step(self.params, self.grads * grad_scale, m, v, step, lr, betas[0], betas[1], eps, wd)


def _oracle_grad_arena(eng, seed):
    cfg = O.fcdensenet67_config(4)
    st = O.init_state(cfg, 0)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 3, 32, 32, generator=g)
    y = torch.randint(0, 4, (2, 32, 32), generator=g)
    _, _, gd, _ = O.train_step(O.TrainState(st), x, y, cfg, O.make_drop_scales(cfg, 2, seed), apply_update=False)
    flat = torch.zeros(eng.n_param)
    for m in eng.metas:
        if m.kind == _lib.T_PARAM:
            flat[m.offset:m.offset + m.numel] = gd[m.name].reshape(-1)
    return flat


def _stepper_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real = Engine(NetSpec(n_classes=4), device="cpu")
    real.load_state(O.init_state(O.fcdensenet67_config(4), 0))
    rec = _oracle_grad_arena(real, 100 + rank)
    eng = _RecordedEngine(real, rec)
    if rank == 1:
        eng.params.add_(1.0)  # broadcast_parameters must overwrite this with rank 0's values
    stepper = TrainStepper(eng, lr=1e-3, weight_decay=1e-4, n_buckets=4)
    stepper.broadcast_parameters()
    assert stepper.world == world and len(stepper.reducer.buckets) >= 2
    stepper.step(None, None)
    torch.save({"params": eng.params, "reduced": eng.grads.clone(), "own": rec, "order": eng.order},
               os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_stepper_real_segments(tmp_path):
    port = _free_port()
    mp.spawn(_stepper_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"s{k}.pt"), weights_only=True) for k in range(2)]
    mean = (r[0]["own"] + r[1]["own"]) / 2
    assert r[0]["own"].abs().max() > 0 and not torch.equal(r[0]["own"], r[1]["own"])
    for k in range(2):
        assert torch.allclose(r[k]["reduced"] / 2, mean, rtol=1e-6, atol=1e-12)   # (i) sum over ranks, mean by 1/world
        assert r[k]["order"] == list(range(len(r[k]["order"])))
    assert torch.equal(r[0]["params"], r[1]["params"])                                # (ii)
    eng = Engine(NetSpec(n_classes=4), device="cpu")
    eng.load_state(O.init_state(O.fcdensenet67_config(4), 0))
    p = eng.params.clone()
    O.adamw_step(p, (r[0]["own"] + r[1]["own"]) * 0.5, torch.zeros_like(p), torch.zeros_like(p), 1, 1e-3,
                 weight_decay=1e-4)
    assert torch.allclose(r[0]["params"], p, rtol=0, atol=2e-7)                       # (iii)


# ---- the module path (what Lightning drives): loss.backward() -> EngineOwner._rln_backward_into_fresh_arena ----------
# Same stand-in idea: the owner method is run as it is; the engine underneath replays recorded per-rank gradient arenas
# segment by segment into whatever buffer is bound.  Checked: autograd receives the MEAN over ranks (torch-DDP
# semantics: d(loss) is scaled by 1/world before the summing all-reduce), as views of one fresh flat buffer, the engine's
# own arena is bound again afterwards, and the device-side loss scale reaches every segment.
class _BindableEngine:
    def __init__(self, real, recorded):
        self.seg_ranges, self.n_seg, self.n_param, self.metas = real.seg_ranges, real.n_seg, real.n_param, real.metas
        self.params = real.params.clone()
        self.grads = torch.zeros(real.n_param)
        self.bound = self.grads
        self.recorded = recorded
        self.calls = []

    def bind_grads(self, flat):
        self.bound = self.grads if flat is None else flat

    def backward(self, loss_scale, sb=0, se=None, loss_scale_dev=None):
        se = self.n_seg if se is None else se
        scale = loss_scale * (float(loss_scale_dev) if loss_scale_dev is not None else 1.0)
        for s in range(sb, se):
            b, e = self.seg_ranges[s]
            self.bound[b:e] = self.recorded[b:e] * scale
        self.calls.append((sb, se))


def _module_path_worker(rank, world, port, out_dir):
    from sim2real_lane_segment_amd.owner import EngineOwner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real = Engine(NetSpec(n_classes=4), device="cpu")
    rec = torch.arange(real.n_param, dtype=torch.float32) * 1e-4 + (rank + 1)
    eng = _BindableEngine(real, rec)

    class Owner(EngineOwner):
        pass

    own = Owner()
    own._rln_param_set = {m.name for m in real.metas if m.kind == _lib.T_PARAM}
    own._rln_reducer = BucketedGradReducer(eng.grads, eng.seg_ranges, 4)
    g_loss = torch.tensor(0.5)
    grads = own._rln_backward_into_fresh_arena(eng, g_loss)
    base = grads[0]._base if grads[0]._base is not None else grads[0]
    flat_ok = all(g._base is base or g is base for g in grads) and base.numel() == real.n_param
    torch.save({"flat": base.clone(), "flat_ok": bool(flat_ok), "rebound": eng.bound is eng.grads,
                "own_untouched": bool((eng.grads == 0).all()), "calls": eng.calls, "rec": rec},
               os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_module_path_overlapped_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_module_path_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"m{k}.pt"), weights_only=True) for k in range(2)]
    mean = 0.5 * (r[0]["rec"] + r[1]["rec"]) / 2          # d(loss) = 0.5 on both ranks, mean over the 2 ranks
    for k in range(2):
        assert r[k]["flat_ok"] and r[k]["rebound"] and r[k]["own_untouched"]
        assert torch.allclose(r[k]["flat"], mean, rtol=1e-6, atol=1e-9)
        assert len(r[k]["calls"]) >= 2 and r[k]["calls"][0][0] == 0          # bucket by bucket, not one backward
        assert all(a[1] == b[0] for a, b in zip(r[k]["calls"][:-1], r[k]["calls"][1:]))
