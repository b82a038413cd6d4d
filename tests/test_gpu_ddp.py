"""Two GPU ranks over RCCL: TrainStepper (bucketed all-reduce overlapped with the segmented HIP backward).

Skips on a one-GPU box.  Each rank runs in a fresh child process (the parent never initialises the GPU for them).
Checked after 2 steps: (i) the averaged gradients of step 1 equal the mean of the two single-rank gradient arenas
(each rank also runs its own batch through an un-reduced single-rank engine), (ii) parameters stay bit-identical
across ranks."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(r'''
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); out_dir = sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    from sim2real_lane_segment_amd.engine import Engine, NetSpec
    from sim2real_lane_segment_amd.synthetic import make_batch
    from sim2real_lane_segment_amd.trainer import TrainStepper
    from oracle import fcdensenet_oracle as O
    spec = NetSpec(n_classes=4)
    state = O.init_state(O.fcdensenet67_config(4), 0)
    eng = Engine(spec, device=dev); eng.load_state(state)
    solo = Engine(spec, device=dev); solo.load_state(state)
    if rank == 1:
        eng.params.add_(1.0)   # must be overwritten by the broadcast
    st = TrainStepper(eng, n_buckets=4)
    st.broadcast_parameters()
    assert st.world == world
    x, y = make_batch(4, 64, 96, seed=42, first_index=rank * 4, device=dev)
    # single-rank gradients of this rank's batch (same dropout seed as the stepper's first step)
    probs, _ = solo.forward(x, training=True, with_backward=True, seed=1)
    solo.loss(probs, y, weighted=True)
    solo.backward(1.0)
    own = solo.grads.clone()
    eng.step_seed = 0
    st.step(x, y)
    reduced = eng.grads.clone()        # sum over ranks (the mean is taken by grad_scale = 1/world in AdamW)
    st.step(x, y)
    torch.cuda.synchronize()
    torch.save({"own": own.cpu(), "reduced": reduced.cpu(), "params": eng.params.cpu()},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.gpu
def test_two_gpu_ranks_train_stepper(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "ddp_worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(script), REPO, str(tmp_path)], env=env))
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0, 0]
    r = [torch.load(os.path.join(str(tmp_path), f"g{k}.pt"), weights_only=True) for k in range(2)]
    mean = (r[0]["own"] + r[1]["own"]) / 2
    for k in range(2):
        err = (r[k]["reduced"] / 2 - mean).norm() / mean.norm()
        assert err < 1e-6, f"rank {k}: averaged gradients differ from the mean of the single-rank arenas ({err:.2e})"
    assert torch.equal(r[0]["params"], r[1]["params"])


MODULE_WORKER = textwrap.dedent(r'''
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); out_dir = sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    from sim2real_lane_segment_amd.synthetic import make_batch
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule
    torch.manual_seed(0)
    model = SimpleTrainModule(num_cls=4).to(dev).train()
    ref = SimpleTrainModule(num_cls=4).to(dev).train()
    ref.load_state_dict(model.state_dict())
    x, y = make_batch(4, 64, 96, seed=42, first_index=rank * 4, device=dev)
    # un-reduced gradients of this rank's batch
    l0 = ref.training_step((x, y), 0, seed=5)
    l0.backward()
    own = torch.cat([p.grad.reshape(-1) for p in ref._rln_params_in_arena_order()]).clone()
    # the module path with the overlapped all-reduce inside loss.backward()
    red = model.enable_grad_allreduce(n_buckets=4, force_collectives=(world == 1))
    (opt,), _ = model.configure_optimizers()
    l1 = model.training_step((x, y), 0, seed=5)
    (2.0 * l1).backward()                       # a non-unit d(loss) reaches the kernels from the device (2: exact)
    got = torch.cat([p.grad.reshape(-1) for p in model._rln_params_in_arena_order()]).clone()
    opt.step()
    torch.cuda.synchronize()
    torch.save({"own": own.cpu(), "got": got.cpu(), "buckets": len(red.buckets),
                "params": torch.cat([p.detach().reshape(-1) for p in model._rln_params_in_arena_order()]).cpu()},
               os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
''')


def _run_module_workers(tmp_path, world):
    script = tmp_path / "ddp_module_worker.py"
    script.write_text(MODULE_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(script), REPO, str(tmp_path)], env=env))
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world
    return [torch.load(os.path.join(str(tmp_path), f"m{k}.pt"), weights_only=True) for k in range(world)]


@pytest.mark.gpu
def test_module_path_allreduce_one_rank_plumbing(tmp_path):
    """One RCCL rank with the collectives forced on: loss.backward() runs its segments bucket by bucket with an
    all-reduce of every finished slice on the side stream; with one rank the result must equal the plain backward
    (times the d(loss) = 2 handed in from the device: a power of two commutes with every rounding), bit for bit."""
    r = _run_module_workers(tmp_path, 1)[0]
    assert r["buckets"] >= 2
    assert torch.equal(r["got"], 2.0 * r["own"])


@pytest.mark.gpu
def test_module_path_allreduce_two_ranks(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _run_module_workers(tmp_path, 2)
    mean = 2.0 * (r[0]["own"] + r[1]["own"]) / 2
    for k in range(2):
        err = (r[k]["got"] - mean).norm() / mean.norm()
        assert err < 1e-6, f"rank {k}: module-path gradients differ from the mean over ranks ({err:.2e})"
    assert torch.equal(r[0]["params"], r[1]["params"])
