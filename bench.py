#!/usr/bin/env python3
"""Headline benchmark: training images/sec of FCDenseNet67 (num_cls=4) at per-GPU batch 64, 3x120x160
synthetic Duckietown-like frames (BASELINE.json metric / configs[1]).

One "step" = one full SimpleTrainModule training step on one batch already resident in HBM:
forward (train-mode BatchNorm + Dropout2d) -> class-weighted CE on softmax probabilities -> backward ->
AdamW, all in hand-written HIP (librln.so).  With N>1 (launched by torch.distributed.run, one process per
GPU) gradients are averaged with bucketed RCCL all-reduce overlapped with backward; scaling is weak.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     achieved/peak of the dominant kernel class, timed live with HIP events on the launch stream,
  "cpu_baseline": the CPU oracle's training step on the host cores (a bounded batch-8 sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (same guide; not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
TRAIN_FLOPS_PER_IMAGE = 47_718_689_280  # 3 x 15,906,229,760
PMC_SUMMARY = "r02_pmc_traffic.json"     # profiles/: FETCH_SIZE / WRITE_SIZE passes of this same command

# kernel-name prefixes of each profiled class in the rocprofv3 --pmc summary
CLASS_KERNELS = {
    "dense3_fwd": ("void rln::d3_fwd_k<",),
    "dense3_wgrad": ("void rln::d3_wgrad_k<",),
    "dense3_dgrad_pull": ("void rln::d3_pull_k<",),
    "dense_conv3x3_fwd": ("void rln::igemm_k<3, 1, 1, 0,",),
    "dense_conv3x3_dgrad": ("void rln::dgrad_loop_k<",),
    "dense_conv3x3_wgrad": ("void rln::wgrad_dense_q_k<", "void rln::wgrad_k<3, 1, 1,"),
}
# classes that run on the 16-bit MFMA pipe with split operands: products issued per algorithmic multiply-add
SPLIT_CLASSES = ("dense3_fwd", "dense3_wgrad", "dense3_dgrad_pull")


def pmc_traffic(class_name):
    """HBM bytes per launch of a kernel class from the committed PMC pass (FETCH_SIZE/WRITE_SIZE collected in their
    own rocprofv3 --pmc runs of this same command and corrected as tools/pmc_traffic.py documents); None if the
    summary or the class mapping is absent.  Not collected live: counters need the profiler around the process."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_SUMMARY)
    pref = CLASS_KERNELS.get(class_name)
    if pref is None or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    tot, n = 0.0, 0
    for k, v in d.items():
        if k.startswith(pref):
            tot += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"]
    return round(tot / n) if n else None


def read_profile(eng):
    from sim2real_lane_segment_amd import _lib
    L = _lib.lib()
    k = L.rln_profile_num_classes()
    ms = (ctypes.c_double * k)()
    fl = (ctypes.c_double * k)()
    by = (ctypes.c_double * k)()
    ln = (ctypes.c_int64 * k)()
    _lib.check(L.rln_profile_read(eng.ctx, ms, fl, by, ln), "rln_profile_read")
    return [dict(name=L.rln_profile_class_name(i).decode(), ms=ms[i], flops=fl[i], bytes=by[i], launches=int(ln[i]))
            for i in range(k)]


def host_cores():
    """CPU threads this job can really use: affinity mask, capped by the cgroup CPU quota when one is set
    (a GPU box shows all host CPUs but grants a share of them), RLN_CPU_THREADS overrides."""
    if os.environ.get("RLN_CPU_THREADS"):
        return max(1, int(os.environ["RLN_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()
            if q != "max":
                quota = int(q) / int(p)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    elif n > 32:
        n = 16  # no quota visible on a many-core host: stay within the documented one-GPU CPU share
    return max(1, n)


def cpu_baseline(batch=8, timed=5, budget_s=60.0):
    """The CPU oracle (a port of the reference step onto stock PyTorch CPU operators) on the host cores.
    Bounded sample: one warm-up step, then `timed` training steps at batch 8 -> median; one more step
    at batch 64 (the metric's batch) when what is left of the budget allows it."""
    from oracle import fcdensenet_oracle as O
    from sim2real_lane_segment_amd.synthetic import make_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} threads ...", file=sys.stderr, flush=True)
    cfg = O.fcdensenet67_config(4)
    ts = O.TrainState(O.init_state(cfg, 0))
    x, y = make_batch(batch, seed=42)
    t_start = time.perf_counter()
    O.train_step(ts, x, y, cfg, O.make_drop_scales(cfg, batch, 99))  # warm-up (allocator, thread pool)
    times = []
    while len(times) < timed and (len(times) < 1 or time.perf_counter() - t_start + times[-1] < budget_s):
        scales = O.make_drop_scales(cfg, batch, 100 + len(times))
        t0 = time.perf_counter()
        O.train_step(ts, x, y, cfg, scales)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    out = {"value": round(batch / med, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"median of {len(times)} timed training steps (fwd+weighted CE+bwd+AdamW) at batch {batch} after "
                     f"one warm-up step, 120x160, fp32, oracle/fcdensenet_oracle.py on stock PyTorch CPU operators"}
    if time.perf_counter() - t_start + 8.5 * med < budget_s:  # a batch-64 step costs ~8x a batch-8 step
        x64, y64 = make_batch(64, seed=43)
        scales = O.make_drop_scales(cfg, 64, 7)
        t0 = time.perf_counter()
        O.train_step(ts, x64, y64, cfg, scales)
        out["batch64_value"] = round(64 / (time.perf_counter() - t0), 4)
        out["sample"] += "; batch64_value = one timed step at batch 64"
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: starts N fresh worker processes (one per device, RCCL
    rendezvous on 127.0.0.1) BEFORE this process touches the GPU, relays rank 0's JSON line and fails loudly if
    the box has fewer than N devices or any rank fails."""
    import socket
    import subprocess
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if have < n:
        print(f"[bench] --gpus {n} requested but only {have} device(s) are visible", file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    if any(codes):
        print(f"[bench] rank exit codes {codes}", file=sys.stderr)
        return 4
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (the metric is quoted at 64)")
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--buckets", type=int, default=4)
    ap.add_argument("--force-dist", action="store_true",
                    help="single-GPU plumbing check: 1-rank RCCL process group + the bucketed all-reduce path")
    ap.add_argument("--api", choices=("engine", "module"), default="engine",
                    help="engine: TrainStepper (direct C-ABI calls, overlapped all-reduce); module: the reference-shaped "
                         "path training_step -> loss.backward() -> optimizer.step() (what Lightning drives)")
    ap.add_argument("--no-module-api", action="store_true", help="skip the secondary module-API timing")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 (default): fp32-parity arithmetic (split 16-bit MFMA operands, DESIGN.md 4.1); bf16: plain "
                         "one-part bf16 MFMA operands in the dense 3x3 kernels (storage and accumulation stay fp32) -- the "
                         "throughput mode of BASELINE.json configs[1]; its mask agreement is measured in "
                         "tests/test_gpu_dense3.py")
    ap.add_argument("--fwd-arith", default=None, help="dense 3x3 forward arithmetic: fp32 | bf16x1..3 | f16x1..2")
    ap.add_argument("--bwd-arith", default=None, help="dense 3x3 backward arithmetic: fp32 | bf16x1..3 | f16x1..2")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus and not args.force_dist:
        print(f"[bench] --gpus {args.gpus} contradicts WORLD_SIZE={env_world}", file=sys.stderr)
        sys.exit(2)

    # Libraries (RCCL prints a banner at communicator creation) must not pollute stdout: the contract is ONE JSON
    # line there.  Everything until the final print goes to stderr at the file-descriptor level.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from sim2real_lane_segment_amd.synthetic import make_batch
    from sim2real_lane_segment_amd.trainer import TrainStepper
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule

    torch.manual_seed(42)
    model = SimpleTrainModule(lr=1e-3, lrRatio=1e3, decay=1e-4, num_cls=4).to(dev)  # random init, FCDenseNet67
    model.train()
    eng = model._rln_sync()
    if args.dtype == "bf16" and not (args.fwd_arith or args.bwd_arith):
        args.fwd_arith = args.bwd_arith = "bf16x1"
    if args.fwd_arith or args.bwd_arith:
        def parse(a):
            if a in (None, "fp32"):
                return 0, "bf16"
            t, n = a.split("x")
            return int(n), t
        fp, ft = parse(args.fwd_arith)
        bp, bt = parse(args.bwd_arith)
        eng.set_dense_arith(fp, ft, bp, bt)
    stepper = TrainStepper(eng, lr=1e-3, weight_decay=1e-4, n_buckets=args.buckets, force_collectives=args.force_dist)
    stepper.broadcast_parameters()
    if dist.is_initialized() and dist.get_world_size() != world:
        print(f"[bench] RCCL world size {dist.get_world_size()} != WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(5)

    class ModuleStepper:
        """The reference-shaped path: what Lightning runs per batch (SimpleTrain.py:11-30)."""

        def __init__(self, module):
            self.module = module
            (self.opt,), _ = module.configure_optimizers()
            if world > 1 or args.force_dist:
                # loss.backward() itself all-reduces finished gradient buckets on a side stream while the remaining
                # backward runs and hands autograd the mean over ranks (owner.EngineOwner.enable_grad_allreduce)
                module.enable_grad_allreduce(args.buckets, force_collectives=args.force_dist)

        def step(self, x, y):
            self.opt.zero_grad(set_to_none=True)
            loss = self.module.training_step((x, y), 0)
            loss.backward()
            self.opt.step()
            return loss.detach().reshape(1)

    engine_stepper = stepper
    if args.api == "module":
        stepper = ModuleStepper(model)

    B = args.batch
    pool = [make_batch(B, args.height, args.width, seed=42, first_index=(rank * 4 + i) * B, device=dev)
            for i in range(2)]
    for i in range(args.warmup):
        x, y = pool[i % len(pool)]
        out = stepper.step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    from sim2real_lane_segment_amd import _lib
    # ---- timed region: EXACTLY K steps, no instrumentation in the stream ----
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x, y = pool[i % len(pool)]
        out = stepper.step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out[0])
    # ---- the same K steps again with HIP events around every kernel class (launch stream) -> roofline.  The
    # ~600 event pairs per step cost a few % of wall time, so they stay out of the region `value` is taken from;
    # the per-class kernel durations they measure are the quantity the roofline needs. ----
    prof = None
    instrumented_ms = None
    module_api = None
    if args.api == "engine" and world == 1 and not args.no_module_api:
        ms = ModuleStepper(model)
        for i in range(2):
            ms.step(*pool[i % len(pool)])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            ms.step(*pool[i % len(pool)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        module_api = {"value": round(B * args.steps / dt, 2), "unit": "images/sec",
                      "ms_per_step": round(1000.0 * dt / args.steps, 3),
                      "path": "SimpleTrainModule.training_step -> loss.backward() -> FusedAdamW.step()"}
    if not args.no_profile:
        _lib.check(_lib.lib().rln_profile_enable(eng.ctx, 1))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            x, y = pool[i % len(pool)]
            stepper.step(x, y)
        torch.cuda.synchronize()
        instrumented_ms = 1000.0 * (time.perf_counter() - t1) / args.steps
        prof = read_profile(eng)
        _lib.check(_lib.lib().rln_profile_enable(eng.ctx, 0))

    if rank == 0:
        images = world * B * args.steps
        result = {
            "metric": "train images/sec @ batch 64, 3x120x160",
            "value": round(images / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # storage and accumulation are fp32 in both modes; "f32" = split 16-bit MFMA operands at fp32-parity error,
            # "bf16" = plain bf16 MFMA operands in the dense 3x3 kernels (config.dense_arith names the exact setting)
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"FCDenseNet67 num_cls=4 SimpleTrainModule training step (fwd+weighted CE+bwd+AdamW), "
                                   f"per-GPU batch {B}, 3x{args.height}x{args.width} synthetic Duckietown frames, "
                                   f"random-init weights, Dropout2d+BatchNorm in train mode",
                       "global_batch": world * B, "parallelism": f"dp{world}", "final_loss": round(loss, 5),
                       "api": args.api,
                       "dense_arith": "fwd %sx%d / bwd %sx%d split-operand MFMA (0 parts = exact fp32 MFMA), fp32 storage "
                                      "and accumulation" % (eng.dense_arith[1], eng.dense_arith[0], eng.dense_arith[3],
                                                            eng.dense_arith[2])},
        }
        if module_api is not None:
            result["module_api"] = module_api
        if prof is not None:
            timed = [p for p in prof if p["launches"] > 0]
            total_ms = sum(p["ms"] for p in timed)
            dom = max((p for p in timed if p["flops"] > 0), key=lambda p: p["ms"])
            fwd_parts, _, bwd_parts, _ = eng.dense_arith
            parts = fwd_parts if dom["name"] == "dense3_fwd" else bwd_parts
            products = {1: 1, 2: 3, 3: 6}.get(parts, 1) if dom["name"] in SPLIT_CLASSES else 1
            if dom["name"] == "dense3_wgrad" and parts == 2:
                products = 1  # one-part operands in the dense weight gradient (rln_set_wgrad_parts)
            mfma_peak = PEAK_16BIT_MFMA_TFLOPS if dom["name"] in SPLIT_CLASSES else PEAK_F32_MFMA_TFLOPS
            tfl = dom["flops"] / (dom["ms"] * 1e-3) / 1e12          # algorithmic flops (2 per multiply-add)
            gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9           # algorithmic bytes
            mfma_frac = tfl * products / mfma_peak                  # matrix-pipe work incl. the split products
            hbm_frac = gbs / PEAK_HBM_GBS
            # the binding roof of a class is the one its work sits closer to
            if mfma_frac >= hbm_frac:
                roof = {"bound": "mfma", "achieved": round(tfl * products, 3), "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": round(mfma_frac, 4)}
            else:
                roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(hbm_frac, 4)}
            roof.update({
                "kernel": dom["name"], "traffic": pmc_traffic(dom["name"]),
                "traffic_unit": f"HBM bytes per launch (rocprofv3 --pmc pass, profiles/{PMC_SUMMARY})",
                "alg_bytes_per_launch": round(dom["bytes"] / dom["launches"]), "launches": dom["launches"],
                "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                "share_of_kernel_time": round(dom["ms"] / total_ms, 4),
                "instrumented_ms_per_step": round(instrumented_ms, 3),
                "alg_tflops": round(tfl, 3), "alg_GBps": round(gbs, 1),
                "mfma_products_per_mac": products, "frac_of_mfma_peak": round(mfma_frac, 4),
                "frac_of_hbm_peak": round(hbm_frac, 4),
                "whole_step_tflops": round(TRAIN_FLOPS_PER_IMAGE * images / elapsed / 1e12, 3)
                if (args.height, args.width) == (120, 160) else None,
                "whole_step_frac_of_f32_mfma_peak": round(TRAIN_FLOPS_PER_IMAGE * images / elapsed / 1e12
                                                          / PEAK_F32_MFMA_TFLOPS, 4)
                if (args.height, args.width) == (120, 160) else None})
            result["roofline"] = roof
            result["kernel_classes"] = [
                {"name": p["name"], "ms_per_step": round(p["ms"] / args.steps, 4),
                 "launches_per_step": p["launches"] // args.steps,
                 "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2) if p["flops"] and p["ms"] else None,
                 "alg_GBps": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1) if p["bytes"] and p["ms"] else None}
                for p in sorted(timed, key=lambda p: -p["ms"])]
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
