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
PMC_SUMMARY = "r03_pmc_traffic_f32.json"    # profiles/: FETCH_SIZE / WRITE_SIZE passes of this same command
PMC_SUMMARY_BF16 = "r03_pmc_traffic_bf16.json"

# kernel-name prefixes of each profiled class in the rocprofv3 --pmc summary
CLASS_KERNELS = {
    "dense3_fwd": ("void rln::d3_fwd_k<10,",),
    "dense3_fwd_small": ("void rln::d3_fwd_k<5,",),
    "dense3_fwd_pair": ("void rln::d3_fwd2_k<",),
    "dense3_fwd_finish": ("void rln::d3_fin_k<",),
    "dense3_wgrad": ("void rln::d3_wgrad_k<",),
    "dense3_dgrad_pull": ("void rln::d3_pull_k<",),
    "dense_conv3x3_fwd": ("void rln::igemm_k<3, 1, 1, 0,",),
    "dense_conv3x3_dgrad": ("void rln::d3_dgl_k<", "void rln::dgrad_loop_k<"),
    "dense_conv3x3_wgrad": ("void rln::wgrad_dense_q_k<", "void rln::wgrad_k<3, 1, 1,"),
}
# classes that run on the 16-bit MFMA pipe with split operands: products issued per algorithmic multiply-add
SPLIT_CLASSES = ("dense3_fwd", "dense3_fwd_small", "dense3_fwd_pair", "dense3_fwd_finish", "dense3_wgrad", "dense3_dgrad_pull")  # (dense_conv3x3_dgrad mixes the 16-bit d3_dgl_k and the exact-fp32 small-level launches)


def pmc_traffic(class_name, summary=None):
    """HBM bytes per launch of a kernel class from the committed PMC pass (FETCH_SIZE/WRITE_SIZE collected in their
    own rocprofv3 --pmc runs of this same command and corrected as tools/pmc_traffic.py documents); None if the
    summary or the class mapping is absent.  Not collected live: counters need the profiler around the process."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", summary or PMC_SUMMARY)
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


def roofline_of(prof, eng, instrumented_ms, images, elapsed, args):
    """Roofline object of the dominant kernel class (largest summed event time among the classes with flops)."""
    timed = [p for p in prof if p["launches"] > 0]
    total_ms = sum(p["ms"] for p in timed)
    dom = max((p for p in timed if p["flops"] > 0), key=lambda p: p["ms"])
    fwd_parts, _, bwd_parts, _ = eng.dense_arith
    parts = fwd_parts if dom["name"] in ("dense3_fwd", "dense3_fwd_small", "dense3_fwd_pair", "dense3_fwd_finish") else bwd_parts
    products = {1: 1, 2: 3, 3: 6}.get(parts, 1) if dom["name"] in SPLIT_CLASSES else 1
    if dom["name"] == "dense3_wgrad":
        products = {1: 1, 2: 3, 3: 6}.get(eng.wgrad_parts, 1)  # rln_set_wgrad_parts
    mfma_peak = PEAK_16BIT_MFMA_TFLOPS if dom["name"] in SPLIT_CLASSES else PEAK_F32_MFMA_TFLOPS
    tfl = dom["flops"] / (dom["ms"] * 1e-3) / 1e12          # algorithmic flops (2 per multiply-add)
    gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9           # algorithmic bytes (2 B per stack element with bf16 storage)
    mfma_frac = tfl * products / mfma_peak                  # matrix-pipe work incl. the split products
    hbm_frac = gbs / PEAK_HBM_GBS
    # the binding roof of a class is the one its work sits closer to
    if mfma_frac >= hbm_frac:
        roof = {"bound": "mfma", "achieved": round(tfl * products, 3), "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": round(mfma_frac, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4)}
    summary = PMC_SUMMARY_BF16 if eng.storage == "bf16" else PMC_SUMMARY
    roof.update({
        "kernel": dom["name"],
        "kernel_symbol": " | ".join(CLASS_KERNELS.get(dom["name"], ("?",))) + "...>  (row of profiles/r03_kernel_stats_*.csv)",
        "traffic": pmc_traffic(dom["name"], summary),
        "traffic_unit": f"HBM bytes per launch (rocprofv3 --pmc pass, profiles/{summary})",
        "alg_bytes_per_launch": round(dom["bytes"] / dom["launches"]), "launches": dom["launches"],
        "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
        "share_of_kernel_time": round(dom["ms"] / total_ms, 4),
        "instrumented_ms_per_step": round(instrumented_ms, 3),
        "alg_tflops": round(tfl, 3), "alg_GBps": round(gbs, 1),
        "mfma_products_per_mac": products, "frac_of_mfma_peak": round(mfma_frac, 4),
        "frac_of_hbm_peak": round(hbm_frac, 4)})
    if images is not None and (args.height, args.width) == (120, 160):
        roof["whole_step_tflops"] = round(TRAIN_FLOPS_PER_IMAGE * images / elapsed / 1e12, 3)
        roof["whole_step_frac_of_f32_mfma_peak"] = round(TRAIN_FLOPS_PER_IMAGE * images / elapsed / 1e12
                                                         / PEAK_F32_MFMA_TFLOPS, 4)
    return roof


def inference_bench(dev, build, h=480, w=640):
    """BASELINE.json configs[3]: eval-mode forward of FCDenseNet67 at the demo size (makeDemoVideo.py:36, test.py:93-94
    call model.forward on .eval() modules), frames resident in HBM, batch 1 (the demo script's) and 16."""
    from sim2real_lane_segment_amd.synthetic import make_batch
    out = {"workload": f"FCDenseNet67 num_cls=4 eval forward (BatchNorm running statistics, no Dropout2d) -> probabilities, "
                       f"3x{h}x{w} synthetic frames resident in HBM; frozen model: weight fragments and folded BN tables "
                       f"built once (rln_set_eval_cache), every conv / activation / softmax of the forward in the timed loop",
           "unit": "frames/sec"}
    x16, _ = make_batch(16, h, w, seed=7, device=dev, work_device=dev)
    # f32: the parity arithmetic (f16x2 operands, fp32 stacks); f32_f16x1: fp32 stacks with ONE f16 operand part (two dense
    # layers per launch; mask agreement with the reference 0.9997, tests/test_gpu_dense3.py); bf16: bf16 stacks + operands
    for storage in ("f32", "f32_f16x1", "bf16"):
        model, eng = build("f32", "f16x1", "bf16x1") if storage == "f32_f16x1" else build(storage)
        model.eval()
        eng.set_eval_cache(True)
        for n in (1, 16):
            x = x16[:n].contiguous()
            for _ in range(3):
                eng.forward(x, training=False)
            torch.cuda.synchronize()
            reps = 20 if n == 1 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.forward(x, training=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[f"{storage}_n{n}"] = {"fps": round(n / dt, 1), "ms_per_batch": round(1000.0 * dt, 3)}
        del model, eng
        torch.cuda.empty_cache()
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
                    help="f32 (default): fp32 activation stacks, split 16-bit MFMA operands at fp32-parity error "
                         "(DESIGN.md 4.1); bf16: the bf16-storage mode of BASELINE.json configs[1] -- activation stacks and "
                         "finalised output gradients of the levels that carry the traffic are bf16 in HBM, plain bf16 MFMA "
                         "operands, fp32 accumulation / statistics / gradient stacks / parameters (rln_set_storage; its "
                         "mask agreement and gradient error are measured in tests/test_gpu_bf16_storage.py)")
    ap.add_argument("--augment", dest="augment", action="store_true", default=True,
                    help="(default) the README baseline command's --augment: uint8 frames resident in HBM go through the "
                         "device transform (HueSaturationValue, RandomSizedCrop, MotionBlur / GaussNoise, Normalize) inside "
                         "every timed step")
    ap.add_argument("--no-augment", dest="augment", action="store_false",
                    help="feed pre-normalised fp32 frames instead (rounds 1-2 did)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other storage mode's line and the inference object")
    ap.add_argument("--no-inference", action="store_true", help="skip the 480x640 inference object")
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

    from sim2real_lane_segment_amd import _lib
    from sim2real_lane_segment_amd.dataManagement.myTransforms import MyTransform
    from sim2real_lane_segment_amd.synthetic import make_batch
    from sim2real_lane_segment_amd.trainer import TrainStepper
    from sim2real_lane_segment_amd.trainingModules.SimpleTrain import SimpleTrainModule

    B = args.batch
    FH, FW = 4 * args.height, 4 * args.width
    if args.augment:
        # BASELINE.json configs[1] / README.md:139 "--augment": full-size uint8 frames + uint8 label masks resident in HBM
        # -> MyTransform(augment=True) on device (rln_augment_u8: HSV jitter, random-sized crop, blur / noise, normalise)
        # INSIDE every timed step; the per-image random parameters are drawn on the host as albumentations draws them
        pool = [make_batch(B, args.height, args.width, seed=42, first_index=(rank * 4 + i) * B, device=dev,
                           frames_u8=True, work_device=dev) for i in range(2)]
        aug = MyTransform(width=args.width, height=args.height, augment=True, device=dev, seed=1234 + rank)
    else:
        pool = [make_batch(B, args.height, args.width, seed=42, first_index=(rank * 4 + i) * B, device=dev)
                for i in range(2)]
        aug = None

    def batch_of(i):
        fx, fy = pool[i % len(pool)]
        return aug(fx, fy) if aug is not None else (fx, fy)

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

    def timed(stepper, steps, warmup, barrier=False):
        for i in range(warmup):
            out = stepper.step(*batch_of(i))
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = stepper.step(*batch_of(i))
        torch.cuda.synchronize()
        if barrier and world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    def build(storage, fwd_arith=None, bwd_arith=None):
        torch.manual_seed(42)
        model = SimpleTrainModule(lr=1e-3, lrRatio=1e3, decay=1e-4, num_cls=4).to(dev)  # random init, FCDenseNet67
        model.train()
        eng = model._rln_sync()
        if storage == "bf16":
            eng.set_storage("bf16")
        if fwd_arith or bwd_arith:
            def parse(a):
                if a in (None, "fp32"):
                    return 0, "bf16"
                t, n = a.split("x")
                return int(n), t
            fp, ft = parse(fwd_arith)
            bp, bt = parse(bwd_arith)
            eng.set_dense_arith(fp, ft, bp, bt)
        return model, eng

    def profile_pass(eng, stepper, steps):
        """The same steps again with HIP events around every kernel class (on the launch stream) -> roofline.  The ~600
        event pairs per step cost a few % of wall time, so they stay out of the region `value` is taken from."""
        _lib.check(_lib.lib().rln_profile_enable(eng.ctx, 1))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            stepper.step(*batch_of(i))
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t1) / steps
        prof = read_profile(eng)
        _lib.check(_lib.lib().rln_profile_enable(eng.ctx, 0))
        return prof, ms

    if args.dtype == "bf16" and (args.fwd_arith or args.bwd_arith):
        print("[bench] --dtype bf16 fixes the arithmetic to one-part bf16 operands", file=sys.stderr)
        sys.exit(2)
    model, eng = build(args.dtype, args.fwd_arith, args.bwd_arith)
    stepper = TrainStepper(eng, lr=1e-3, weight_decay=1e-4, n_buckets=args.buckets, force_collectives=args.force_dist)
    stepper.broadcast_parameters()
    if dist.is_initialized() and dist.get_world_size() != world:
        print(f"[bench] RCCL world size {dist.get_world_size()} != WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(5)
    engine_stepper = stepper
    if args.api == "module":
        stepper = ModuleStepper(model)

    # ---- timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize, no instrumentation ----
    elapsed, out = timed(stepper, args.steps, args.warmup, barrier=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out[0])

    prof = None
    instrumented_ms = None
    module_api = None
    secondary = {}
    single = world == 1 and not args.force_dist
    if args.api == "engine" and single and not args.no_module_api:
        dt, _ = timed(ModuleStepper(model), args.steps, 2)
        module_api = {"value": round(B * args.steps / dt, 2), "unit": "images/sec",
                      "ms_per_step": round(1000.0 * dt / args.steps, 3),
                      "path": "SimpleTrainModule.training_step -> loss.backward() -> FusedAdamW.step()"}
    if not args.no_profile:
        prof, instrumented_ms = profile_pass(eng, stepper, args.steps)
    if single and not args.no_secondary and args.api == "engine" and not (args.fwd_arith or args.bwd_arith):
        # the other storage mode, same workload and region, as an object of the same line
        other = "bf16" if args.dtype == "f32" else "f32"
        model2, eng2 = build(other)
        st2 = TrainStepper(eng2, lr=1e-3, weight_decay=1e-4)
        dt2, out2 = timed(st2, args.steps, args.warmup)
        sec = {"value": round(B * args.steps / dt2, 2), "unit": "images/sec", "dtype": other,
               "ms_per_step": round(1000.0 * dt2 / args.steps, 3), "final_loss": round(float(out2[0]), 5)}
        if not args.no_profile:
            prof2, ims2 = profile_pass(eng2, st2, args.steps)
            sec["roofline"] = roofline_of(prof2, eng2, ims2, None, None, args)
        secondary["storage_" + other] = sec
        del st2, model2, eng2
        torch.cuda.empty_cache()
        if args.dtype == "f32":
            # the same step on the exact-fp32 MFMA family only (v_mfma_f32_16x16x4_f32 everywhere): what the split-operand
            # arithmetic of the headline line is measured against (report an fp32,fp32 line for comparison)
            model3, eng3 = build("f32", "fp32", "fp32")
            st3 = TrainStepper(eng3, lr=1e-3, weight_decay=1e-4)
            n3 = max(5, args.steps // 2)
            dt3, out3 = timed(st3, n3, 2)
            secondary["exact_fp32_mfma"] = {"value": round(B * n3 / dt3, 2), "unit": "images/sec", "dtype": "f32",
                                            "ms_per_step": round(1000.0 * dt3 / n3, 3),
                                            "final_loss": round(float(out3[0]), 5),
                                            "dense_arith": "0 parts: exact fp32 MFMA kernels for every convolution"}
            del st3, model3, eng3
            torch.cuda.empty_cache()
        if not args.no_inference:
            secondary["inference_480x640"] = inference_bench(dev, build)

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
                       "input": (f"uint8 {FH}x{FW} frames + label masks resident in HBM -> MyTransform(augment=True) on device "
                                 "(rln_augment_u8) inside every timed step (README.md:139 --augment)") if args.augment
                       else "pre-normalised fp32 frames resident in HBM (no augmentation in the step)",
                       # the exact arithmetic of this line (dtype names the storage / accumulate class only)
                       "storage": ("bf16 activation stacks + finalised output gradients on the levels with rows >= 40 px, "
                                   "fp32 deep levels / gradient stacks / statistics / parameters") if eng.storage == "bf16"
                       else "fp32 everywhere",
                       "dense_arith": "fwd %sx%d / data-gradient %sx%d / dense weight-gradient %sx%d (where N*H*W >= 2400, "
                                      "else as the data gradient) split-operand 16-bit MFMA products, fp32 accumulation "
                                      "(0 parts = exact fp32 MFMA)" % (eng.dense_arith[1], eng.dense_arith[0],
                                                                      eng.dense_arith[3], eng.dense_arith[2],
                                                                      eng.dense_arith[3], eng.wgrad_parts)},
        }
        if module_api is not None:
            result["module_api"] = module_api
        if prof is not None:
            result["roofline"] = roofline_of(prof, eng, instrumented_ms, images, elapsed, args)
            timed_classes = [q for q in prof if q["launches"] > 0]
            result["kernel_classes"] = [
                {"name": q["name"], "ms_per_step": round(q["ms"] / args.steps, 4),
                 "launches_per_step": q["launches"] // args.steps,
                 "tflops": round(q["flops"] / (q["ms"] * 1e-3) / 1e12, 2) if q["flops"] and q["ms"] else None,
                 "alg_GBps": round(q["bytes"] / (q["ms"] * 1e-3) / 1e9, 1) if q["bytes"] and q["ms"] else None}
                for q in sorted(timed_classes, key=lambda q: -q["ms"])]
        result.update(secondary)
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
