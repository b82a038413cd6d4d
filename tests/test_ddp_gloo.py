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
from sim2real_lane_segment_amd.trainer import BucketedGradReducer, plan_buckets


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
