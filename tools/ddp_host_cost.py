"""Host and GPU cost of the data-parallel machinery on ONE GPU (RCCL communicator of world size 1): the bench step with the
gradient hooks off / on, wall time to ISSUE a step and per step, and the time spent inside the hook bodies and in
torch.distributed.all_reduce.  Run on the GPU box:  python tools/ddp_host_cost.py [--batch 6] [--steps 20]"""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("TDX_PERSISTENT_CUS", "224")
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
    import torch

    import bench
    from turbdiff_amd import parallel
    from turbdiff_amd.models.conditioning import Conditioning

    parallel.init_from_env("nccl", force=True)
    dev = torch.device("cuda", 0)
    diff = bench.build_model(dev)
    x, c_local, cell_idx = bench.synthetic_inputs(args.batch, dev)
    C = {Conditioning.Type.CELL_TYPE: c_local}
    md = SimpleNamespace(cell_idx=cell_idx)
    bench.set_mode(diff, args.dtype)
    ddp = parallel.BucketedDataParallel(diff, force=True)
    opt = bench.new_optimizer(diff, args.dtype, bench.LOSS_ELEMENTS(args.batch, cell_idx))
    acc = {"stage": 0.0, "launch": 0.0, "all_reduce": 0.0}

    def wrap(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            acc[key] += time.perf_counter() - t
            return r

        setattr(obj, name, g)

    wrap(ddp, "_stage", "stage")
    wrap(ddp, "_launch", "launch")
    wrap(torch.distributed, "all_reduce", "all_reduce")
    parallel.dist.all_reduce = torch.distributed.all_reduce

    def step(parts):
        t0 = time.perf_counter()
        loss, _ = diff(x, C, md, None)
        t1 = time.perf_counter()
        opt.scale_loss(loss).backward() if hasattr(opt, "scale_loss") else loss.backward()
        t2 = time.perf_counter()
        ddp.finish()
        t3 = time.perf_counter()
        opt.step()
        opt.zero_grad(set_to_none=True)
        t4 = time.perf_counter()
        for k, v in zip(("forward", "backward", "finish", "optimizer"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            parts[k] = parts.get(k, 0.0) + v

    out = {}
    for leg, on in (("hooks_off", False), ("hooks_on", True), ("hooks_off_again", False), ("hooks_on_again", True)):
        ddp.enabled = on
        for _ in range(5):
            step({})
        torch.cuda.synchronize()
        for k in acc:
            acc[k] = 0.0
        parts = {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(parts)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[leg] = {"ms_per_step": 1e3 * (t2 - t0) / args.steps, "host_issue_ms_per_step": 1e3 * (t1 - t0) / args.steps,
                    "host_ms": {k: round(1e3 * v / args.steps, 3) for k, v in parts.items()},
                    "inside_ms": {k: round(1e3 * v / args.steps, 3) for k, v in acc.items()}}
    out["params"], out["buckets"] = len(ddp.params), ddp.bucket_layout()
    print(json.dumps(out, indent=1))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
