#!/usr/bin/env python3
"""Headline benchmark: Pix2Pix side2side train images/s on synthetic 64x64x4 sprite batches (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU; every rank runs the full train step (G fwd, D fwd x2, losses, both backward passes,
gradient all-reduce over RCCL when N > 1, two Adam updates, weight-copy refresh) on its own shard of
B images (weak scaling).  Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import dataset_utils as DU  # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402
from palette_and_histo_gan_amd import flops as FL        # noqa: E402
from palette_and_histo_gan_amd import parallel as PAR    # noqa: E402

CONFIGS = {
    # name: (model, per-GPU batch, img size, lambda_l1, lambda_hist, palette)
    "c1": ("baseline", 4, 64, 100.0, None, None),
    "c2": ("baseline", 256, 64, 100.0, None, None),
    "c3": ("histogram", 256, 64, 30.0, 1.0, 24),
    "c4": ("indexed", 128, 64, 0.01, None, 24),          # lambda = lambda_segmentation (experiments.ipynb:228)
    "c5": ("histogram", 256, 128, 30.0, 1.0, 24),
}
MFMA_PEAK = {"bf16": 2500.0, "f32": 157.3}      # TFLOP/s dense, MI355X_MICROARCH.md chip-level parameters


def synthetic_batch(rank, B, S, palette):
    rng = np.random.default_rng([47, rank])
    return DU.synthetic_rgba_batch(rng, B, S, palette_size=palette)


def cpu_baseline(model, S, lambda_l1, lambda_hist, budget_s=20.0):
    """The oracle (torch-CPU f32 restatement of the reference graph, TF 2.9.1 is not installable here) timed
    on the host cores, on a bounded sample: B=4 batches (the reference's own batch size, configuration.py:24).
    The ONLY place where bench.py touches oracle/ (oracle/__init__.py)."""
    from oracle import reference_graph as rg
    # the box's CPU share for one GPU is 16 cores; more threads than that only adds contention in the small
    # (B=4) convolutions (256 threads: 70 s per step, measured)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(47)
    Gp = rg.init_params(rg.generator_param_shapes(4, 4), rng, torch.float32)
    Dp = rg.init_params(rg.discriminator_param_shapes(4), rng, torch.float32)
    B = 4
    src, tgt = rg.synthetic_rgba_batch(rng, B, S)
    src, tgt = torch.tensor(src), torch.tensor(tgt)
    gm, gv, dm, dv = (rg.zeros_like_params(Gp), rg.zeros_like_params(Gp), rg.zeros_like_params(Dp), rg.zeros_like_params(Dp))
    steps, t0, t = 0, None, 0
    while True:
        masks = [torch.tensor(rng.integers(0, 2, size=s).astype(np.float32)) for s in rg.dropout_mask_shapes(B, S)]
        out = rg.train_step_rgba(Gp, Dp, src, tgt, masks, lambda_l1, lambda_hist)
        t += 1
        Gp, gm, gv = rg.keras_adam(Gp, out["g_grads"], gm, gv, t)
        Dp, dm, dv = rg.keras_adam(Dp, out["d_grads"], dm, dv, t)
        if t0 is None:           # first step = warm-up
            t0 = time.perf_counter()
            continue
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 200:
            break
    return {"value": round(steps * B / el, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of batch 4 at {S}x{S}, torch-CPU f32 restatement of the reference graph "
                      f"({model} model), {el:.1f}s"}


def _collective_info():
    """what the gradient all-reduce of an N > 1 run actually went through: the driver's SCALE record then shows that RCCL saw N ranks
    (no N > 1 RCCL curve exists from this build's one-GPU boxes)"""
    import torch.distributed as dist
    info = {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MAX_NCHANNELS") if k in os.environ}
    if dist.is_available() and dist.is_initialized():
        info["world_size"] = dist.get_world_size()
        info["backend"] = dist.get_backend()
        try:
            info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # gloo rehearsal on a build without the binding
            info["nccl_version"] = f"unavailable ({type(e).__name__})"
    return info


def kernel_profile(eng, run_step, n_steps=3):
    """Per-entry-point device time with HIP events on the launch stream, one event pair per C-ABI call."""
    records = []
    orig = L.call

    def timed(name, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        orig(name, *args)
        b.record()
        # the fused block (convolution + InstanceNorm + activation) is the same kernel family as p2p_igemm, which dispatches to it
        records.append(("p2p_igemm" if name == "p2p_igemm_norm_act" else name, args, a, b))

    L.call = timed
    E.L.call = timed
    side_was = eng.side.enabled
    eng.side.enabled = eng.side_hist.enabled = False   # serialise everything on the launch stream so that each event pair brackets its kernel
    try:
        run_step()              # untimed: first use of the serialised schedule (allocator pools of the launch stream, lazy code objects)
        torch.cuda.synchronize()
        records.clear()
        for _ in range(n_steps):
            run_step()
        torch.cuda.synchronize()
    finally:
        L.call = orig
        E.L.call = orig
        eng.side.enabled = eng.side_hist.enabled = side_was
    agg = {}
    for name, args, a, b in records:
        key = name
        d = agg.setdefault(key, [0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += 1
    return {k: {"ms_per_step": v[0] / n_steps, "launches_per_step": v[1] / n_steps} for k, v in agg.items()}, records


def issue_profile(eng, run_step, n, replay=True, one_stream=False):
    """How the step was fed (a driver line must say by itself whether the step was host-fed, overlapped or neither).
    Issues n steps WITHOUT synchronising and times the issue loop and the drain separately: `issue_ms_per_step` is what the host
    needs per step (median: the runtime makes the host wait for a whole batch of steps now and then, once its queue is ~50 steps
    deep -- those waits are device time, not host time), `wall_ms_per_step` the device-limited step time of that mode.
    replay=False: Python/ctypes per launch instead of one p2p_replay call; one_stream=True: no side streams, no events."""
    was = (eng.replay_enabled, eng.side.enabled, eng.side_hist.enabled)
    eng.replay_enabled = replay
    if one_stream:
        eng.side.enabled = eng.side_hist.enabled = False
    try:
        for _ in range(3):          # first step of a key is eager, the second is recorded, the third replays
            run_step()
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for _ in range(n):
            run_step()
            ts.append(time.perf_counter())
        torch.cuda.synchronize()
        t_end = time.perf_counter()
    finally:
        eng.replay_enabled, eng.side.enabled, eng.side_hist.enabled = was
    d = np.diff(np.array(ts)) * 1e3
    return {"issue_ms_per_step": round(float(np.median(d)), 4), "issue_ms_per_step_mean": round(float(d.mean()), 4),
            "drain_ms": round(1e3 * (t_end - ts[-1]), 3), "wall_ms_per_step": round(1e3 * (t_end - ts[0]) / n, 4)}


def _host_info():
    info = {"cpus": os.cpu_count()}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-feed-profile", action="store_true", help="skip the host-issue / one-stream measurements after the timed loop")
    ap.add_argument("--no-replay", action="store_true", help="issue every launch from Python (P2P_REPLAY=0) instead of one p2p_replay call per step")
    ap.add_argument("--one-stream", action="store_true", help="no weight-gradient / histogram side streams")
    ap.add_argument("--host-batches", action="store_true", help="hand train_step HOST (numpy) batches: every step uploads its batch over PCIe "
                    "inside the timed region, on a copy stream (dataset_utils.upload_async).  NOT the headline (the product's datasets keep the sprite set in HBM and produce device batches); "
                    "DESIGN.md section 6 quotes this rate beside it")
    ap.add_argument("--no-mfma", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one captured hipGraph (N=1 only). Measured "
                    "slower than eager two-stream launching on ROCm 7.2 (the replay serialises the weight-gradient branch), so off by default")
    ap.add_argument("--detail", default=None, help="write a per-call (entry point, shape) device-time table to this file")
    # RCCL tuning for the gradient all-reduce (N > 1).  xGMI is point-to-point, 7 links x ~153 GB/s per GPU: a single ring moves
    # 2 x 7/8 x 117 MB over one link direction (>= 1.34 ms); more channels light more links.  Unset = RCCL's
    # own choice.  No N > 1 measurement exists from this build's one-GPU boxes: the knobs are here for the 8-GPU node.
    ap.add_argument("--rccl-channels", type=int, default=None, help="NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS")
    ap.add_argument("--rccl-algo", default=None, help="NCCL_ALGO (Ring | Tree)")
    ap.add_argument("--rccl-proto", default=None, help="NCCL_PROTO (Simple | LL | LL128)")
    ap.add_argument("--bucket-mb", type=float, default=None, help="minimum size of a gradient bucket (default 16 MB)")
    args = ap.parse_args()
    if args.rccl_channels:
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
    if args.rccl_algo:
        os.environ["NCCL_ALGO"] = args.rccl_algo
    if args.rccl_proto:
        os.environ["NCCL_PROTO"] = args.rccl_proto
    if args.bucket_mb:
        E.ParamStore.BUCKET_MIN = int(args.bucket_mb * 1024 * 1024 / 4)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal on a one-GPU box (P2P_REHEARSE=1, with P2P_DP_BACKEND=gloo): every rank shares cuda:0, the collectives go
    # through gloo -- same bench code path as the driver's RCCL run, which needs one GPU per rank
    rehearse = os.environ.get("P2P_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    # under torchrun (RANK set) the RCCL process group is created even for one rank, so the collective path is the
    # same code at N = 1, 2, 4, 8
    comm = PAR.init_data_parallel(device, os.environ.get("P2P_DP_BACKEND")) if (world > 1 or "RANK" in os.environ) else None

    model, B, S, lam_l1, lam_hist, palette = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    dtype = L.BF16 if args.dtype == "bf16" else L.F32
    indexed = model == "indexed"
    if indexed:
        eng = E.Pix2PixEngine(1, 256, "softmax", S, dtype, device=device, seed=47, use_mfma=not args.no_mfma)
        src, tgt, _pal = DU.synthetic_indexed_batch(np.random.default_rng([47, rank]), B, S, palette)
    else:
        eng = E.Pix2PixEngine(4, 4, "tanh", S, dtype, device=device, seed=47, use_mfma=not args.no_mfma)
        src, tgt = synthetic_batch(rank, B, S, palette)
    src_d = torch.as_tensor(src).to(device)
    tgt_d = torch.as_tensor(tgt).to(device)
    if args.host_batches:       # the class API also takes host batches (Dataset.from_batches of numpy arrays): PCIe-inclusive rate
        src_d, tgt_d = np.ascontiguousarray(src), np.ascontiguousarray(tgt)
    if args.no_replay:
        eng.replay_enabled = False
    if args.one_stream:
        eng.side.enabled = eng.side_hist.enabled = False

    def run_step_eager():
        if args.host_batches:       # as the model classes do for host batches: upload beside the previous step's kernels
            s_d, t_d = DU.upload_async([src_d, tgt_d], device)
            if indexed:
                return eng.train_step_indexed(s_d, t_d, lam_l1, global_batch=B * world, dp=comm, batch_offset=rank * B)
            return eng.train_step_rgba(s_d, t_d, lam_l1, lam_hist, global_batch=B * world, dp=comm, batch_offset=rank * B)
        if indexed:
            return eng.train_step_indexed(src_d, tgt_d, lam_l1, global_batch=B * world, dp=comm, batch_offset=rank * B)
        return eng.train_step_rgba(src_d, tgt_d, lam_l1, lam_hist, global_batch=B * world, dp=comm, batch_offset=rank * B)

    use_graph = world == 1 and args.graph and not indexed
    if use_graph:
        graphed = eng.graphed_rgba_step(B, lam_l1, lam_hist, global_batch=B)

        def run_step():
            return graphed(src_d, tgt_d)
    else:
        run_step = run_step_eager

    for _ in range(args.warmup):
        run_step()
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = run_step()
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max_scalar(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    result = {
        "metric": "train images/sec (64x64x4 sprites)" if S == 64 else f"train images/sec ({S}x{S}x4 sprites)",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" + (", host batches uploaded inside every step" if args.host_batches else ""),
        "config": {"workload": f"{args.config}: {model} Pix2Pix train step, per-GPU batch {B}, {S}x{S} "
                               + ("palette-index sprites (1 -> 256-way softmax), " f"lambda_seg={lam_l1}" if indexed else
                                  "RGBA sprites, " f"lambda_l1={lam_l1}") + (f", lambda_hist={lam_hist}, palette {palette}" if lam_hist else ""),
                   "global_batch": B * world, "img_size": S, "parallelism": f"dp{world}",
                   "launch": "hipGraph replay" if use_graph else (("one p2p_replay call per step" if world == 1 else "replayed: p2p_replay segments with the collectives in between")
                              if eng._replays else "eager (Python/ctypes per launch)"),
                   "streams": 1 if not eng.side.enabled else (3 if lam_hist else 2),
                   **({"rccl": _collective_info(), "grad_buckets": len(eng.G.buckets)} if world > 1 else {})},
        "losses": [round(float(x), 5) for x in losses.cpu().numpy()],
    }

    if rank == 0:
        flops_img = FL.train_step_flops_per_image(S, 1, 256, indexed=True) if indexed else FL.train_step_flops_per_image(S, 4, 4)
        result["conv_tflops"] = round(flops_img * value / 1e12, 2)
        result["conv_mfma_frac_of_peak"] = round(flops_img * value / 1e12 / (MFMA_PEAK[args.dtype] * world), 4)
        if not args.no_feed_profile and not use_graph:
            # how the step was fed, measured in this process right after the timed loop (rank 0's LOCAL step, no collectives)
            def run_step_feed():
                if indexed:
                    return eng.train_step_indexed(src_d, tgt_d, lam_l1, global_batch=B)
                return eng.train_step_rgba(src_d, tgt_d, lam_l1, lam_hist, global_batch=B)
            n_feed = max(10, min(args.steps, 40))
            feed = {"replayed": issue_profile(eng, run_step_feed, n_feed), "eager": issue_profile(eng, run_step_feed, n_feed, replay=False),
                    "one_stream": issue_profile(eng, run_step_feed, n_feed, one_stream=True), "host": _host_info()}
            result["host_issue_ms_per_step"] = feed["replayed"]["issue_ms_per_step"]
            result["host_issue_ms_per_step_eager"] = feed["eager"]["issue_ms_per_step"]
            result["ms_per_step_eager_issue"] = feed["eager"]["wall_ms_per_step"]
            result["ms_per_step_one_stream"] = feed["one_stream"]["wall_ms_per_step"]
            result["feed"] = feed
        if not args.no_profile:
            # per-call device times of rank 0's LOCAL step (no collectives: the other ranks are already at the barrier)
            def run_step_local():
                if indexed:
                    return eng.train_step_indexed(src_d, tgt_d, lam_l1, global_batch=B)
                return eng.train_step_rgba(src_d, tgt_d, lam_l1, lam_hist, global_batch=B)
            prof, records = kernel_profile(eng, run_step_local)
            result["kernel_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms_per_step"])}
            # every launch of a step on ONE stream with an event pair around it: what the step would take with no overlap at all
            result["serialised_kernel_ms"] = round(sum(v["ms_per_step"] for v in prof.values()), 4)
            result["roofline"] = FL.roofline_for_dominant(prof, records, B, S, args.dtype)
            # every entry point that matters, each against the roofline that bounds it (bf16 / exact-f32 MFMA peak, or HBM)
            result["rooflines"] = FL.rooflines_top(prof, records, args.dtype, k=10)
            # HBM bytes per launch of that kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB; reduced by
            # tools/pmc_traffic.py from separate --pmc runs of this same command and committed under profiles/).  The profile is
            # stamped with a fingerprint of the kernel sources + engine: a profile taken from other code is NOT quoted.
            from palette_and_histo_gan_amd.build import source_fingerprint
            pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.config}_{args.dtype}.json")
            result["roofline"]["traffic"] = None
            if os.path.exists(pmc):
                prof_json = json.load(open(pmc))
                k = prof_json["kernels"].get(result["roofline"]["kernel"])
                if prof_json.get("fingerprint") != source_fingerprint():
                    result["roofline"]["traffic_note"] = "profiles/ counter file was taken from different sources: not quoted"
                elif k and k.get("hbm_bytes_per_launch"):
                    result["roofline"]["traffic"] = round(k["hbm_bytes_per_launch"])
                    result["roofline"]["traffic_unit"] = "bytes/launch (PMC)"
                    result["roofline"]["traffic_launches_profiled"] = k.get("launches_fetch_pass")
            if args.detail:
                FL.write_detail(records, args.detail, n_steps=3, dtype=args.dtype)
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only
            result["cpu_baseline"] = cpu_baseline("baseline" if indexed else model, S, 100.0 if indexed else lam_l1, lam_hist)
        print(json.dumps(result), flush=True)
    if comm is not None:
        comm.barrier()
        comm.destroy()


if __name__ == "__main__":
    main()
