#!/usr/bin/env python3
"""Headline benchmark: images/s, HRNet-W32 256x192, batch 64 per GPU, forward (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward of the drop-in ``models.pose_hrnet`` over one batch of synthetic crops that is
already resident in HBM.  Crops are independent, so N GPUs run N independent shards (weak scaling, no
collective on the data path).  Prints ONE JSON line on rank 0.

How the timed region is run:
  * consecutive steps are issued on two alternating HIP streams (--streams, default 2): the model keeps one plan per stream,
    two batches are in flight and overlap on the GPU; `value` / `ms_per_step` / `roofline.whole_step` are of that region,
    `single_stream` in the same line is the same K steps issued one after the other, `config.parallelism` says which;
  * in front of the W warm-up steps the step runs untimed until its time stops improving (settle_clocks: a card that sat idle
    comes back slowly); the {W, K} region timed first, cold, is kept as `config.cold_start`;
  * nothing sits between the warm-up and the timed region (the finiteness check comes after it);
  * the kernel figures (`roofline.achieved/frac`, `families`) come from a per-launch pass over the launch list running alone on
    one stream, event pairs around every launch, the duration of an empty event pair taken off.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GFLOP_PER_IMG = {"hrnet_w32_256x192": 15.29, "rsgnet_w32_256x192_coco": 17.80,
                 "rsgnet_w32_256x192_crowdpose": 17.78, "hrnet_w48_384x288": 70.6, "rsgnet_w48_384x288_coco": 89.9}
PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table


def hip_lib_path():
    from rsgnet_amd import hip
    return hip.lib_path()


def profile_build(path):
    """The library build a committed profile was taken on: tools/gpu_r04_prof.sh writes <tag>_library_build.txt beside the
    summaries of a visit; older evidence has none (None)."""
    tag = os.path.basename(path).split("_")[0]
    side = os.path.join(os.path.dirname(path), tag + "_library_build.txt")
    if not os.path.exists(side):
        return None
    with open(side) as f:
        words = f.read().split()
    return words[words.index("build") + 1] if "build" in words else None


def pmc_traffic(kernel_key, workload="hrnet_w32_256x192"):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/*_pmc_summary.json:
    FETCH_SIZE x2-corrected as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE), or None.  The passes are
    collected on the headline workload only (tools/gpu_pmc.sh); every other workload reports null."""
    import glob
    if workload != "hrnet_w32_256x192":
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f).get(kernel_key)
    if not d:
        return None
    return {"bytes_per_launch": int((d["fetch_mb_per_launch"] + d["write_mb_per_launch"]) * 1e6),
            "fetch_mb": d["fetch_mb_per_launch"], "write_mb": d["write_mb_per_launch"],
            "source": os.path.basename(files[-1]), "source_build": profile_build(files[-1])}
def rocprof_kernel(kernel_prefix, workload, gflop_per_step, launches_per_step):
    """Average duration of a kernel in the committed `rocprofv3 --kernel-trace --stats` summary of the one-stream bench
    command (profiles/*_kernel_stats_one_stream.csv, tools/gpu_r04_prof.sh) and the roofline fraction it implies; None when
    there is none.  Headline workload only."""
    import csv
    import glob
    if workload != "hrnet_w32_256x192":
        return None
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats_one_stream.csv"))
                   if "w48" not in os.path.basename(f) and "rsgnet" not in os.path.basename(f))
    if not files:
        return None
    with open(files[-1]) as f:
        for r in csv.DictReader(f):
            if r["Name"].replace("void ", "").startswith(kernel_prefix):
                us = float(r["AverageNs"]) / 1e3
                tf = gflop_per_step / (launches_per_step * us) * 1e3
                return {"avg_us_per_launch": round(us, 2), "tflops": round(tf, 1), "frac": round(tf / PEAK_TFLOPS, 4),
                        "source": os.path.basename(files[-1]), "source_build": profile_build(files[-1])}
    return None


def pmc_mfma_busy(kernel_prefix, workload="hrnet_w32_256x192"):
    """Matrix-pipe utilisation of a kernel from the committed SQ counter pass (profiles/*_sq_summary.json, tools/gpu_sq.sh +
    tools/sq_summary.py): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- the fraction of SIMD
    cycles AT THE CLOCK THE CHIP HELD in which an MFMA executed (the clock under load is 2.0-2.1 GHz, not the 2.4 GHz
    behind the 2.5 PFLOP/s peak that `frac` is priced against).  Headline workload only."""
    import glob
    if workload != "hrnet_w32_256x192":
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_summary.json")))
    for path in reversed(files):
        with open(path) as f:
            d = json.load(f)
        for k, v in d.items():
            if k.startswith(kernel_prefix) and v.get("mfma_busy_frac") is not None:
                return {"mfma_busy_frac": v["mfma_busy_frac"], "wait_any_frac": v.get("wait_any_frac"),
                        "wait_inst_any_frac": v.get("wait_inst_any_frac"), "source": os.path.basename(path),
                        "source_build": profile_build(path)}
    return None


WORKLOADS = {
    "hrnet_w32_256x192": ("pose_hrnet", 32, (192, 256), 17, 64),
    "rsgnet_w32_256x192_coco": ("pose_rsgnet", 32, (192, 256), 17, 64),
    "rsgnet_w32_256x192_crowdpose": ("pose_rsgnet", 32, (192, 256), 14, 64),     # BASELINE configs[4]
    "hrnet_w48_384x288": ("pose_hrnet", 48, (288, 384), 17, 32),                 # BASELINE configs[2]
    "rsgnet_w48_384x288_coco": ("pose_rsgnet", 48, (288, 384), 17, 32),          # ... with the TRP / SGM head (S = 6912)
}
FLIP_PAIRS = {17: [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]],   # coco.py:92
              14: [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]}                        # crowdpose.py:988


def make_model(tag):
    from rsgnet_amd import make_cfg, models
    from rsgnet_amd.weights import synth_state_dict
    name, width, wh, joints, batch = WORKLOADS[tag]
    cfg = make_cfg(name, width, wh, joints)
    model = getattr(models, name).get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", f"bnstats_{tag}.npz")))
    sd = synth_state_dict(model.state_dict(), 0, bn)
    model.load_state_dict(sd)
    return cfg, model.eval(), sd, wh, batch


def cpu_baseline(tag, cfg, sd, wh, budget_s=20.0, gpu_model=None):
    """The CPU oracle (port of the reference's torch-CPU forward) timed on this box's host cores; the
    same crops then go through the GPU model once, which gives the metric's "heatmap max-abs err"."""
    from oracle import pose_oracle as po
    from rsgnet_amd.weights import synth_input
    # the GPU box gives one GPU's share of the host: 16 cores (more threads than that only thrash)
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(cores)
    b = 4
    x = synth_input(b, wh, 1)
    fwd = (lambda: po.hrnet_forward(sd, cfg, x)) if cfg.MODEL.NAME == "pose_hrnet" else \
          (lambda: po.rsgnet_forward(sd, cfg, x, want_scores=True))
    fwd()
    best, t_end, n = 1e9, time.time() + budget_s, 0
    while n < 5 and time.time() < t_end:
        t0 = time.time()
        fwd()
        best = min(best, time.time() - t0)
        n += 1
    out = {"value": round(b / best, 2), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{n} forwards of batch {b} (best), {tag}, torch {torch.__version__} CPU fp32, eval"}
    # SURVEY 8(d) asks for the CPU path at the metric's own batch too: the same port at B = 64 (32 for the 384x288 configs),
    # one untimed forward, then at most three timed ones inside the rest of the budget
    b2 = 64 if wh[0] * wh[1] <= 192 * 256 else 32
    x2 = synth_input(b2, wh, 2)
    f2 = (lambda: po.hrnet_forward(sd, cfg, x2)) if cfg.MODEL.NAME == "pose_hrnet" else \
         (lambda: po.rsgnet_forward(sd, cfg, x2, want_scores=False))
    f2()
    best2, t_end2, n2 = 1e9, time.time() + budget_s / 2, 0
    while n2 < 3 and (n2 == 0 or time.time() < t_end2):
        t0 = time.time()
        f2()
        best2 = min(best2, time.time() - t0)
        n2 += 1
    out["at_metric_batch"] = {"value": round(b2 / best2, 2), "unit": "images/s", "cores": cores, "batch": b2,
                              "sample": f"{n2} forwards of batch {b2} (best), {tag}"
                                        + ("" if cfg.MODEL.NAME == "pose_hrnet" else ", relation map not materialised")}
    if gpu_model is not None:
        ref = fwd()
        with torch.no_grad():
            got = gpu_model(x.to(next(gpu_model.parameters()).device))
        pairs = [("heatmaps", got, ref)] if torch.is_tensor(got) else \
                [(k, g, r) for k, g, r in zip(("multi_kpt_scores", "kpt_scores", "limbs_scores"), got[:3], ref[:3])]
        out["parity_vs_fp32_cpu"] = {k: {"max_abs_err": float(f"{(g.float().cpu() - r).abs().max().item():.3e}"),
                                         "ref_abs_max": round(float(r.abs().max()), 3),
                                         "ref_std": round(float(r.std()), 3)} for k, g, r in pairs}
        out["parity_vs_fp32_cpu"]["weights"] = "calibrated trained-like weights (heat maps peak near 1): bar 0.05 x ref_std"
        if cfg.MODEL.NAME == "pose_hrnet":
            # BASELINE configs[0], the case the metric's 1e-3 max-abs bar is written for: the reference's own
            # random-init network (torch default init, identity BatchNorm), batch 4, GPU against the fp32 CPU forward
            from rsgnet_amd import models
            from rsgnet_amd.weights import synth_state_dict
            m0 = models.pose_hrnet.get_pose_net(cfg, is_train=False)
            sd0 = synth_state_dict(m0.state_dict(), 0, regime="init")
            m0.load_state_dict(sd0)
            m0 = m0.eval().to(next(gpu_model.parameters()).device)
            m0.compute_dtype = gpu_model.compute_dtype
            r0 = po.hrnet_forward(sd0, cfg, x)
            with torch.no_grad():
                g0 = m0(x.to(next(m0.parameters()).device)).float().cpu()
            out["heatmap_max_abs_err_random_init"] = {"value": float(f"{(g0 - r0).abs().max().item():.3e}"), "bar": 1e-3,
                                                      "ref_abs_max": float(f"{r0.abs().max().item():.3e}"),
                                                      "dtype": gpu_model.compute_dtype,
                                                      "config": "BASELINE configs[0]: random-init HRNet-W32 256x192, batch 4"}
            # ... and the same regime on the path north_star names (backbone -> TRP -> SGM): random-init RSGNet-W32, 17 joints
            if tag == "hrnet_w32_256x192":
                from rsgnet_amd import make_cfg
                cfg1 = make_cfg("pose_rsgnet", 32, (wh[0], wh[1]), 17)
                m1 = models.pose_rsgnet.get_pose_net(cfg1, is_train=False)
                sd1 = synth_state_dict(m1.state_dict(), 0, regime="init")
                m1.load_state_dict(sd1)
                m1 = m1.eval().to(next(gpu_model.parameters()).device)
                m1.compute_dtype = gpu_model.compute_dtype
                r1 = po.rsgnet_forward(sd1, cfg1, x, want_scores=False)
                with torch.no_grad():
                    g1 = m1(x.to(next(m1.parameters()).device))
                out["rsgnet_max_abs_err_random_init"] = {
                    **{k: float(f"{(g.float().cpu() - r).abs().max().item():.3e}") for k, g, r in
                       zip(("multi_kpt_scores", "kpt_scores", "limbs_scores"), g1[:3], r1[:3])},
                    "bar": 1e-3, "dtype": gpu_model.compute_dtype, "config": "random-init RSGNet-W32 256x192 (17 joints), batch 4"}
    return out


def settle_clocks(step, warmup, steps, dist, dev, max_seconds, block=20, patience=5, join=None):
    """Untimed, in front of the contract's warm-up: run the step until its time stops improving.  A card that has sat idle
    (model building, a host-side check, even 5 ms of nothing) comes back slowly: the first 20-step window after 3 s of idle
    time measures 1.64 ms per step against 1.48 ms for every later one, and 5 / 20 / 100 ms of idle time in front of a
    20-step region cost it 1.0 / 1.8 / 2.4 ms (profiles/r03m_step_time_along_a_run.txt).  Five warm-up steps (7 ms) do not
    cover that.  Blocks of `block` steps, each timed on the host around a synchronize; stops after `patience` blocks in a row
    without a new best (0.5 % margin) or after max_seconds; with several ranks, when EVERY rank says so (one 4-byte all-reduce
    per block keeps them in step, so that none waits, idle, at the timed region's barrier).  Returns what goes into the line's
    config.cold_start: the {warm-up, steps} region timed first, the way rounds 1-2 timed `value`, and the blocks run since."""
    if max_seconds <= 0:
        return None
    on_gpu = dev is not None and dev.type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    for _ in range(max(warmup, 1)):
        step()
    sync()
    wall, _ = timed_steps(step, steps, dist, dev, join)
    best, since, hist, t_start = float("inf"), 0, [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        for _ in range(block):
            step()
        sync()
        dt = (time.perf_counter() - t0) / block
        hist.append(dt)
        if dt < best * 0.995:
            best, since = dt, 0
        else:
            since += 1
        go_on = since < patience and time.perf_counter() - t_start < max_seconds
        if dist is not None:
            flag = torch.tensor([1.0 if go_on else 0.0], device=dev if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            go_on = bool(flag.item() > 0)
        if not go_on:
            break
    return {"ms_per_step": round(wall / steps * 1e3, 4), "settle_blocks": len(hist), "steps_per_block": block,
            "settle_block_ms_per_step": {"first": round(hist[0] * 1e3, 4), "best": round(min(hist) * 1e3, 4),
                                         "last": round(hist[-1] * 1e3, 4)}}


def timed_steps(step, steps, dist, dev, join=None):
    """The contract's timed region: barrier + synchronize on both sides of EXACTLY `steps` steps; returns
    (max wall seconds over ranks, this rank's device ms between HIP events on the launch stream)."""
    on_gpu = dev is not None and dev.type == "cuda"

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    e0 = e1 = None
    if on_gpu:
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    if on_gpu:
        e0.record(stream)
    for _ in range(steps):
        step()
    if on_gpu:
        if join is not None:
            join()                      # steps issued on side streams: the timing stream waits for them
        e1.record(stream)
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    dev_ms = e0.elapsed_time(e1) if on_gpu else wall * 1e3
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev if on_gpu else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), dev_ms


def family_table(plan, per_op):
    """Group the launch list by kernel family; FLOPs are the algorithmic conv FLOPs each launch covers."""
    flops = {}
    for e in plan.conv_log:
        if "tile" in e:      # a single convolution
            flops[e["name"]] = 2.0 * plan.batch * e["hw"][0] * e["hw"][1] * e["cin"] * e["cout"] * e["k"] ** 2
        elif "flops" in e:   # a fused op that states its own algorithmic FLOPs (the stem)
            flops[e["name"]] = e["flops"]
        elif e.get("pair"):  # Bottleneck conv2 (3x3, cin -> 64) + conv3 (1x1, 64 -> cout) in one launch
            flops[e["name"]] = 2.0 * plan.batch * e["hw"][0] * e["hw"][1] * (e["cin"] * 64 * 9 + 64 * e["cout"])
    chain = {}
    for e in plan.conv_log:
        if "chain" in e:
            chain[e["name"]] = 2.0 * plan.batch * e["hw"][0] * e["hw"][1] * e["cin"] * e["cout"] * 9 * e["layers"]
        elif "group" in e and e["group"] and isinstance(e["group"][0], (list, tuple)):
            chain[e["name"]] = sum(2.0 * plan.batch * h * w * c * c * 9 * n for c, h, w, n in e["group"]) + e.get("tail_flops", 0.0)
        elif "group" in e:
            flops[e["name"]] = sum(flops.get(n, 0.0) for n in e["group"])
    fams = {}
    for o in per_op:
        n = o["op"]
        if n in chain:
            k, f = "chain_group_kernel/chain_kernel (LDS-resident BasicBlock chains)", chain[n]
        elif ".fuse" in n:
            k, f = "conv_group_kernel (fuse layers)", flops.get(n, 0.0)
        elif n in flops:
            k, f = "conv_igemm / conv_pair / stem kernels (stem, layer1, transitions, heads)", flops[n]
        else:
            k, f = "other (layout, TRP, bilinear)", 0.0
        d = fams.setdefault(k, {"launches": 0, "us": 0.0, "gflop": 0.0})
        d["launches"] += 1
        d["us"] += o["us"]
        d["gflop"] += f / 1e9
    for d in fams.values():
        d["us"] = round(d["us"], 1)
        d["gflop"] = round(d["gflop"], 1)
        d["tflops"] = round(d["gflop"] / d["us"] * 1e3, 1) if d["us"] else 0.0
    # algorithmic HBM bytes of the chain launches: every chain reads its input once and writes its output
    # once (16-bit), plus one pass over its weights
    ck = "chain_group_kernel/chain_kernel (LDS-resident BasicBlock chains)"
    if ck in fams:
        tot = 0.0
        for e in plan.conv_log:
            items = []
            if "chain" in e:
                items = [(e["cin"], e["hw"][0], e["hw"][1], e["layers"])]
            elif "group" in e and e["group"] and isinstance(e["group"][0], (list, tuple)):
                items = e["group"]
            for c, h, w, n in items:
                tot += 2 * plan.batch * h * w * c * 2 + n * c * c * 9 * 2
        fams[ck]["alg_mb_per_launch"] = round(tot / fams[ck]["launches"] / 1e6, 1)
    return fams


def train_bench(a, rank, world, local, dev, dist):
    """BASELINE configs[3]: HRNet-W32 256x192 training step (forward + backward + Adam), 32 crops per GPU (256 over 8),
    gradients all-reduced over RCCL between backward and step.  A step = the reference's loop body
    (lib/core/function.py:46-64) through rsgnet_amd.parallel.DataParallelStep.  Prints its own JSON line."""
    from rsgnet_amd import make_cfg, models
    from rsgnet_amd.loss import JointsMSELoss
    from rsgnet_amd.optim import FusedAdam
    from rsgnet_amd.parallel import DataParallelStep
    from rsgnet_amd.weights import gaussian_targets, limb_targets, synth_input, synth_state_dict
    rsg = a.workload.startswith("rsgnet")            # --workload rsgnet_w32_256x192_coco --train: the step of `rsgnet_train`
    tag = "rsgnet_w32_256x192_coco" if rsg else "hrnet_w32_256x192"
    batch = a.batch or 32
    cfg = make_cfg("pose_rsgnet" if rsg else "pose_hrnet", 32, (192, 256), 17)
    model = (models.pose_rsgnet if rsg else models.pose_hrnet).get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", f"bnstats_{tag}.npz")))
    sd = synth_state_dict(model.state_dict(), 0, bn)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    x = synth_input(batch, (192, 256), seed=100 + rank).to(dev)
    losses = []
    if rsg:
        from rsgnet_amd.loss import relation_target_vector, rsgnet_loss
        crit = JointsMSELoss(True)
        step_fn = DataParallelStep(model, lambda outs, *t: rsgnet_loss(outs, *t, crit), opt)
        target, tw = gaussian_targets(batch, 17, (128, 96), 200 + rank)
        all_t, all_w = gaussian_targets(batch, 17, (128, 96), 300 + rank)
        limbs_t = limb_targets(batch, 18, (128, 96), 400 + rank)
        target, tw, all_t, all_w, limbs_t = (t.to(dev) for t in (target, tw, all_t, all_w, limbs_t))

        def step():
            loss, _ = step_fn(x, target, tw, all_t, all_w, limbs_t, relation_target=relation_target_vector(target))
            losses.append(loss)
    else:
        step_fn = DataParallelStep(model, JointsMSELoss(True), opt)
        target, tw = gaussian_targets(batch, 17, (64, 48), 200 + rank)
        target, tw = target.to(dev), tw.to(dev)

        def step():
            loss, _ = step_fn(x, target, tw)
            losses.append(loss)

    for _ in range(max(a.warmup, 1)):              # (no settle_clocks here: the 13 ms step measures the same cold and settled)
        step()
    torch.cuda.synchronize()
    wall, dev_ms = timed_steps(step, a.steps, dist, dev)
    vals = [float(l.detach()) for l in losses]
    assert all(np.isfinite(vals)), "non-finite loss"
    if rank != 0:
        return
    plan = next(iter(model._train_plans.values()))
    # per-family device time of one step (event pairs around the forward list, the backward list's kernel kinds)
    st = torch.cuda.current_stream(dev).cuda_stream
    fams = {}
    for name_ops, ops in (("forward", plan.fwd), ("backward", plan.bwd)):
        evs = []
        for rec in ops:
            nm, op = rec[0], rec[1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op(st)
            e1.record()
            evs.append((nm, e0, e1))
        torch.cuda.synchronize()
        for nm, e0, e1 in evs:
            if name_ops == "forward":
                kind = "input" if nm == "input" else "batchnorm" if ".bn" in nm else \
                    "fuse sum" if ".fuse." in nm and nm.count(".") == 3 else "conv"
            else:
                kind = "wgrad" if ".wgrad" in nm else "dgrad" if (".dgrad" in nm or ".spread" in nm) else \
                    "batchnorm" if ".bn" in nm else "gradient plumbing"
            d = fams.setdefault(f"{name_ops}: {kind}", {"launches": 0, "us": 0.0})
            d["launches"] += 1
            d["us"] += e0.elapsed_time(e1) * 1e3
    for d in fams.values():
        d["us"] = round(d["us"], 1)
    ms = wall / a.steps * 1e3
    imgs = world * batch * a.steps / wall
    gflop = 3 * GFLOP_PER_IMG[tag]                 # forward + data gradient + weight gradient
    whole = gflop * batch / (dev_ms / a.steps)
    dom_name, dom = max(fams.items(), key=lambda kv: kv[1]["us"])
    line = {"metric": f"images/sec {'RSGNet' if rsg else 'HRNet'}-W32 256x192 training step (fwd+bwd+Adam)", "value": round(imgs, 1), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": tag + "_train" + (" (backbone: the HIP training plan; head: autograd nodes on the library's kernels)" if rsg else ""), "per_gpu_batch": batch, "global_batch": batch * world,
                       "image": "256x192", "parallelism": f"dp{world} (bucketed RCCL all-reduce of the flat fp32 gradient buffer)",
                       "optimizer": "Adam lr 1e-3 (FusedAdam)", "loss_first_last": [round(vals[0], 6), round(vals[-1], 6)]},
            "roofline": {"bound": "mfma", "achieved": round(whole, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(whole / PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": f"whole training step (3 x {GFLOP_PER_IMG[tag]} GFLOP per image)", "device_ms_per_step": round(dev_ms / a.steps, 3),
                         "largest_family": dom_name, "families": fams}}
    if not a.no_cpu_baseline and world == 1 and not rsg:
        cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
        torch.set_num_threads(cores)
        from oracle import train_step_oracle as tso        # the cpu_baseline leg: the only use of oracle/ on this path
        xs, ts, ws = x[:2].cpu(), target[:2].cpu(), tw[:2].cpu()
        t0 = time.time()
        tso.hrnet_train_step(sd, cfg, xs, ts, ws, dtype=torch.float32)
        dt = time.time() - t0
        line["cpu_baseline"] = {"value": round(2 / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"1 forward+backward of batch 2 (no optimizer), torch {torch.__version__} CPU fp32"}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hrnet_w32_256x192", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the metric's 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2,
                    help="issue consecutive batches on this many alternating HIP streams (each has its own plan: arena + graph), "
                         "so that two batches in flight overlap on the GPU (same-visit: 1.54 -> 1.36 ms per batch of 64); the line "
                         "carries the one-stream figure too (`single_stream`).  1 with --flip-test / --pipeline (forward_flip_test overlaps its own two forwards) and --from-host")
    ap.add_argument("--settle-seconds", type=float, default=2.0,
                    help="inference: at most this long, untimed steps of the same workload in front of the warm-up until "
                         "the step time stops improving (settle_clocks; 0: time the cold region, as rounds 1-2 did)")
    ap.add_argument("--no-other-dtype", action="store_true", help="skip the bf16 (fp16 with --dtype bf16) leg of the line (--no-cpu-baseline skips it too)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"],
                    help="16-bit storage / MFMA operand type of the inference plan (fp32 accumulation either way).  fp16 is the "
                         "default because the 1e-3 heat-map bar needs its mantissa (DESIGN.md section 6); bf16 runs the same kernels")
    ap.add_argument("--flip-test", action="store_true",
                    help="a step = flip-tested heatmaps: one forward over [x ; x mirrored] + the fused merge")
    ap.add_argument("--pipeline", action="store_true",
                    help="a step = person boxes -> crops (rsg_crop_warp_affine) -> flip-tested heatmaps -> decoded "
                         "keypoints, from uint8 pictures resident in HBM (SURVEY 8(f) #1, #3, #4 around the forward)")
    ap.add_argument("--from-host", action="store_true",
                    help="a step starts from the fp32 batch in pinned HOST memory (PCIe-inclusive rate; never the "
                         "headline value, which has inputs resident in HBM)")
    ap.add_argument("--per-op", default="", help="write a per-op timing table (JSON) to this path")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[3] instead of the headline: one training step (forward + backward + Adam, "
                         "gradient all-reduce when N > 1), 32 crops per GPU")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an AMD GPU (no CPU fallback path exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)

    if a.train:
        train_bench(a, rank, world, local, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    from rsgnet_amd.weights import synth_input
    cfg, model, sd, wh, batch = make_model(a.workload)
    if a.batch:
        batch = a.batch
    model = model.to(dev)
    model.compute_dtype = a.dtype
    x = synth_input(batch, wh, seed=rank).to(dev)          # resident in HBM before the timed region

    joints = WORKLOADS[a.workload][3]
    if a.pipeline:
        a.flip_test = True
    if a.flip_test:
        from rsgnet_amd.flip import forward_flip_test
    if a.pipeline:
        import numpy as np
        from rsgnet_amd import crops as crops_
        from rsgnet_amd.decode import get_final_preds
        rs = np.random.RandomState(rank)
        pictures = [torch.from_numpy(rs.randint(0, 256, (480, 640, 3)).astype(np.uint8)).to(dev) for _ in range(8)]
        which = [i % 8 for i in range(batch)]
        cs = [crops_.box_to_center_scale([rs.uniform(0, 400), rs.uniform(0, 200), rs.uniform(60, 200),
                                          rs.uniform(120, 280)], wh[0] / wh[1]) for _ in range(batch)]
        centers, scales = np.stack([c for c, _ in cs]), np.stack([s_ for _, s_ in cs])
        producer = crops_.CropProducer(wh)
        x2 = torch.empty(2 * batch, 3, wh[1], wh[0], device=dev)

    x_host = x.cpu().pin_memory() if a.from_host else None

    if a.pipeline or a.from_host or a.flip_test:
        a.streams = 1           # (flip test: forward_flip_test itself runs its two forwards of batch B beside each other;
                                #  two such calls in flight on top measured slower, 3.52 -> 3.91 ms)
    side = [torch.cuda.Stream(dev) for _ in range(a.streams)] if a.streams > 1 else None
    issued = [0]

    def join():
        cur = torch.cuda.current_stream(dev)
        for s_ in side:
            cur.wait_stream(s_)

    def step():
        if side is None:
            return step1()
        issued[0] += 1
        with torch.cuda.stream(side[issued[0] % len(side)]):      # consecutive batches on alternating streams: they may overlap
            return step1()

    def step1():
        with torch.no_grad():
            if a.from_host:
                x.copy_(x_host, non_blocking=True)
            if a.pipeline:
                xin = producer(pictures, which, centers, scales, out=x2, with_flip=True)
                hm = forward_flip_test(model, xin, FLIP_PAIRS[joints], prepared=True)
                return get_final_preds(hm, centers, scales)[0]
            if a.flip_test:
                return forward_flip_test(model, x, FLIP_PAIRS[joints])
            return model(x)

    jn = join if side is not None else None
    cold = settle_clocks(step, a.warmup, a.steps, dist, dev, a.settle_seconds, join=jn)
    for _ in range(max(a.warmup, 1)):
        out = step()
    # nothing between the warm-up and the timed region: the finiteness check (first use: tens of ms of host time with the
    # card idle) sat here until round 3 and cost the 20-step region 2 ms (tools/region_cost.py)
    wall, dev_ms = timed_steps(step, a.steps, dist, dev, jn)
    o0 = out if torch.is_tensor(out) else out[1]
    assert torch.isfinite(o0).all(), "non-finite heatmaps"

    if side is not None:        # the per-launch pass below runs the DEFAULT stream's plan (tuned for a forward running alone;
        torch.cuda.synchronize()                  # the side streams' plans trade makespan for CU-time, plan.py CHAIN_PACK)
        keep_side, side = side, None
        step()                                    # builds / binds it
        side = keep_side
        torch.cuda.synchronize()
    if a.flip_test:             # the flip test runs two forwards of batch B beside each other (rsgnet_amd/flip.py): the pass
        with torch.no_grad():   # below times the launch list of ONE of them, on the default stream's plan
            model(x)
        torch.cuda.synchronize()
    plan = model.plan_for(x)
    plan.check_handoff()        # the in-launch hand-off's error word, read back (raises if any poll gave up)
    # per-launch timing pass (outside the timed region, rank 0): the launch list runs IN ORDER, eagerly, with a HIP
    # event pair around every launch on the launch stream, several passes; each launch therefore meets the caches
    # in the state the real forward leaves them (timing a launch repeated back to back flattered the chain
    # kernels by 22 %: warm weights and inputs).  Gives the kernel-family table and the dominant kernel's roofline;
    # rocprofv3's average duration of the same kernel (profiles/) agrees with it.
    fam = None
    if rank == 0:
        st = torch.cuda.current_stream(dev).cuda_stream
        reps = 10
        for _, op in plan.ops:
            op(st)
        evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan.ops]
               for _ in range(reps)]
        for r in range(reps):
            for (name, op), (e0, e1) in zip(plan.ops, evs[r]):
                e0.record()
                op(st)
                e1.record()
        # what an event pair measures with NOTHING between its two records (the markers' own processing, 5-7 us on this
        # stack: 8 chain launches carried 50 us of it against rocprofv3's durations of the same kernels) is taken off
        empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for z0, z1 in empty:
            z0.record()
            z1.record()
        torch.cuda.synchronize()
        pair_us = sorted(z0.elapsed_time(z1) for z0, z1 in empty)[len(empty) // 2] * 1e3
        # "us": what the event pair around the launch measured; "us_net": less the empty pair.  `roofline.frac` and the family
        # table are priced on the RAW figure (rocprofv3's kernel durations sit between the two: profiles/, `rocprof` below)
        raw = [sum(evs[r][i][0].elapsed_time(evs[r][i][1]) for r in range(reps)) / reps * 1e3 for i in range(len(plan.ops))]
        per_op = [{"op": name, "us": round(raw[i], 2), "us_net": round(max(0.0, raw[i] - pair_us), 2)}
                  for i, (name, _) in enumerate(plan.ops)]
        fam = family_table(plan, per_op)
        if a.per_op:
            with open(a.per_op, "w") as f:
                json.dump({"workload": a.workload, "batch": batch, "ops": per_op, "convs": plan.conv_log,
                           "families": fam}, f)

    if rank == 0:
        ms = wall / a.steps * 1e3
        imgs = world * batch * a.steps / wall
        gflop = GFLOP_PER_IMG[a.workload]
        fwd_imgs = batch * (2 if a.flip_test else 1)            # a flip-tested crop costs two forwards
        whole = gflop * fwd_imgs / (dev_ms / a.steps)            # GFLOP / ms = TFLOP/s over the whole step
        dom_name, dom = max(fam.items(), key=lambda kv: kv[1]["us"])
        ach = dom["gflop"] / (dom["us"] / 1e3)                   # GFLOP / ms = TFLOP/s
        line = {
            "metric": {"hrnet_w32_256x192": "images/sec HRNet-W32 256x192 bs64 fwd"}.get(
                a.workload, f"images/sec {a.workload} bs{batch} fwd") + (" flip-test" if a.flip_test else "")
                + (" boxes->crops->keypoints" if a.pipeline else "") + (" from host memory" if a.from_host else ""),
            "value": round(imgs, 1), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": model.compute_dtype, "data": "synthetic",
            "config": {"workload": a.workload, "per_gpu_batch": batch, "global_batch": batch * world,
                       "image": f"{wh[1]}x{wh[0]}", "parallelism": f"dp{world} (independent shards)" + (f", {a.streams} batches in flight per GPU (alternating streams)" if side is not None else "")
                       + (", the flip test's two forwards of batch B overlap on two streams" if a.flip_test else ""),
                       "weights": "synthetic name-keyed rule + calibrated BN stats", "graph": plan.graph_mode,
                       "launches_per_step": len(plan.ops), "streams": a.streams,
                       # the figures of the same {warm-up, steps} region timed cold, as rounds 1-2 timed `value`, and the
                       # untimed steps run since (settle_clocks)
                       "cold_start": cold},
            # dominant kernel family: algorithmic FLOPs of its launches / their summed durations (HIP events
            # on the launch stream, per-launch pass after the timed region)
            "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic("chain_group_kernel" if "chain" in dom_name else "conv_igemm_kernel", a.workload),
                         "mfma_util": pmc_mfma_busy("chain_group_kernel" if "chain" in dom_name else "conv_igemm_kernel", a.workload),
                         "algorithmic_mb_per_launch": dom.get("alg_mb_per_launch"),
                         "kernel": dom_name, "launches": dom["launches"], "kernel_us_per_step": dom["us"],
                         # an event pair with nothing between its records takes this long; NOT taken off the figures above
                         # (`achieved_net_of_event_pair` shows the family figure with it taken off every launch)
                         "empty_event_pair_us": round(pair_us, 2),
                         "achieved_net_of_event_pair": round(dom["gflop"] / (max(dom["us"] - dom["launches"] * pair_us, 1e-3) / 1e3), 1),
                         "rocprof": rocprof_kernel("chain_group_kernel" if "chain" in dom_name else "conv_igemm_kernel", a.workload, dom["gflop"], dom["launches"]),
                         # the kernel figures (achieved, frac, families) are a property of the kernels: measured with the
                         # launch list running alone on one stream; with 2 batches in flight (whole_step, value) kernels of
                         # both batches share the CUs and a profiler's per-kernel durations grow accordingly
                         "kernel_figures_measured": "launch list alone on one stream (HIP event pair around every launch; rocprofv3 of the --streams 1 command: profiles/*_kernel_stats_one_stream.csv)",
                         "kernel_gflop_per_step": dom["gflop"],
                         "whole_step": {"achieved": round(whole, 1), "frac": round(whole / PEAK_TFLOPS, 4),
                                        "device_ms_per_step": round(dev_ms / a.steps, 4),
                                        "algorithmic_gflop_per_image": gflop},
                         "families": fam},
        }
        mismatch = False
        if side is not None:
            # the same K steps issued one after the other on ONE stream (rounds 1-3 measured `value` this way)
            keep, side = side, None
            for _ in range(max(a.warmup, 1) + 40):
                step()
            w1, d1 = timed_steps(step, a.steps, None, dev)
            # ... and the overlapped forwards must be the one-stream forward bit for bit: 8 more steps on the alternating
            # streams beside each other, every output against the one-stream result of the same input
            ref1 = step()
            ref1 = [t.clone() for t in ((ref1,) if torch.is_tensor(ref1) else ref1) if torch.is_tensor(t)]
            torch.cuda.synchronize()
            side = keep
            outs8 = []
            for _ in range(8):
                o_ = step()
                outs8.append([t for t in ((o_,) if torch.is_tensor(o_) else o_) if torch.is_tensor(t)])
            torch.cuda.synchronize()
            line["overlapped_outputs_bit_identical"] = all(torch.equal(p_, q_) for o_ in outs8 for p_, q_ in zip(o_, ref1))
            line["single_stream"] = {"value": round(batch * a.steps / w1, 1), "unit": "images/s",
                                     "ms_per_step": round(w1 / a.steps * 1e3, 4),
                                     "whole_step_frac": round(gflop * fwd_imgs / (d1 / a.steps) / PEAK_TFLOPS, 4)}
            if not line["overlapped_outputs_bit_identical"]:
                # never seen since the TRP fix (DESIGN 9b'); if it happens the overlapped figure is not a valid one: the line
                # falls back to the one-stream region (world 1; with several ranks the max over ranks is not available here)
                print("bench.py: forwards overlapping on two streams differ from the one-stream forward; reporting the one-stream "
                      "figures", file=sys.stderr)
                line["overlapped_rejected"] = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
                line["value"], line["ms_per_step"] = world * line["single_stream"]["value"], line["single_stream"]["ms_per_step"]
                line["roofline"]["whole_step"]["frac"] = line["single_stream"]["whole_step_frac"]
                line["config"]["streams"] = 1
                line["error"] = "forwards overlapping on two streams are not bit-identical to the one-stream forward"
                mismatch = True
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(a.workload, cfg, sd, wh, gpu_model=model)
        if world == 1 and not (a.no_other_dtype or a.no_cpu_baseline) and not (a.flip_test or a.pipeline or a.from_host):
            # BASELINE.json configs[1]/[2] name bf16: the same workload, steps and timing on the other 16-bit format (a second
            # plan of the same model, measured AFTER and outside the timed region above; `value` stays the --dtype run)
            other = "bf16" if a.dtype == "fp16" else "fp16"
            ref = out.float() if torch.is_tensor(out) else None
            model.compute_dtype = other
            for _ in range(max(a.warmup, 1) + 100):                  # the plan build left the card idle (settle_clocks)
                out2 = step()
            wall2, dev2 = timed_steps(step, a.steps, None, dev, jn)
            model.compute_dtype = a.dtype
            line["other_dtype"] = {"dtype": other, "value": round(batch * a.steps / wall2, 1), "unit": "images/s",
                                   "ms_per_step": round(wall2 / a.steps * 1e3, 4),
                                   "whole_step_frac": round(gflop * batch / (dev2 / a.steps) / PEAK_TFLOPS, 4)}
            if ref is not None and torch.is_tensor(out2):
                line["other_dtype"]["max_abs_vs_" + a.dtype] = float((out2.float() - ref).abs().max())
        from rsgnet_amd import build as _b
        line["library"] = {"build": (_b.info().get("hash") or "")[:16], "path": os.path.relpath(hip_lib_path(), ROOT)}
        print(_b.describe() if not os.environ.get("RSGNET_HIP_LIB") else f"library: {hip_lib_path()}", file=sys.stderr)
        print(json.dumps(line), flush=True)
        if mismatch:            # a wrong result under overlap is a failure of the run, not a footnote of the line
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
