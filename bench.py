#!/usr/bin/env python3
"""bench.py -- QA-pairs/sec of the arch1 VQA training step on N MI355X (BASELINE.json metric).

One "step" = dataset:next_batch() gather on the device + forward + backward + (N>1: RCCL
all-reduce of the flat gradient) + clamp + RMSprop, B = 512 QA pairs per GPU, on synthetic data
that is resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the CPU baseline (OpenMP oracle) runs on this job's share of the host: at most 16 threads per GPU
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

# BASELINE.json configs[1]: arch1 baseline, VGG fc7 feats, 1000-way answers, batch 512 fp32
WORKLOAD = dict(arch=1, B=512, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000)
# BASELINE.json configs[3] (secondary, --arch 2): arch2 "deeper LSTM + Inception-v3 feats": -num_layers 2,
# -nhimage 2048 (001_prepro_img_inc.lua:82), E = R = 512, T + 2 = 28 encoder steps, weight decay 1e-4
WORKLOAD_ARCH2 = dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000)
N_QUESTIONS, N_IMAGES = 65536, 8192
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense


def flops_per_qa(w):
    """Algorithmic FLOPs per QA pair per training step (SURVEY.md 8d / BASELINE.md section 4):
    3 x forward (fwd + dgrad + wgrad) minus the never-computed gradient to the image input;
    the embedding is a gather (0 FLOP)."""
    E, R, L, I, C, A, T = w["E"], w["R"], w["L"], w["I"], w["C"], w["A"], w["T"]
    if w["arch"] == 2:  # image projection + (T+2) encoder steps + classifier; the lookup is a gather
        per_tok = sum(2 * 4 * R * ((E if l == 0 else R) + R) for l in range(L))
        fwd = (T + 2) * per_tok + 2 * E * I + 2 * A * R
        return 3 * fwd - 2 * E * I, fwd
    per_tok = 0
    for l in range(L):
        inn = E if l == 0 else R
        per_tok += 2 * 4 * R * (inn + R)
    head = 2 * (C * 2 * R * L + C * I + A * C)
    fwd = T * per_tok + head
    # no d/d(image) and no d/d(embedding output beyond the table): drop dX of the image Linear
    return 3 * fwd - 2 * C * I, fwd


def synth_dataset(w, seed, ragged=False):
    rng = np.random.default_rng(seed)
    q = rng.integers(1, w["V"] + 1, (N_QUESTIONS, w["T"]), dtype=np.int32)  # all lengths = T
    lens = np.full(N_QUESTIONS, w["T"], np.int32)
    if ragged:  # secondary case: lengths ~ U{3..T}, right-aligned, 0 = left padding
        lens = rng.integers(3, w["T"] + 1, N_QUESTIONS).astype(np.int32)
        if w["arch"] == 1:
            q[np.arange(w["T"])[None, :] < (w["T"] - lens)[:, None]] = 0
        else:           # arch2 keeps the stored left-aligned rows, 0 = null after the question
            q[np.arange(w["T"])[None, :] >= lens[:, None]] = 0
    img_pos = rng.integers(1, N_IMAGES + 1, N_QUESTIONS, dtype=np.int32)
    ans = rng.integers(1, w["A"] + 1, N_QUESTIONS, dtype=np.int32)
    feats = np.abs(rng.standard_normal((N_IMAGES, w["I"]), dtype=np.float32))  # normalised on device
    return q, lens, img_pos, ans, feats


def cpu_baseline(w, budget_s=12.0, max_steps=24):
    """The oracle (CPU restatement, NOT Torch7) timed on this host: whole training steps of the same
    B-row batch shape (forward + backward + clamp + RMSprop), repeated for about `budget_s` seconds."""
    from oracle import oracle as orc
    d = orc.make_dims(**w)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d)
    if w["arch"] == 2:
        lens = None
    o = orc.Oracle(np.float32)
    small = orc.make_dims(**{**w, "B": 16})
    ts, ls, ims, las = orc.synth_batch(small)
    o.step(small, params, ts, ls if w["arch"] == 1 else None, ims, las, orc.Dropout(1, 0.5, 123, 0))  # warm up threads/pages
    x, m2 = params.copy(), np.zeros_like(params)
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        r = o.step(d, x, tok, lens, img, lab, orc.Dropout(1, 0.5, 123, steps))
        o.rmsprop(x, r["grads"], m2, 3e-4)
        steps += 1
    dt = time.perf_counter() - t0
    cores = min(int(os.environ["OMP_NUM_THREADS"]), len(os.sched_getaffinity(0)))  # threads the oracle actually used
    out = {"value": round(w["B"] * steps / dt, 2), "unit": "QA-pairs/s", "cores": cores, "kind": "port",
           "sample": f"{steps} training steps (forward + backward + clamp + RMSprop) of a B={w['B']} batch, "
                     f"OpenMP C restatement oracle/nvqa_oracle.c, {dt:.1f} s"}
    if w["arch"] == 1:  # BASELINE.json configs[0]: the reference's own CPU-runnable shape, batch 16 (about 2 s)
        ts, ls, ims, las = orc.synth_batch(small)
        xs, ms = params.copy(), np.zeros_like(params)
        n16, t1 = 0, time.perf_counter()
        while n16 < 64 and (n16 == 0 or time.perf_counter() - t1 < 2.0):
            r = o.step(small, xs, ts, ls, ims, las, orc.Dropout(1, 0.5, 123, n16))
            o.rmsprop(xs, r["grads"], ms, 3e-4)
            n16 += 1
        out["batch16"] = {"value": round(16 * n16 / (time.perf_counter() - t1), 2), "unit": "QA-pairs/s", "steps": n16}
    return out


def mean_len(w, ragged):
    """expected question length of a drawn batch (bench datasets: all T, or U{3..T})"""
    return (3 + w["T"]) / 2.0 if ragged else float(w["T"])


def flops_per_qa_actual(w, ragged):
    """FLOPs per QA pair of the work actually done: the LSTM part scales with the question length (arch1: padding
    steps are skipped; arch2 runs every row to the longest question of the batch, i.e. T + 2 steps here)."""
    full, _ = flops_per_qa(w)
    if not ragged or w["arch"] == 2:
        return full
    E, R, L = w["E"], w["R"], w["L"]
    per_tok = sum(2 * 4 * R * ((E if l == 0 else R) + R) for l in range(L))
    return full - 3 * per_tok * (w["T"] - mean_len(w, True))


def bench_one(pkg, w, args, rank, local_rank, world, dist, steps, warmup, ragged=False, bf16=False, roofline=True,
              host_batches=False):
    """One configuration: warmup, `steps` timed steps between barriers, optional per-kernel pass.  host_batches: the
    JdJ-shaped nvqa_step entry (host batch in, three synchronous copies per call) instead of nvqa_step_indices."""
    import torch
    dims = pkg.binding.Dims(*[w[k] for k in ("arch", "B", "T", "V", "E", "R", "L", "I", "C", "A")])
    try:
        # rank: own sample ids and own dropout masks per rank; the parameter seed is common to all ranks
        tr = pkg.trainer.VQATrainer(dims, device=local_rank, seed=123, rank=rank)
    except pkg.binding.NvqaError as e:
        print(f"bench.py rank {rank}/{world}: nvqa_create failed: {e}", file=sys.stderr, flush=True)
        if dist:
            dist.destroy_process_group()
        raise SystemExit(3)
    tr.init_params()  # same on every rank (counter-based)
    if bf16:
        tr.ctx.set_precision(1)
    q, lens, img_pos, ans, feats = synth_dataset(w, 123, ragged)
    tr.load_dataset(q, lens, img_pos, ans, feats, img_norm=True)
    if world > 1:
        ids = [tr.ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        tr.ctx.comm_init(rank, world, ids[0])
    fn = feats / np.sqrt((feats * feats).sum(1, keepdims=True)) if host_batches else None

    def barrier():
        tr.ctx.sync()
        torch.cuda.synchronize(local_rank)
        if dist:
            dist.barrier()

    hb, hb_i = [], [0]
    if host_batches:  # eight host batches assembled BEFORE the timed region (the numpy gather of 8.4 MB of features per batch is
        for _ in range(8):  # the trainer's business, 2-3 ms in Python; what is timed is the library's host-batch entry)
            qi = tr.next_batch()
            hb.append((np.ascontiguousarray(q[qi]), np.ascontiguousarray(lens[qi]) if w["arch"] == 1 else None,
                       np.ascontiguousarray(fn[img_pos[qi] - 1]), np.ascontiguousarray(ans[qi])))

    def one_step():
        if host_batches:
            t_, l_, f_, a_ = hb[hb_i[0] % len(hb)]
            hb_i[0] += 1
            tr.ctx.step(t_, l_, f_, a_, tr._dropout(), want_loss=False)
        else:
            tr.ctx.step_indices(tr.next_batch(), tr._dropout(), want_loss=False)
        tr.rmsprop()

    persistent = tr.ctx.persistent_state()   # does this shape take the persistent LSTM kernels? (reported in the line)
    for _ in range(warmup):
        one_step()
    # `blocks` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + device synchronisation on
    # both sides, MAX over ranks per region; ms_per_step / value come from the MEDIAN region and the line carries the
    # spread (a single 20-step region of a 3 ms step is 61 ms: one number, no error bar -- VERDICT r3)
    dts = []
    for _ in range(max(1, args.blocks)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        barrier()
        dt_b = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt_b], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_b = float(t[0])
        dts.append(dt_b)
    dt = float(np.median(dts))
    loss = tr.ctx.get_loss()

    fl_qa = flops_per_qa_actual(w, ragged)
    peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    value = w["B"] * world * steps / dt
    out = {"value": round(value, 1), "ms_per_step": round(1e3 * dt / steps, 4),
           "timed_blocks": {"blocks": len(dts), "steps_per_block": steps, "statistic": "median",
                            "ms_per_step_min": round(1e3 * min(dts) / steps, 4), "ms_per_step_max": round(1e3 * max(dts) / steps, 4)},
           "flop_per_qa": round(fl_qa), "persistent": persistent,
           "step_mfma_frac": round(value * fl_qa / (world * peak * 1e12), 4), "final_loss": round(loss, 5)}
    if roofline:
        # per-kernel HIP-event timing on the library's stream, in a separate (untimed) pass;
        # every rank runs the steps (the all-reduce needs them), rank 0 records
        # (four profiled steps first, discarded: they create the event pool, and the bracketed launches run 5-10 % long until the
        # host is ahead of the device again; measured: 3 / 8 / 20 profiled steps -> forward launch 0.978 / 0.944 / 0.924 ms)
        nprof = int(os.environ.get("NVQA_BENCH_NPROF", "20"))
        if rank == 0:
            tr.ctx.profile_enable(True)
        for _ in range(4):
            one_step()
        tr.ctx.sync()
        if rank == 0:
            tr.ctx.profile_reset()
        for _ in range(nprof):
            one_step()
        tr.ctx.sync()
        if rank == 0:
            prof = tr.ctx.profile()
            tr.ctx.profile_enable(False)
            # (ride_gemm: FLOPs of the head products that run in the BPTT launch's idle workgroups -- no time of their own)
            ridden = prof.pop("ride_gemm", {"flops": 0.0})["flops"] / nprof
            gemm = {k: v for k, v in prof.items() if v["flops"] > 0 and v["launches"] > 0}
            # the library books full-length FLOPs for the time-batched / recurrent products: scale to the rows that exist
            def achieved(k):
                pv = gemm[k]
                lstm_part = k.startswith("lstm") or k in ("gemm_wgrad", "gemm_dgrad", "gemm_i2h_fwd")
                scale = mean_len(w, ragged) / w["T"] if (ragged and w["arch"] == 1 and lstm_part) else 1.0
                return scale * pv["flops"] / (pv["ms"] * 1e-3) / 1e12
            # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes,
            # + WRITE_SIZE), recorded by tools/pmc_traffic.py under profiles/ -- a profile of the same command, not of this run
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            tj = json.load(open(tpath)) if os.path.exists(tpath) and not ragged and not bf16 and w is WORKLOAD else {}
            # every MFMA-bound phase of the step with its own fraction of the peak (the BPTT phase is ONE entry whatever
            # kernels implement it: the persistent launch, or levels + finishers on the fallback path)
            phases = {}
            for k in gemm:
                ms = gemm[k]["ms"] + (prof.get("lstm_bwd_finish", {"ms": 0})["ms"] if k == "lstm_step_bwd" else 0.0)
                tf = achieved(k) * gemm[k]["ms"] / ms
                phases[k] = {"ms_per_step": round(ms / nprof, 4), "achieved_tflops": round(tf, 2), "frac": round(tf / peak, 4),
                             "launches_per_step": gemm[k]["launches"] // nprof}
                if k == "lstm_step_bwd" and ridden > 0:  # (not counted in any phase's own fraction)
                    phases[k]["note"] = "head products ride in idle workgroups of the persistent launches: see ridden_head_gflop_per_step"
                if k in tj:
                    tb = tj[k]["hbm_bytes_per_launch"] + (tj.get("lstm_bwd_finish", {}).get("hbm_bytes_per_launch", 0) if k == "lstm_step_bwd" and "lstm_bwd_finish" in prof and prof["lstm_bwd_finish"]["launches"] else 0)
                    phases[k]["traffic_bytes_per_launch"] = tb
            # dominant kernel = the phase that takes the most time, no tie-breaking
            dom = max(phases, key=lambda k: phases[k]["ms_per_step"])
            pv = gemm[dom]
            avg_ms = phases[dom]["ms_per_step"] / max(1, phases[dom]["launches_per_step"])
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": phases[dom]["achieved_tflops"],
                               "peak": peak, "unit": "TFLOP/s",
                               "frac": phases[dom]["frac"], "traffic": phases[dom].get("traffic_bytes_per_launch"),
                               "traffic_source": "profiles/traffic.json (rocprofv3 PMC passes of this command, FETCH_SIZE x 2 + WRITE_SIZE; not measured in this run)" if phases[dom].get("traffic_bytes_per_launch") else None,
                               "avg_launch_ms": round(avg_ms, 5),
                               "launches_per_step": phases[dom]["launches_per_step"]}
            out["phases"] = phases
            if ridden > 0:  # head products computed by otherwise idle workgroups INSIDE the persistent LSTM launches (no time of their own)
                out["ridden_head_gflop_per_step"] = round(ridden / 1e9, 3)
            # the HBM-bound sub-kernels in GB/s (algorithmic bytes booked by the library / HIP-event time)
            hbm = {}
            for k in ("rmsprop", "emb_fwd", "emb_bwd", "softmax_ce", "gather_batch", "colsum", "reduce_slabs", "head_prep", "lstm_bwd_finish"):
                v = prof.get(k)
                if v and v["launches"] and v["bytes"] > 0 and v["ms"] > 0:
                    hbm[k] = {"GB_per_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "ms_per_step": round(v["ms"] / nprof, 4),
                              "frac_of_hbm_peak": round(v["bytes"] / (v["ms"] * 1e-3) / 8.0e12, 3)}
            out["hbm_kernels"] = hbm
            out["kernel_ms_per_step"] = {k: round(v["ms"] / nprof, 4) for k, v in prof.items()
                                         if v["launches"]}
    tr.close()
    return out


def vgg_synth_weights(rng):
    """He-scaled random VGG-16 weights in the flat Caffe order (no caffemodel offline)"""
    chans = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]
    parts, cin = [], 3
    for c in chans:
        parts += [rng.standard_normal(c * cin * 9, dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9))), np.zeros(c, np.float32)]
        cin = c
    for k in (25088, 4096):
        parts += [rng.standard_normal(4096 * k, dtype=np.float32) * np.float32(np.sqrt(2.0 / k)), np.zeros(4096, np.float32)]
    return np.concatenate(parts)


def bench_end_to_end(pkg, w, local_rank, steps=2, bf16_vgg=False):
    """BASELINE.json configs[4]: the VGG-16 fc7 extractor in front of the arch1 step (nvqa_step_images: B = 512 host
    images of 3 x 224 x 224 in, features stay on the device, then the usual forward / backward; + RMSprop)."""
    dims = pkg.binding.Dims(*[w[k] for k in ("arch", "B", "T", "V", "E", "R", "L", "I", "C", "A")])
    rng = np.random.default_rng(0)
    v = pkg.binding.Vgg16(local_rank, 1, 224, max_batch=w["B"])
    v.set_weights(vgg_synth_weights(rng))
    if bf16_vgg:
        v.set_precision(1)
    ctx = pkg.binding.Context(dims, local_rank)
    ctx.init_params(123, -0.08, 0.08)
    x = rng.uniform(-120, 130, (w["B"], 3, 224, 224)).astype(np.float32)
    tok = rng.integers(1, w["V"] + 1, (w["B"], w["T"]), dtype=np.int32)
    lens = np.full(w["B"], w["T"], np.int32)
    lab = rng.integers(1, w["A"] + 1, w["B"], dtype=np.int32)
    dr = pkg.binding.Dropout(1, 0.5, 123, 0)

    def one():
        ctx.step_images(v, x, tok, lens, lab, dr)
        ctx.rmsprop_update(3e-4)
    one()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    ctx.close()
    v.close()
    return {"value": round(w["B"] / dt, 1), "unit": "QA-pairs/s", "ms_per_step": round(dt * 1e3, 2),
            "extractor": "bf16 operands, f32 accumulate" if bf16_vgg else "f32",
            "note": "on-the-fly VGG-16 fc7 (30.9 GFLOP per image) + arch1 training step; 308 MB of host images per step (PCIe inside)"}


def bench_vgg(pkg, n=32, iters=4, bf16=False):
    """VGG-16 fc7 extractor (001_prepro_img_vgg.lua), full 224x224 network, synthetic weights: images/s with host
    images in and host features out (nvqa_vgg16_fc7 as the reference script would call it)."""
    v = pkg.binding.Vgg16(0, 1, 224, max_batch=n)
    rng = np.random.default_rng(0)
    v.set_weights(vgg_synth_weights(rng))
    if bf16:
        v.set_precision(1)
    x = rng.uniform(-120, 130, (n, 3, 224, 224)).astype(np.float32)
    v.fc7(x)
    t0 = time.perf_counter()
    for _ in range(iters):
        v.fc7(x)
    dt = (time.perf_counter() - t0) / iters
    v.close()
    tf = 30.93 * n / dt / 1e3
    return {"value": round(n / dt, 1), "unit": "images/s", "batch": n, "ms_per_batch": round(dt * 1e3, 2),
            "mfma_frac": round(tf / (BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS), 4),
            "dtype": "bf16 operands, f32 accumulate" if bf16 else "f32",
            "note": "host images in / host features out (PCIe inside the timed call)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed regions of --steps steps each; ms_per_step is their median, min / max are reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (N = 1 only)")
    ap.add_argument("--ragged", action="store_true", help="secondary case: question lengths ~ U{3..26}")
    ap.add_argument("--arch", type=int, default=1, choices=(1, 2),
                    help="2 = secondary case: arch2 deeper LSTM + Inception feats (BASELINE configs[3])")
    ap.add_argument("--bf16", action="store_true",
                    help="secondary case: nvqa_set_precision(1), dense products on the bf16 matrix cores "
                         "(operands rounded to bf16, f32 accumulate); the headline metric is the f32 run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU) and starts
        # the N ranks as children under torch.distributed.run, exactly as the driver does
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_device = os.environ.get("NVQA_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        # Rehearsal of the multi-rank launcher on a ONE-GPU box (tests/test_gpu_bench_launch.py): every rank runs on device 0
        # and NVQA_RCCL_LIB must name a collective library that accepts several ranks on one device (librccl refuses that):
        # tests/shim/libnccl_shim.so in its shared-memory mode.  The line says so ("rehearsal"); it is not a scaling number.
        if not os.environ.get("NVQA_RCCL_LIB"):
            raise SystemExit("NVQA_BENCH_ONE_DEVICE=1 needs NVQA_RCCL_LIB (librccl refuses two ranks on one device)")
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch  # noqa: F401  (device synchronisation in bench_one)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo carries only the rendezvous, the barrier and the max-over-ranks; the gradient
        # all-reduce is RCCL inside libnvqa
        dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = ge.load_package()
    w = WORKLOAD if args.arch == 1 else WORKLOAD_ARCH2
    res = bench_one(pkg, w, args, rank, local_rank, world, dist, args.steps, args.warmup, ragged=args.ragged,
                    bf16=args.bf16, roofline=not args.no_roofline)
    out = {
        "metric": "QA-pairs/sec training step (batch 512, seq 26)" + ("" if args.arch == 1 else " [arch2, secondary]"), "value": res["value"],
        "unit": "QA-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16 operands, f32 accumulate" if args.bf16 else "f32", "data": "synthetic",
        "config": {"workload": ("arch1 002_train_baseline: V=14773 E=200 R=512 L=2 I=4096 C=1024 A=1000, " if args.arch == 1 else
                                "arch2 002_train_baseline (deeper LSTM + Inception feats): V=14773 E=R=512 L=2 I=2048 A=1000, 28 steps, wd 1e-4, ")
                               + ("lengths U{3..26}" if args.ragged else "all lengths 26") + ", dropout 0.5 on, HBM-resident dataset, RMSprop",
                   "global_batch": w["B"] * world, "seq_len": w["T"], "parallelism": f"dp{world}"},
    }
    out.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step")})
    if one_device:
        out["rehearsal"] = "NVQA_BENCH_ONE_DEVICE=1: all ranks on device 0 through " + os.environ["NVQA_RCCL_LIB"] + " -- launcher rehearsal, not a scaling measurement"
    if world > 1 and rank == 0:
        out["collective_library"] = pkg.binding.load_library().nvqa_comm_library().decode()
    if dist:
        dist.barrier()
    headline = args.arch == 1 and not args.ragged and not args.bf16
    if rank == 0 and world == 1 and headline and not args.no_secondary:
        # the other BASELINE.json configurations that fit one GPU, each with its own roofline fraction (same --steps / --warmup
        # as the headline: with 10 steps the one un-overlapped host enqueue behind the opening barrier weighed 5 % in the host-batch case)
        sec = {}
        def brief(r):
            return {"value": r["value"], "unit": "QA-pairs/s", "ms_per_step": r["ms_per_step"],
                    "ms_per_step_min_max": [r["timed_blocks"]["ms_per_step_min"], r["timed_blocks"]["ms_per_step_max"]],
                    "persistent": r["persistent"], "step_mfma_frac": r["step_mfma_frac"],
                    "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms")}}
        sec["arch1_ragged_U3_26"] = brief(bench_one(pkg, WORKLOAD, args, 0, local_rank, 1, None, args.steps, args.warmup, ragged=True))
        # the reference's own default batch (-batch_size 500, 002_train_baseline.lua:31): a partly filled last row block in both persistent kernels
        sec["arch1_batch500_reference_default"] = brief(bench_one(pkg, dict(WORKLOAD, B=500), args, 0, local_rank, 1, None, args.steps, args.warmup))
        sec["arch2_f32"] = brief(bench_one(pkg, WORKLOAD_ARCH2, args, 0, local_rank, 1, None, args.steps, args.warmup))
        sec["arch2_bf16"] = brief(bench_one(pkg, WORKLOAD_ARCH2, args, 0, local_rank, 1, None, args.steps, args.warmup, bf16=True))
        hb = bench_one(pkg, WORKLOAD, args, 0, local_rank, 1, None, args.steps, args.warmup, roofline=False, host_batches=True)
        sec["arch1_nvqa_step_host_batches"] = {"value": hb["value"], "unit": "QA-pairs/s", "ms_per_step": hb["ms_per_step"],
                                               "note": "JdJ-shaped entry nvqa_step: host batch validated, staged in pinned memory and copied on the side stream into the device set the running step does not read (8 pre-assembled host batches cycled)"}
        sec["vgg16_fc7"] = bench_vgg(pkg)
        sec["vgg16_fc7_bf16"] = bench_vgg(pkg, bf16=True)
        # the same call at a batch that fills the chip in the deep layers too (the reference's -batch_size is a script flag)
        sec["vgg16_fc7_batch128"] = bench_vgg(pkg, n=128, iters=3)
        sec["vgg16_fc7_bf16_batch128"] = bench_vgg(pkg, n=128, iters=3, bf16=True)
        sec["arch1_end_to_end_vgg16"] = bench_end_to_end(pkg, WORKLOAD, local_rank)
        sec["arch1_end_to_end_vgg16_bf16_extractor"] = bench_end_to_end(pkg, WORKLOAD, local_rank, bf16_vgg=True)
        out["secondary"] = sec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
