#!/usr/bin/env python
"""Headline benchmark: mel-frames/sec of the CFM decoder ODE solve (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by the driver as python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
   --gpus N ..., or -- when WORLD_SIZE is not set -- bench.py re-executes itself under torch.distributed.run, one rank
   per GPU, like the reference's train.py:101-102 spawns its ranks itself)

A "step" is ONE pass of the hot path over one batch: CFMDecoder.forward on BASELINE config 2
(31M decoder, B=32 utterances x T=1000 frames synthetic mu/mask, n_timesteps=10 Euler, CFG 3.0), 16-bit MFMA
operands -- f16 by default: the same width as BASELINE's "bf16" and the type for which every parity gate of
tests/test_gpu_parity.py is <= 1e-3 (bf16 measures ~4e-3 and is reported under "other_dtype") --, inputs already
resident in HBM, explicit noise z.  With N GPUs every rank
solves its own 32-utterance batch (utterances are independent units: no data-path collective,
weak scaling); value = frames solved by all ranks / max-over-ranks wall time.

  --ragged   BASELINE config 4 as the headline: the SAME 256 utterances with len ~ U{600..1000} at every N (strong
             scaling), cut by stabletts_amd.sharding.assign_batches into equal-cost length buckets (mean 32 utterances,
             variable count) and dealt to the ranks; a rank solves its buckets back to back inside the timed region
             (N=1: eight buckets); value counts VALID frames only, the line carries the sharder's imbalance / padding.
  Every default line (any N) also carries the legs "ragged" (config 4 as above, one timed pass per step) and, for N > 1,
  "train_ddp" (config 5: DistributedDataParallel around compute_loss, B=64 per rank, gradient all-reduce over RCCL);
  for N = 1 "train_step" is the same step without the process group.
  --dtype    MFMA operand type of the headline line (f16 = the shipping default, parity-gated at 1e-3; bf16 = BASELINE's
             word for "16-bit operands", ~4e-3).  The other type is timed after the headline region ("other_dtype").

Extra objects on the JSON line:
  roofline     -- the dominant kernel class (largest total time, measured live with HIP events on the
                  launch stream during the timed steps): algorithmic FLOPs per launch / mean launch
                  time vs the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline -- the oracle (fp32 PyTorch CPU restatement of the reference) timed on this host's
                  cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/f16, MI355X_MICROARCH.md chip-level table
HBM_PEAK_GBPS = 8000.0         # HBM3E, same table
B_PER_GPU, T_FRAMES, N_STEPS, CFG = 32, 1000, 10, 3.0
PROFILE_STRIDE = 4             # timed region: HIP events around every 4th launch of the dominant kernel class


def algorithmic_flops_per_frame(T, n_evals, B):
    """F_alg(frame) = E*body + prenet*(1 + 1/B)."""
    body = 2.0 * (12320768 + 3072 * T)
    prenet = 2.0 * 4325376
    return n_evals * body + prenet * (1.0 + 1.0 / B)


CPU_THREADS = 32               # one of the two fixed thread counts timed (the other: os.cpu_count()); the faster is reported


def cpu_baseline(sd, cfg_params, all_cores=False):
    """SURVEY.md section 8(d)'s protocol for config-2-sized inputs: time ONE evaluation of the workload's own batch -- B=32 x T=1000,
    the cond and the uncond estimator call of one cfg_wrapper step (flow_matching.py:58-67), prenet recomputed in each as the
    reference does -- and scale by the step count (every Euler step costs the same two evaluations).  The oracle (fp32 torch-CPU
    restatement of the reference) on 32 threads (fixed).  SURVEY's rule is os.cpu_count() threads: on the 256-thread MI355X hosts that
    is ~20x SLOWER (measured in round 6 with this function: 146-148 s against 6.9-8.1 s per step, profiles/r06_s1_bench_sharder.json,
    r06_bench_final.json), so the faster count is the baseline; --cpu-all-cores times both again (adds ~2.5 minutes)."""
    import oracle
    from oracle.inputs import make_inputs
    fs, fc = cfg_params
    inp = make_inputs(B_PER_GPU, T_FRAMES, seed=0)
    small = make_inputs(2, 256, seed=0)
    t_span = oracle.linspace_f32(N_STEPS)
    ncpu = os.cpu_count() or 1
    t32 = max(1, min(CPU_THREADS, ncpu))
    timed = {}
    for threads in sorted({t32, ncpu} if all_cores else {t32}):
        torch.set_num_threads(threads)
        with torch.inference_mode():
            oracle.cfg_wrapper(sd, t_span[0], small["z"], small["mask"], small["mu"], small["c"], fs, fc, CFG)      # warm-up (thread pool, oneDNN primitives)
            t0 = time.perf_counter()
            oracle.cfg_wrapper(sd, t_span[0], inp["z"], inp["mask"], inp["mu"], inp["c"], fs, fc, CFG)
            timed[threads] = time.perf_counter() - t0
    threads = min(timed, key=timed.get)
    t_eval = timed[threads]
    per_solve = t_eval * N_STEPS
    return dict(value=B_PER_GPU * T_FRAMES / per_solve, unit="mel-frames/sec", cores=threads, kind="port",
                seconds_per_cfg_step_by_threads={str(k): v for k, v in timed.items()},
                all_cores_reference=None if all_cores or ncpu <= CPU_THREADS else
                "os.cpu_count() threads measured in round 6 on this host class (256 threads): 146-148 s per cfg_wrapper step = 22 frames/s "
                "(profiles/r06_s1_bench_sharder.json, profiles/r06_bench_final.json); re-measure with --cpu-all-cores",
                sample=f"oracle (fp32 torch-CPU restatement of the reference; prenet recomputed every evaluation as the reference does), "
                       f"the workload's own batch B={B_PER_GPU} x T={T_FRAMES}, cfg={CFG}: ONE cfg_wrapper step (cond + uncond evaluation) timed "
                       f"with {' and '.join(str(k) for k in timed)} torch threads, the faster kept ({threads} threads: {t_eval:.2f} s) and scaled by the "
                       f"{N_STEPS} euler steps (SURVEY 8d); os.cpu_count()={ncpu}")


def train_step_leg(dev, sd, B=64, T=1000, dtype="f16", steps=5, dropout=True, fused_adamw=False):
    """BASELINE config 5 on ONE GPU (train.py:78-82 shape: B=64 utterances per GPU, T <= 1000 ragged, dropout on):
    CFMDecoder.compute_loss forward (native, keeps activations) + loss.backward() (native dgrad / wgrad / attention
    backward) + AdamW step, timed phase by phase (a device sync between phases) and as whole back-to-back steps (one
    sync at the end).  FLOPs: 3 x the forward's algorithmic FLOPs (forward + dgrad + wgrad), prenet included."""
    import oracle  # noqa: F401  (seeded synthetic inputs only)
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.to(dev).train(dropout)
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4, fused=True) if fused_adamw else torch.optim.AdamW(dec.parameters(), lr=1e-4)      # train.py:60
    raw = make_inputs(B, T, seed=0, ragged=True)
    inp = {k: v.to(dev) for k, v in raw.items() if k != "lengths"}
    valid = int(raw["lengths"].sum())
    x1 = make_inputs(B, T, seed=1)["z"].to(dev)
    times = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
    with torch.enable_grad():
        def step(timed):
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            loss, _ = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(dev); t2 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize(dev); t3 = time.perf_counter()
            if timed:
                times["fwd"] += t1 - t0; times["bwd"] += t2 - t1; times["opt"] += t3 - t2
            return loss.detach()
        for _ in range(2):
            step(False)
        losses = [float(step(True)) for _ in range(steps)]
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(steps):                      # whole steps back to back: no host sync inside
            opt.zero_grad(set_to_none=True)
            loss, _ = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
            loss.backward()
            opt.step()
        torch.cuda.synchronize(dev)
        whole = (time.perf_counter() - t0) / steps
    # one evaluation incl. the prenet over the VALID frames (SURVEY 8(d)'s ragged rule: linear terms on sum(len), the attention
    # term on sum(len^2)) -- the kernels skip the work past each utterance's end, so padded frames are not algorithmic work
    lens = raw["lengths"].double()
    fwd_flops = float(2.0 * ((12320768 + 4325376) * lens.sum() + 3072 * (lens * lens).sum()))
    fwd_flops_padded = 2.0 * (12320768 + 3072 * T + 4325376) * B * T
    fb = (times["fwd"] + times["bwd"]) / steps
    res = {"workload": f"BASELINE config 5 on one GPU: compute_loss forward + backward + AdamW, B={B} x T={T} ragged "
                       f"({valid} valid frames), per-item t, dropout {'0.1' if dropout else 'off (eval mode)'}, {dtype} operands",
           "ms_forward": times["fwd"] / steps * 1e3, "ms_backward": times["bwd"] / steps * 1e3,
           "ms_optimizer_incl_repack": times["opt"] / steps * 1e3, "ms_step_back_to_back": whole * 1e3,
           "mel_frames_per_sec_step": valid / whole, "tflops_fwd_bwd_3x_forward": 3 * fwd_flops / fb / 1e12,
           "frac_of_mfma_peak": 3 * fwd_flops / fb / 1e12 / MFMA_PEAK_TFLOPS,
           "flops_basis": "valid frames: sum(len), sum(len^2)", "tflops_if_padded_frames_counted": 3 * fwd_flops_padded / fb / 1e12,
           "loss_first_last": [losses[0], losses[-1]],
           "torch_GB": torch.cuda.max_memory_allocated(dev) / 1e9, "engine_GB": dec.estimator.engine().device_bytes() / 1e9,
           "optimizer": "torch.optim.AdamW(fused=True)" if fused_adamw else "torch.optim.AdamW(params, lr) as train.py:60 constructs it (foreach)"}
    if not fused_adamw:
        # the same steps with torch's single-kernel AdamW: a one-keyword change to train.py:60, same update rule; reported beside
        # the reference's construction, which stays the figure of this leg
        opt = torch.optim.AdamW(dec.parameters(), lr=1e-4, fused=True)
        with torch.enable_grad():
            def one():
                opt.zero_grad(set_to_none=True)
                loss, _ = dec.compute_loss(x1, inp["mask"], inp["mu"], inp["c"])
                loss.backward()
                opt.step()
            for _ in range(2):
                one()
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize(dev)
        res["ms_step_back_to_back_fused_adamw"] = (time.perf_counter() - t0) / steps * 1e3
    del dec, opt
    return res


# Variables the engine reads (getenv in csrc/engine*.cpp, stabletts_amd/_lib.py) that change which kernels run or which library is loaded;
# tests/test_cabi_cpu.py checks this list against the sources.  Unknown ST_* names are refused too (a new switch must be classified).
ENGINE_ENV = ("STABLETTS_HIP_LIB", "ST_BIG_MIN_BLOCKS", "ST_PHASED", "ST_FUSED_FFN", "ST_RAGGED_SKIP", "ST_SKIP_CLASSES", "ST_QKV_WS",
              "ST_QKV_WS_MIN_TILES", "ST_OPROJ_WS", "ST_OPROJ_WS_MIN_TILES", "ST_QKV_RC1", "ST_SMALL_GRID", "ST_FUSE_SILU", "ST_FUSE_TRAIN_LN",
              "ST_TRAIN_SIDE", "ST_TRAIN_VLO")
# ... and those that do not: ST_SPLIT / ST_HIP_GRAPH change how the same kernels are enqueued (bitwise identical, tests/test_gpu_engine.py),
# ST_BUILD_* are read by stabletts_amd.build only (a left-over from a build step must not cost the harness its line)
ENGINE_NEUTRAL_ENV = ("ST_SPLIT", "ST_HIP_GRAPH", "ST_BUILD_OUT", "ST_BUILD_DEFS")
DDP_WATCHDOG_S = 240           # bench.py --gpus N: seconds the config-5 DDP-over-RCCL leg may take before the line is printed without it
RAGGED_UTTERANCES = 256        # BASELINE config 4: "batch=256 utterances sharded"; the same set at every N (strong scaling)


def ragged_batches(world, rank, dev, n_utt=RAGGED_UTTERANCES):
    """BASELINE config 4: n_utt utterances, len ~ U{600..1000} (fixed seed: identical on every rank and at every world size),
    cut into equal-cost length buckets (mean 32, variable count) and dealt to the ranks by the sharder.  Returns this rank's
    buckets as device-resident input dicts (longest first) and the sharding figures of the whole assignment."""
    import numpy as np
    from oracle.inputs import make_inputs
    from stabletts_amd import sharding
    lengths = np.random.default_rng(4).integers(600, T_FRAMES + 1, size=n_utt).tolist()
    per_rank = sharding.assign_batches(lengths, B_PER_GPU, world)
    mine = []
    for b in sorted(per_rank[rank], key=lambda b: -max(lengths[i] for i in b)):
        bl = [lengths[i] for i in b]
        inp = make_inputs(len(b), max(bl), seed=1000 + b[0], lengths=bl)
        mine.append({"g": {k: v.to(dev) for k, v in inp.items() if k != "lengths"}, "B": len(b), "T": max(bl), "valid": sum(bl)})
    imb, pad = sharding.imbalance(lengths, per_rank)
    fixed = sharding.assign_batches(lengths, B_PER_GPU, world, equal_cost=False)
    info = {"utterances": n_utt, "valid_frames": sum(lengths), "imbalance_max_over_mean": imb, "padded_over_valid_frames": pad,
            "batches_per_rank": [[len(b) for b in bs] for bs in per_rank],
            "batch_max_len_per_rank": [[max(lengths[i] for i in b) for b in bs] for bs in per_rank],
            "cost_model_speedup_over_one_rank": sharding.scaling_ceiling(lengths, per_rank),
            "fixed_count_buckets_imbalance": sharding.imbalance(lengths, fixed)[0],
            "policy": "equal-cost, variable-count length buckets (cost = count x len_max x (12,320,768 + 3,072 len_max)), LPT deal"}
    return mine, info, lengths


def ragged_leg(dec, kw, world, rank, dev, sync, allreduce_max, reps, n_utt=RAGGED_UTTERANCES):
    """Config 4 as a leg of the default line: every rank solves its buckets back to back, `reps` timed passes between two
    barriers, max over ranks.  At N = 1 the eight buckets are also timed one by one: their measured times give the 8-GPU
    ceiling this assignment has on real hardware (sum / max), beside the cost model's."""
    mine, info, _ = ragged_batches(world, rank, dev, n_utt)

    def solve(b):
        g = b["g"]
        return dec(g["mu"], g["mask"], N_STEPS, 1.0, g["c"], "euler", kw, z=g["z"])
    for b in mine:
        solve(b)
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for b in mine:
            solve(b)
    sync()
    sec = allreduce_max(time.perf_counter() - t0) / reps
    res = {"workload": f"BASELINE config 4: the same {n_utt} ragged utterances len~U{{600..1000}} at every N (strong scaling), equal-cost length "
                       f"buckets dealt to the ranks, n_timesteps={N_STEPS} euler, cfg={CFG}; valid frames only",
           "value": info["valid_frames"] / sec, "unit": "mel-frames/sec", "ms_per_pass": sec * 1e3, "scaling": "strong", "sharding": info}
    if world == 1:
        per = []
        for b in mine:
            solve(b)
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            for _ in range(3):
                solve(b)
            torch.cuda.synchronize(dev)
            per.append((time.perf_counter() - t1) / 3 * 1e3)
        res["ms_per_bucket"] = per
        res["bucket_B_x_T"] = [[b["B"], b["T"]] for b in mine]
        if len(per) >= 2:
            res["measured_ceiling_one_bucket_per_gpu"] = {"gpus": len(per), "speedup": sum(per) / max(per), "max_over_mean": max(per) / (sum(per) / len(per))}
    return res


class _LossModule(torch.nn.Module):
    """What StableTTS.forward does with the decoder (models/model.py:173): calls compute_loss.  DDP prepares its gradient hooks
    inside ITS forward, so compute_loss has to run under a module's forward for the reducer to see the backward."""
    def __init__(self, dec):
        super().__init__()
        self.dec = dec

    def forward(self, x1, mask, mu, c):
        return self.dec.compute_loss(x1, mask, mu, c)


def train_ddp_leg(dev, sd, world, rank, dist, share_gpu, B=64, T=T_FRAMES, dtype="f16", steps=5):
    """BASELINE config 5: train.py:49-51,78-81 around the native decoder -- DistributedDataParallel(compute_loss), B utterances
    per rank (ragged, T <= 1000, dropout on), AdamW; the gradient all-reduce is DDP's, over RCCL ("nccl" backend) with one
    GPU per rank.  Ranks sharing a device (BENCH_SHARE_GPU=1, tests) use gloo: RCCL refuses two ranks on one device."""
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    backend = "gloo" if share_gpu else "nccl"
    pg = dist.new_group(backend=backend)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.to(dev).train(True)
    ddp = torch.nn.parallel.DistributedDataParallel(_LossModule(dec), device_ids=[dev.index], process_group=pg)
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
    raw = make_inputs(B, T, seed=100 + rank, ragged=True)
    inp = {k: v.to(dev) for k, v in raw.items() if k != "lengths"}
    x1 = make_inputs(B, T, seed=200 + rank)["z"].to(dev)
    grad_bytes = sum(p.numel() for p in dec.parameters() if p.requires_grad) * 4

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = ddp(x1, inp["mask"], inp["mu"], inp["c"])
        loss.backward()
        opt.step()
        return loss
    with torch.enable_grad():
        for _ in range(3):              # DDP rebuilds its buckets in arrival order after the first step
            step()
        torch.cuda.synchronize(dev); dist.barrier(group=pg); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize(dev); dist.barrier(group=pg); torch.cuda.synchronize(dev)
        sec = (time.perf_counter() - t0) / steps
    tt = torch.tensor([sec, float(raw["lengths"].sum()), float(loss.detach())], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=pg)
    sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=pg)
    sec, valid = float(mx[0]), float(sm[1])
    res = {"workload": f"BASELINE config 5: DistributedDataParallel around CFMDecoder.compute_loss, B={B} x T={T} ragged per rank, dropout 0.1, "
                       f"AdamW, {dtype} operands, native forward / backward; {steps} steps between barriers, max over ranks",
           "ms_per_step": sec * 1e3, "mel_frames_per_sec": valid / sec, "ranks": world, "backend": backend,
           "rccl_ranks": world if backend == "nccl" else 0, "allreduce_bytes_per_step": grad_bytes,
           "mean_loss_last_step": float(sm[2]) / world, "scaling": "weak"}
    del ddp, dec, opt
    return res


# kernel that implements each profiled class on the default path (for the PMC traffic lookup)
CLASS_KERNEL = {
    "ffn_conv1": "conv_gemm_phased3_kernel<st::Op{DT}, 0, false>",
    "ffn_conv2": "ffn_fused_kernel<st::Op{DT}, 0, 0>",        # the whole FFN since round 4 (conv_1 + SiLU + conv_2 in one launch); f16: CLASS_KERNEL_F16
    "lsc_conv": "conv_gemm_phased3_kernel<st::Op{DT}, 1, true>",
    "attention": "attention_kernel<st::Op{DT}, false, false",   # (inference, single-operand scores; prefix: the template has grown parameters)
    "qkv_rope": "qkv_ws_kernel<st::Op{DT}, 0>",               # weight-stationary persistent kernel (qkv_ws.hip)
    "out_proj": "oproj_ws_kernel<st::Op{DT}>",                # weight-stationary persistent kernel (oproj_ws.hip)
}


CLASS_KERNEL_F16 = {"ffn_conv2": "ffn_wino_kernel<0>"}       # opt-in (ST_FUSED_FFN=3, f16 only): the fused FFN on Winograd F(2,3) (ffn_wino.h)


def class_kernel(cls, dtype):
    if dtype != "bf16" and os.environ.get("ST_FUSED_FFN") == "3" and cls in CLASS_KERNEL_F16:
        return CLASS_KERNEL_F16[cls]
    return CLASS_KERNEL.get(cls, "").replace("{DT}", "BF16" if dtype == "bf16" else "F16")


def _pmc_table():
    """Committed rocprofv3 PMC passes of this same command (newest round first)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic_before_wino.json", "r03_pmc_traffic.json", "r02_pmc_traffic_v2.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name))), name
        except Exception:
            continue
    return None, None


def pmc_source():
    """Which committed table the `traffic` figures come from, and whether it was taken from the kernels this run executes: the
    table's `csrc_digest` (written by tools/rocprof_summary.py) against the digest of the sources the loaded library was built from."""
    table, name = _pmc_table()
    if table is None:
        return None
    from stabletts_amd import build
    now = build._digest()[:16]
    was = table.get("_summary", {}).get("csrc_digest")
    return {"file": "profiles/" + name, "csrc_digest_of_table": was, "csrc_digest_now": now,
            "stale": (None if was is None else was != now)}


def pmc_solve_bytes():
    """HBM bytes per solve over ALL kernels (same PMC passes; "_summary" of the table)."""
    table, _ = _pmc_table()
    try:
        return table["_summary"]["hbm_bytes_per_solve"]
    except Exception:
        return None


def pmc_traffic(cls, dtype):
    """HBM bytes per launch of the class's kernel, from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when not available."""
    table, _ = _pmc_table()
    if table is None:
        return None
    want = class_kernel(cls, dtype)
    for name, v in table.items():
        if want and want in name and "fetch_bytes_per_launch" in v:
            return v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
    return None


def main():
    global N_STEPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="cpu_baseline: also time the oracle with os.cpu_count() threads (SURVEY 8d's rule; ~2.5 min on a 256-thread host, where it is ~20x slower than 32 threads)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other-dtype and config-1 latency legs")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the config-5 training-step leg of the extras")
    ap.add_argument("--ragged", action="store_true", help="BASELINE config 4: ragged utterances through the sharder")
    ap.add_argument("--ragged-utterances", type=int, default=RAGGED_UTTERANCES, help="size of the config-4 utterance set (256 = BASELINE; tests use fewer)")
    ap.add_argument("--train-batch", type=int, default=64, help="utterances per rank of the config-5 legs (64 = BASELINE; tests use fewer)")
    ap.add_argument("--n-timesteps", type=int, default=N_STEPS,
                    help="Euler steps per solve: 10 = BASELINE config 2 (default, the headline metric); 50 = config 3, "
                         "the long-ODE stress case")
    ap.add_argument("--dev-env", action="store_true",
                    help="developer runs only: accept engine-changing ST_* / STABLETTS_HIP_LIB variables (the line is then marked "
                         "'dev_env' and is NOT a headline measurement)")
    args = ap.parse_args()
    N_STEPS = args.n_timesteps
    # The headline line describes the library AS SHIPPED: refuse to run with any variable that changes which kernels run or what
    # they return.  (ST_SPLIT / ST_HIP_GRAPH only change how the same kernels are enqueued -- results are bitwise identical,
    # tests/test_gpu_engine.py -- and the profiling scripts set ST_SPLIT=1 for per-kernel passes.)
    dev_env = {k: v for k, v in os.environ.items() if k in ENGINE_ENV or (k.startswith("ST_") and k not in ENGINE_NEUTRAL_ENV)}
    if dev_env and not args.dev_env:
        raise SystemExit("bench.py: refusing to measure with engine-changing variables set: " + ", ".join(sorted(dev_env)) +
                         " (unset them, or pass --dev-env for a developer run that is marked as such)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) and pass the JSON line of
        # rank 0 through.  Rendezvous on 127.0.0.1 with a free port.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the native path has no CPU fallback)")
    # BENCH_SHARE_GPU=1 (test hook): all ranks use the visible devices round-robin, so the N>1 path can be
    # exercised on a 1-GPU box
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if os.environ.get("BENCH_SHARE_GPU") == "1" else local_rank
    if dev_index >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} HIP device(s) visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The inference path shards independent utterances: there is NO data-path collective.  The only
        # cross-rank traffic is the measurement protocol (barrier + max of the elapsed time), a few bytes on
        # the host, so it runs over gloo; RCCL ("nccl" backend) is reserved for paths with a real exchange
        # step (training gradients).  BENCH_SYNC_BACKEND=nccl forces RCCL for the sync.
        backend = os.environ.get("BENCH_SYNC_BACKEND", "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import oracle
    from oracle.inputs import make_inputs
    from stabletts_amd.flow_matching import CFMDecoder
    from stabletts_amd import sharding

    print(f"[bench] rank {rank}/{world} cpu_count={os.cpu_count()} affinity={len(os.sched_getaffinity(0))} "
          f"torch_threads={torch.get_num_threads()}", file=sys.stderr, flush=True)
    sd = oracle.make_state_dict(1234)
    fs, fc = oracle.make_cfg_params(4321)
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=args.dtype)
    dec.estimator.load_state_dict(sd)
    dec = dec.to(dev)

    # workload: a list of device-resident batches this rank solves back to back in one step
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    kw = dict(fake_speaker=fs.to(dev), fake_content=fc.to(dev), cfg_strength=CFG)
    if args.ragged:             # config 4 as the headline: the same utterances at every N, equal-cost buckets
        batches, shard_info, _ = ragged_batches(world, rank, dev, args.ragged_utterances)
        valid_frames_total = shard_info["valid_frames"]
        if not batches:
            raise SystemExit(f"rank {rank}: no bucket (fewer buckets than ranks)")
    else:                       # config 2: one all-ones batch of 32 x 1000 per rank (independent units, weak scaling)
        inp = make_inputs(B_PER_GPU, T_FRAMES, seed=rank)
        batches = [{"g": {k: v.to(dev) for k, v in inp.items() if k != "lengths"}, "B": B_PER_GPU, "T": T_FRAMES, "valid": B_PER_GPU * T_FRAMES}]
        valid_frames_total = B_PER_GPU * T_FRAMES * world
        shard_info = {"utterances": B_PER_GPU * world, "valid_frames": valid_frames_total, "imbalance_max_over_mean": 1.0,
                      "padded_over_valid_frames": 1.0, "batches_per_rank": [[B_PER_GPU]] * world}
    g, T_batch, B_batch = batches[0]["g"], batches[0]["T"], batches[0]["B"]      # the rank's first (longest) batch: survey / roofline sampling

    def solve(gg):
        return dec(gg["mu"], gg["mask"], N_STEPS, 1.0, gg["c"], "euler", kw, z=gg["z"])

    def step_all():
        for b in batches:
            out = solve(b["g"])
        return out

    def step():
        return solve(g)

    eng = dec.estimator.engine()
    for _ in range(args.warmup):
        step_all()
    heavy = ["ffn_conv1", "ffn_conv2", "attention", "qkv_rope", "lsc_conv", "out_proj"]
    # Survey pass (untimed): every heavy class timed with HIP events around every launch -> class breakdown and
    # the dominant class.  An event pair costs ~10 us of idle stream time (rocprofv3 kernel trace: 3.5 ms per
    # solve with all six classes instrumented), so the TIMED region below instruments the dominant class only
    # and samples every 4th launch of it: the roofline's launch duration is still measured live inside the
    # timed steps, on the launch stream, at <0.2 ms of overhead per solve.
    # The survey runs as ONE launch sequence (ST_SPLIT=1): with the default two-part solve an event pair brackets the
    # other part's concurrent kernels as well and the class times would add up to far more than the solve.
    split_env = os.environ.get("ST_SPLIT")
    os.environ["ST_SPLIT"] = "1"
    step()
    eng.profile_enable(True, heavy)
    torch.cuda.synchronize(dev)
    ts = time.perf_counter()
    step()
    torch.cuda.synchronize(dev)
    survey_ms = (time.perf_counter() - ts) * 1e3
    survey = eng.profile_read()
    eng.profile_enable(False)
    dom = max(heavy, key=lambda k: survey[k]["total_ms"])
    if split_env is None:
        del os.environ["ST_SPLIT"]
    else:
        os.environ["ST_SPLIT"] = split_env
    step()
    # The timed region below runs the solve the way the library does by default: for a batch this large as TWO
    # half-batch launch sequences on two streams (st_cfm_solve, ST_SPLIT), so that one part's MFMA-bound K loops run
    # under the other part's HBM-bound epilogues.  A launch's event-bracketed duration then includes kernels of the
    # other part; the dominant kernel's own duration (the roofline figure) is therefore sampled in a second loop of
    # K single-sequence solves (ST_SPLIT=1) right after the timed region -- same process, same tensors, same kernel.

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def allreduce_max(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    if len(batches) > 1:
        step_all()              # back in the multi-shape rhythm (the survey ran one shape)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step_all()
    sync()
    elapsed = allreduce_max(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    os.environ["ST_SPLIT"] = "1"
    step()
    eng.profile_enable(True, [dom], stride=PROFILE_STRIDE)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    single_seq_ms = (time.perf_counter() - t1) / args.steps * 1e3
    prof = eng.profile_read()
    eng.profile_enable(False)
    if split_env is None:
        del os.environ["ST_SPLIT"]
    else:
        os.environ["ST_SPLIT"] = split_env

    frames_total = valid_frames_total * args.steps
    value = frames_total / elapsed

    def time_variant(d, gg, n_steps, kwv, reps):
        for _ in range(2):
            d(gg["mu"], gg["mask"], n_steps, 1.0, gg["c"], "euler", kwv, z=gg["z"])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(reps):
            d(gg["mu"], gg["mask"], n_steps, 1.0, gg["c"], "euler", kwv, z=gg["z"])
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t1) / reps

    extras = {}
    if not args.no_extras and not args.ragged:
        # legs every rank takes part in: config 4 (strong scaling of one 256-utterance workload) and, for N > 1, config 5 (DDP)
        extras["ragged"] = ragged_leg(dec, kw, world, rank, dev, sync, allreduce_max, max(2, args.steps // 3), args.ragged_utterances)
    if rank == 0 and world == 1 and not args.no_extras:
        # (a) the other MFMA operand type on the same workload (f16 is the parity-gated configuration: it meets
        #     north_star's 1e-3 on the displacement metric; bf16 is BASELINE's named dtype and measures ~4e-3)
        other = "f16" if args.dtype == "bf16" else "bf16"
        dec2 = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=other)
        dec2.estimator.load_state_dict(sd)
        dec2 = dec2.to(dev)
        sec = time_variant(dec2, g, N_STEPS, kw, max(3, args.steps // 2))
        extras["other_dtype"] = {"dtype": other, "workload": f"rank 0's first batch, {B_batch} x {T_batch}", "ms_per_step": sec * 1e3, "value": batches[0]["valid"] / sec,
                                 "unit": "mel-frames/sec"}
        del dec2
        # (a2) the opt-in split-precision attention operands (attention_precision="split": q, k as hi + lo pairs, 3x the QK^T MFMAs) on the
        #      same workload: what the mode costs -- it is for checkpoints in the arg-max regime, not the default
        dec3 = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=args.dtype, attention_precision="split")
        dec3.estimator.load_state_dict(sd)
        dec3 = dec3.to(dev)
        sec3 = time_variant(dec3, g, N_STEPS, kw, max(3, args.steps // 2))
        extras["attention_precision_split"] = {"workload": f"rank 0's first batch, {B_batch} x {T_batch}", "ms_per_step": sec3 * 1e3,
                                               "value": batches[0]["valid"] / sec3, "unit": "mel-frames/sec",
                                               "max_attention_log_sum_exp": dec.estimator.engine().attention_stats(torch.cuda.current_stream(dev).cuda_stream)}
        del dec3
        # (b) BASELINE config 1 shape on the GPU: one utterance, T=500, n=10 euler, CFG off (interactive latency)
        one = {k: v.to(dev) for k, v in make_inputs(1, 500, seed=0).items() if k != "lengths"}
        sec1 = time_variant(dec, one, 10, None, 10)
        extras["config1_latency"] = {"workload": "B=1 x T=500, n_timesteps=10 euler, CFG off", "ms_per_solve": sec1 * 1e3,
                                     "mel_frames_per_sec": 500 / sec1, "dtype": args.dtype}
        # (c) the step after the path: the batch's mel through the native Vocos vocoder (seeded weights of the
        #     oracle's generator; 44.1 kHz, hop 512) -> seconds of audio per second for decoder + vocoder
        import types
        from oracle import vocos_oracle as vo
        from stabletts_amd.vocos import Vocos
        vc = vo.VocosConfig
        voc = Vocos(types.SimpleNamespace(input_channels=vc.input_channels, dim=vc.dim, intermediate_dim=vc.intermediate_dim,
                                          num_layers=vc.num_layers), types.SimpleNamespace(n_fft=vc.n_fft, hop_length=vc.hop_length))
        voc.load_state_dict({k: torch.from_numpy(v) for k, v in vo.make_vocos_state_dict(77).items()})
        voc = voc.to(dev)
        for _ in range(2):
            voc(out)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(10):
            audio = voc(out)
        torch.cuda.synchronize(dev)
        voc_s = (time.perf_counter() - t2) / 10
        assert torch.isfinite(audio).all()
        audio_seconds = batches[-1]["valid"] * 512 / 44100.0
        dec_s = time_variant(dec, batches[-1]["g"], N_STEPS, kw, 3)
        extras["vocoder"] = {"workload": f"Vocos (8 ConvNeXt blocks dim 512 + ISTFT head), the last batch's {out.shape[0]} x {out.shape[2]} mel frames, f16 operands",
                             "ms_per_batch": voc_s * 1e3, "mel_frames_per_sec": out.shape[0] * out.shape[2] / voc_s,
                             "decoder_plus_vocoder_audio_seconds_per_second": audio_seconds / (dec_s + voc_s),
                             "real_time_factor": (dec_s + voc_s) / audio_seconds}
        del voc
        # (d) BASELINE config 5 on this GPU: one training step (forward with activations + backward + AdamW), native kernels
        if not args.no_train_leg:
            dec.estimator.release_engine()      # as in a train.py process: no inference engine (arena, part streams) alive next to the training one (-0.15 ms per step)
            extras["train_step"] = train_step_leg(dev, sd, args.train_batch, T_FRAMES, args.dtype, 5)

    def emit(more):
        """Rank 0 assembles and prints THE line (called once; `more` = legs measured after the headline)."""
        if rank != 0:
            return
        p = prof[dom]
        avg_s = p["total_ms"] / max(p["launches"], 1) * 1e-3
        achieved = p["flops_per_launch"] / avg_s / 1e12
        n_evals = 2 * N_STEPS
        if args.ragged:         # padded-frame algorithmic FLOPs of every bucket of every rank (SURVEY 8d at each bucket's own B, T)
            job_flops = sum(algorithmic_flops_per_frame(t, n_evals, n) * n * t for ns, ts in zip(shard_info["batches_per_rank"], shard_info["batch_max_len_per_rank"])
                            for n, t in zip(ns, ts))
        else:
            job_flops = algorithmic_flops_per_frame(T_FRAMES, n_evals, B_PER_GPU) * B_PER_GPU * T_FRAMES * world
        line = {
            "metric": f"mel-frames/sec (whole node), 31M DiT, n_timesteps={N_STEPS}+CFG",
            "value": value, "unit": "mel-frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.ragged else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE config 4: 31M CFM decoder, the same {shard_info['utterances']} ragged utterances len~U{{600..1000}} at every N "
                                    f"(strong scaling), equal-cost length buckets (mean 32 utterances, variable count) dealt to the ranks, a rank's buckets "
                                    f"solved back to back in one step, n_timesteps={N_STEPS} euler, cfg=3.0; value counts valid frames"
                                    if args.ragged else
                                    f"BASELINE config {2 if N_STEPS == 10 else 3}: 31M CFM decoder (hidden 256, filter 1024, 4 heads, 6 DiT blocks, "
                                    f"n_mels 128), batch 32 x T=1000 synthetic mu/mask per GPU, n_timesteps={N_STEPS} euler, "
                                    "cfg=3.0, seeded random weights (adaLN re-randomised)") +
                                   (f"; {args.dtype} MFMA operands" + (" -- same width as BASELINE's bf16, the type that meets north_star's 1e-3 "
                                                                        "(every gate of tests/test_gpu_parity.py)" if args.dtype == "f16" else
                                                                        " (BASELINE's named dtype; ~4e-3 on the displacement metric)")),
                       "global_batch": shard_info["utterances"], "seq_len": T_FRAMES,
                       "parallelism": f"utterance-sharded x{world}, no data-path collective"},
            "sharding": {**shard_info, "rank0_first_batch_B_x_T": [B_batch, T_batch]},
            "parity": "f16 operands (default, this line unless --dtype bf16) meet north_star's 1e-3 on the displacement metric and per "
                      "evaluation (tests/test_gpu_parity.py, gates 7e-4; this workload 3.0e-4 against the fp32 oracle, tools/parity_c2.py; with "
                      "trained-like O(1) adaLN gates at this size 7.3e-4 per evaluation / 4.1e-4 displacement, tools/parity_trained.py); "
                      "bf16 operands measure ~4e-3 (other_dtype)",
            "roofline": {"bound": "mfma", "kernel": ((class_kernel(dom, args.dtype) or "conv_gemm2_kernel") + f" [{dom}]"), "achieved": achieved,
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "traffic": pmc_traffic(dom, args.dtype), "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": pmc_source(),
                         "launches_sampled": p["launches"], "sample_stride": PROFILE_STRIDE, "avg_launch_us": avg_s * 1e6,
                         "flops_per_launch": p["flops_per_launch"],
                         **({"flops_note": "algorithmic = the direct convolutions' multiply-adds (SURVEY 8d); ffn_wino_kernel executes 2/3 of them as MFMAs "
                                           "(Winograd F(2,3) along the frame axis, DESIGN.md section 4): its matrix-pipe utilisation is 2/3 of frac"}
                            if "ffn_wino" in class_kernel(dom, args.dtype) else {}),
                         "sampled_over": f"{args.steps} single-sequence solves (ST_SPLIT=1, {single_seq_ms:.2f} ms each) run right after "
                                         "the timed region, whose concurrent part sequences (two streams by default) would fold the other parts' kernels "
                                         "into a launch's event-bracketed duration"},
            "whole_solve_tflops": job_flops / (elapsed / args.steps) / 1e12,
            "solve_parts": int(os.environ.get("ST_SPLIT", "-1")),
            "solve_parts_note": "-1 = library default: batches >= 24000 (CFG-doubled) frames run as two part-batch launch sequences on two streams (four only with the generic q/k/v tile, ST_QKV_WS=0)",
            "whole_solve_hbm": (lambda b: None if b is None else {
                "bytes_per_solve_pmc": b, "achieved": b / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS})(pmc_solve_bytes() if N_STEPS == 10 and not args.ragged else None),
            "kernel_classes_ms_per_step": {k: v["total_ms"] for k, v in survey.items() if v["launches"]},
            "kernel_classes_note": f"untimed single-sequence survey solve (ST_SPLIT=1, {survey_ms:.2f} ms with its ~360 event pairs at "
                                   "~10 us each) with every launch of these six classes bracketed by HIP events on the launch stream; "
                                   "the dominant class of the roofline object is the largest entry",
        }
        line.update(extras); line.update(more)
        if dev_env:
            line["dev_env"] = dev_env      # NOT the library as shipped
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, (fs, fc), args.cpu_all_cores)
        print(json.dumps(line), flush=True)

    if world > 1 and not args.no_extras and not args.ragged and not args.no_train_leg:
        # Config 5 over RCCL is the one leg with a collective on the data path; it runs LAST and under a watchdog, so that a
        # communicator that cannot be set up on this node (or hangs) costs the line its "train_ddp" object, not the line itself.
        import threading

        def fire():
            emit({"train_ddp": {"error": f"no result within {DDP_WATCHDOG_S} s (RCCL communicator set-up or the first all-reduce hung); the other objects of this line are unaffected"}})
            os._exit(0)
        dog = threading.Timer(DDP_WATCHDOG_S, fire)
        dog.daemon = True
        dog.start()
        try:
            leg = train_ddp_leg(dev, sd, world, rank, dist, share_gpu, args.train_batch, T_FRAMES, args.dtype, 5)
        except Exception as ex:      # noqa: BLE001  (whatever the backend raises: the headline must still be printed)
            leg = {"error": f"{type(ex).__name__}: {ex}"[:400]}
        dog.cancel()
        emit({"train_ddp": leg})
    else:
        emit({})
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
