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

  --ragged   BASELINE config 4: 32*N utterances with len ~ U{600..1000}, length-sorted and dealt to the ranks by
             stabletts_amd.sharding.assign_batches (32 per GPU); value counts VALID frames only, the line carries
             the sharder's imbalance / padding figures.
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


CPU_THREADS = 32               # fixed (min with the host's cores): the figure must not move with a calibration sweep


def cpu_baseline(sd, cfg_params):
    """SURVEY.md section 8(d)'s protocol for config-2-sized inputs: time ONE evaluation of the workload's own batch -- B=32 x T=1000,
    the cond and the uncond estimator call of one cfg_wrapper step (flow_matching.py:58-67), prenet recomputed in each as the
    reference does -- and scale by the step count (every Euler step costs the same two evaluations).  The oracle (fp32 torch-CPU
    restatement of the reference) on a FIXED thread count; ~10-30 s of CPU work on the MI355X host."""
    import oracle
    from oracle.inputs import make_inputs
    fs, fc = cfg_params
    threads = max(1, min(CPU_THREADS, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    inp = make_inputs(B_PER_GPU, T_FRAMES, seed=0)
    small = make_inputs(2, 256, seed=0)
    t_span = oracle.linspace_f32(N_STEPS)
    with torch.inference_mode():
        oracle.cfg_wrapper(sd, t_span[0], small["z"], small["mask"], small["mu"], small["c"], fs, fc, CFG)      # warm-up (thread pool, oneDNN primitives)
        t0 = time.perf_counter()
        oracle.cfg_wrapper(sd, t_span[0], inp["z"], inp["mask"], inp["mu"], inp["c"], fs, fc, CFG)
        t_eval = time.perf_counter() - t0
    per_solve = t_eval * N_STEPS
    return dict(value=B_PER_GPU * T_FRAMES / per_solve, unit="mel-frames/sec", cores=threads, kind="port",
                sample=f"oracle (fp32 torch-CPU restatement of the reference; prenet recomputed every evaluation as the reference does), "
                       f"the workload's own batch B={B_PER_GPU} x T={T_FRAMES}, cfg={CFG}: ONE cfg_wrapper step (cond + uncond evaluation) timed "
                       f"({t_eval:.2f} s) and scaled by the {N_STEPS} euler steps (SURVEY 8d); {threads} torch threads (fixed), "
                       f"os.cpu_count()={os.cpu_count()}")


def train_step_leg(dev, sd, B=64, T=1000, dtype="f16", steps=5, dropout=True):
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
    opt = torch.optim.AdamW(dec.parameters(), lr=1e-4)
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
           "torch_GB": torch.cuda.max_memory_allocated(dev) / 1e9, "engine_GB": dec.estimator.engine().device_bytes() / 1e9}
    del dec, opt
    return res


# kernel that implements each profiled class on the default path (for the PMC traffic lookup)
CLASS_KERNEL = {
    "ffn_conv1": "conv_gemm_phased3_kernel<st::Op{DT}, 0, false>",
    "ffn_conv2": "ffn_fused_kernel<st::Op{DT}, 0, 0>",        # the whole FFN since round 4 (conv_1 + SiLU + conv_2 in one launch); f16: CLASS_KERNEL_F16
    "lsc_conv": "conv_gemm_phased3_kernel<st::Op{DT}, 1, true>",
    "attention": "attention_kernel<st::Op{DT}, false>",
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
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic_before_wino.json", "r03_pmc_traffic.json", "r02_pmc_traffic_v2.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name))), name
        except Exception:
            continue
    return None, None


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
    ap.add_argument("--no-extras", action="store_true", help="skip the other-dtype and config-1 latency legs")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the config-5 training-step leg of the extras")
    ap.add_argument("--ragged", action="store_true", help="BASELINE config 4: ragged utterances through the sharder")
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
    dev_env = {k: v for k, v in os.environ.items() if (k.startswith("ST_") and k not in ("ST_SPLIT", "ST_HIP_GRAPH")) or k == "STABLETTS_HIP_LIB"}
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

    # utterance sharding: world*32 utterances, dealt as length-sorted batches (no collective)
    if args.ragged:
        import numpy as np
        lengths = np.random.default_rng(4).integers(600, T_FRAMES + 1, size=B_PER_GPU * world).tolist()
    else:
        lengths = [T_FRAMES] * (B_PER_GPU * world)
    per_rank = sharding.assign_batches(lengths, B_PER_GPU, world)
    my_batches = per_rank[rank]
    assert len(my_batches) == 1 and len(my_batches[0]) == B_PER_GPU
    my_lengths = [lengths[i] for i in my_batches[0]]
    T_batch = max(my_lengths)
    inp = make_inputs(B_PER_GPU, T_batch, seed=rank, lengths=my_lengths)
    g = {k: v.to(dev) for k, v in inp.items() if k != "lengths"}
    kw = dict(fake_speaker=fs.to(dev), fake_content=fc.to(dev), cfg_strength=CFG)
    valid_frames_total = sum(lengths)
    shard_imbalance, shard_padding = sharding.imbalance(lengths, per_rank)

    def step():
        return dec(g["mu"], g["mask"], N_STEPS, 1.0, g["c"], "euler", kw, z=g["z"])

    eng = dec.estimator.engine()
    for _ in range(args.warmup):
        step()
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

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
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
    if rank == 0 and world == 1 and not args.no_extras:
        # (a) the other MFMA operand type on the same workload (f16 is the parity-gated configuration: it meets
        #     north_star's 1e-3 on the displacement metric; bf16 is BASELINE's named dtype and measures ~4e-3)
        other = "f16" if args.dtype == "bf16" else "bf16"
        dec2 = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=other)
        dec2.estimator.load_state_dict(sd)
        dec2 = dec2.to(dev)
        sec = time_variant(dec2, g, N_STEPS, kw, max(3, args.steps // 2))
        extras["other_dtype"] = {"dtype": other, "ms_per_step": sec * 1e3, "value": valid_frames_total / sec,
                                 "unit": "mel-frames/sec"}
        del dec2
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
        audio_seconds = valid_frames_total * 512 / 44100.0
        extras["vocoder"] = {"workload": f"Vocos (8 ConvNeXt blocks dim 512 + ISTFT head), the batch's {B_PER_GPU} x {T_batch} mel frames, f16 operands",
                             "ms_per_batch": voc_s * 1e3, "mel_frames_per_sec": B_PER_GPU * T_batch / voc_s,
                             "decoder_plus_vocoder_audio_seconds_per_second": audio_seconds / (elapsed / args.steps + voc_s),
                             "real_time_factor": (elapsed / args.steps + voc_s) / audio_seconds}
        del voc
        # (d) BASELINE config 5 on this GPU: one training step (forward with activations + backward + AdamW), native kernels
        if not args.no_train_leg:
            extras["train_step"] = train_step_leg(dev, sd, 64, T_FRAMES, args.dtype, 5)

    if rank == 0:
        p = prof[dom]
        avg_s = p["total_ms"] / max(p["launches"], 1) * 1e-3
        achieved = p["flops_per_launch"] / avg_s / 1e12
        n_evals = 2 * N_STEPS
        falg = algorithmic_flops_per_frame(T_batch, n_evals, B_PER_GPU)
        line = {
            "metric": f"mel-frames/sec (whole node), 31M DiT, n_timesteps={N_STEPS}+CFG",
            "value": value, "unit": "mel-frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE config 4: 31M CFM decoder, {world * B_PER_GPU} ragged utterances len~U{{600..1000}} "
                                    f"sharded 32 per GPU (length-sorted batches), n_timesteps={N_STEPS} euler, cfg=3.0; value counts valid frames"
                                    if args.ragged else
                                    f"BASELINE config {2 if N_STEPS == 10 else 3}: 31M CFM decoder (hidden 256, filter 1024, 4 heads, 6 DiT blocks, "
                                    f"n_mels 128), batch 32 x T=1000 synthetic mu/mask per GPU, n_timesteps={N_STEPS} euler, "
                                    "cfg=3.0, seeded random weights (adaLN re-randomised)") +
                                   (f"; {args.dtype} MFMA operands" + (" -- same width as BASELINE's bf16, the type that meets north_star's 1e-3 "
                                                                        "(every gate of tests/test_gpu_parity.py)" if args.dtype == "f16" else
                                                                        " (BASELINE's named dtype; ~4e-3 on the displacement metric)")),
                       "global_batch": world * B_PER_GPU, "seq_len": T_FRAMES,
                       "parallelism": f"utterance-sharded x{world}, no data-path collective"},
            "sharding": {"imbalance_max_over_mean": shard_imbalance, "padded_over_valid_frames": shard_padding,
                         "valid_frames": valid_frames_total, "padded_T_this_rank": T_batch},
            "parity": "f16 operands (default, this line unless --dtype bf16) meet north_star's 1e-3 on the displacement metric and per "
                      "evaluation (tests/test_gpu_parity.py, gates 7e-4; this workload 3.0e-4 against the fp32 oracle, tools/parity_c2.py; with "
                      "trained-like O(1) adaLN gates at this size 7.3e-4 per evaluation / 4.1e-4 displacement, tools/parity_trained.py); "
                      "bf16 operands measure ~4e-3 (other_dtype)",
            "roofline": {"bound": "mfma", "kernel": ((class_kernel(dom, args.dtype) or "conv_gemm2_kernel") + f" [{dom}]"), "achieved": achieved,
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "traffic": pmc_traffic(dom, args.dtype), "traffic_unit": "HBM bytes per launch (PMC)",
                         "launches_sampled": p["launches"], "sample_stride": PROFILE_STRIDE, "avg_launch_us": avg_s * 1e6,
                         "flops_per_launch": p["flops_per_launch"],
                         **({"flops_note": "algorithmic = the direct convolutions' multiply-adds (SURVEY 8d); ffn_wino_kernel executes 2/3 of them as MFMAs "
                                           "(Winograd F(2,3) along the frame axis, DESIGN.md section 4): its matrix-pipe utilisation is 2/3 of frac"}
                            if "ffn_wino" in class_kernel(dom, args.dtype) else {}),
                         "sampled_over": f"{args.steps} single-sequence solves (ST_SPLIT=1, {single_seq_ms:.2f} ms each) run right after "
                                         "the timed region, whose concurrent part sequences (two streams by default) would fold the other parts' kernels "
                                         "into a launch's event-bracketed duration"},
            "whole_solve_tflops": falg * B_PER_GPU * T_batch / (elapsed / args.steps) / 1e12 * world,
            "solve_parts": int(os.environ.get("ST_SPLIT", "-1")),
            "solve_parts_note": "-1 = library default: batches >= 24000 (CFG-doubled) frames run as two part-batch launch sequences on two streams (four only with the generic q/k/v tile, ST_QKV_WS=0)",
            "whole_solve_hbm": (lambda b: None if b is None else {
                "bytes_per_solve_pmc": b, "achieved": b / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS})(pmc_solve_bytes() if N_STEPS == 10 else None),
            "kernel_classes_ms_per_step": {k: v["total_ms"] for k, v in survey.items() if v["launches"]},
            "kernel_classes_note": f"untimed single-sequence survey solve (ST_SPLIT=1, {survey_ms:.2f} ms with its ~360 event pairs at "
                                   "~10 us each) with every launch of these six classes bracketed by HIP events on the launch stream; "
                                   "the dominant class of the roofline object is the largest entry",
        }
        line.update(extras)
        if dev_env:
            line["dev_env"] = dev_env      # NOT the library as shipped
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, (fs, fc))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
