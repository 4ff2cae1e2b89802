"""Engine behaviour on a real MI355X that is not numerics-vs-oracle: solve-part splitting, HIP-graph replay across
table reallocations, weight synchronisation, device checks, and the multi-process utterance-sharded path
(2 ranks sharing the one visible GPU).  Run with ``-m gpu``."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

import oracle
from oracle.inputs import make_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dec(sd):
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="bf16")
    d.estimator.load_state_dict(sd)
    return d.to("cuda:0")


def _kw(cfg_params, s):
    fs, fc = cfg_params
    return dict(fake_speaker=fs.cuda(), fake_content=fc.cuda(), cfg_strength=s)


def _solve(d, inp, n, solver, kw):
    return d(inp["mu"].cuda(), inp["mask"].cuda(), n, 1.0, inp["c"].cuda(), solver, kw, z=inp["z"].cuda()).cpu()


@pytest.mark.parametrize("solver,n,cfg", [("euler", 4, 3.0), ("rk4", 2, None), ("midpoint", 3, 2.0)])
def test_two_part_solve_is_bitwise_identical(sd, cfg_params, monkeypatch, solver, n, cfg):
    """ST_SPLIT=2: the batch is solved as two parts on two streams (MFMA-bound kernels of one part overlap the
    HBM-bound epilogues of the other).  Utterances never share a tile, so every output must equal the one-part
    solve bit for bit -- odd batch sizes, ragged lengths, with and without CFG, eager and graph replay.
    (Engine created with ST_SMALL_GRID=0: the split-K path of small grids picks its split count -- hence its
    summation order -- from the launch's block count, which differs between a batch and its parts.)"""
    from stabletts_amd.flow_matching import CFMDecoder
    monkeypatch.setenv("ST_SMALL_GRID", "0")
    dec = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="bf16")
    dec.estimator.load_state_dict(sd)
    dec = dec.to("cuda:0")
    kw = _kw(cfg_params, cfg) if cfg is not None else None
    for B, T, lengths in ((5, 200, [200, 133, 64, 200, 7]), (2, 70, [70, 51])):
        inp = make_inputs(B, T, seed=80 + B, lengths=lengths)
        monkeypatch.setenv("ST_SPLIT", "1")
        one = _solve(dec, inp, n, solver, kw)
        monkeypatch.setenv("ST_SPLIT", "2")
        two = _solve(dec, inp, n, solver, kw)
        assert torch.equal(one, two)
        monkeypatch.setenv("ST_SPLIT", "4")             # up to four parts (B >= 4; B = 2 stays at two)
        assert torch.equal(_solve(dec, inp, n, solver, kw), one)
        monkeypatch.setenv("ST_SPLIT", "2")
        monkeypatch.setenv("ST_HIP_GRAPH", "1")
        for _ in range(3):                      # eager, capture, replay
            assert torch.equal(_solve(dec, inp, n, solver, kw), one)
        monkeypatch.delenv("ST_HIP_GRAPH")


def _fresh(sd, monkeypatch, dtype="bf16", **env):
    """A decoder whose engine is created under the given environment (the tile-policy knobs are read at st_create)."""
    from stabletts_amd.flow_matching import CFMDecoder
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype=dtype)
    d.estimator.load_state_dict(sd)
    d = d.to("cuda:0")
    d.estimator.engine()
    for k in env:
        monkeypatch.delenv(k)
    return d


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_phased_k_loop_is_bit_identical_to_the_two_buffer_kernel(sd, cfg_params, monkeypatch, dtype):
    """conv_gemm_phased3_kernel (K-slice phases, staggered wave groups, three weight buffers, counted LDS-DMA waits,
    254-frame tiles) accumulates every output in the same order as conv_gemm2_kernel's 256 x 256 tile: whole solves
    must agree bit for bit -- at tile-edge lengths (254 | 255 | 509 frames: one tile, one frame into the second, two
    tiles minus 1+...), ragged masks and with the long-skip (two-source) convolutions.  ST_BIG_MIN_BLOCKS=1 forces the
    256-wide tiles at these small batch sizes; a race in the hand-counted vmcnt / barrier protocol would show up as
    run-to-run differences, so every solve is repeated."""
    kw = _kw(cfg_params, 3.0)
    old = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1", ST_PHASED="0", ST_SMALL_GRID="0")
    new = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1", ST_PHASED="1", ST_SMALL_GRID="0")
    for B, T, lengths in ((2, 254, [254, 100]), (1, 255, [255]), (3, 509, [509, 508, 3]), (2, 700, [700, 255])):
        inp = make_inputs(B, T, seed=90 + T, lengths=lengths)
        ref = _solve(old, inp, 2, "euler", kw)
        for _ in range(3):
            assert torch.equal(_solve(new, inp, 2, "euler", kw), ref), (B, T)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_fused_ffn_is_bit_identical_to_the_two_kernel_path(sd, cfg_params, monkeypatch, dtype):
    """ffn_fused_kernel (conv_1 -> SiLU / mask / 16-bit rounding in LDS -> conv_2 -> RESGATE(+LN) epilogue in ONE launch,
    126-frame tiles, 256-channel chunks of the intermediate, operand areas shared by h and u, a 5-slab weight ring with
    counted LDS-DMA waits) contracts in the two-kernel path's K order with the same SiLU / rounding instructions: whole
    solves must agree bit for bit -- at tile-edge lengths (126 | 127 | 252 | 253 frames), ragged masks with tile skipping,
    a length-1 row, and repeated (a race in the barrier / vmcnt protocol would show up as run-to-run differences)."""
    kw = _kw(cfg_params, 3.0)
    old = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="0", ST_SMALL_GRID="0")
    new = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="1", ST_SMALL_GRID="0")
    for B, T, lengths in ((2, 126, [126, 100]), (1, 127, [127]), (3, 253, [253, 252, 1]), (2, 700, [700, 255]), (4, 1000, [1000, 873, 640, 377])):
        inp = make_inputs(B, T, seed=60 + T, lengths=lengths)
        ref = _solve(old, inp, 2, "euler", kw)
        for _ in range(3):
            out = _solve(new, inp, 2, "euler", kw)
            assert torch.equal(out, ref), (B, T, float((out - ref).abs().max()))
    # the headline shape: four solve parts in flight on four streams.  (Round 4: with the next block's LayerNorm output written into
    # the buffer the FFN input is read from -- fine for two kernels -- blocks of the fused kernel overwrote halo rows their
    # neighbours had yet to read; only visible with several launch sequences in flight, as 4e-5 run-to-run differences.)
    inp = make_inputs(32, 1000, seed=0)
    ref = _solve(old, inp, 2, "euler", kw)
    for _ in range(3):
        assert torch.equal(_solve(new, inp, 2, "euler", kw), ref)
    # one evaluation with a per-item t (the training-shaped entry point) through the same launches
    inp = make_inputs(3, 400, seed=61, lengths=[400, 399, 17])
    t = torch.tensor([0.1, 0.5, 0.9])
    args = (t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    with torch.no_grad():
        assert torch.equal(new.estimator(*args), old.estimator(*args))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_weight_stationary_qkv_is_bit_identical_to_the_generic_tile(sd, cfg_params, monkeypatch, dtype):
    """qkv_ws_kernel (weights of a q / k / v plane in registers, 64-frame activation tiles streamed through an LDS ring by a
    persistent block, counted waits over LDS-DMA pieces AND row stores) contracts K in the generic tile's order with the same
    RoPE / scaling expressions: the captured q, k, v^T planes and whole solves must agree with conv_gemm2_kernel<EPI_QKV> bit
    for bit -- T not a multiple of 64, one-tile and many-tile items, ragged batches with tile skipping (work lists with holes),
    a length-1 row, more items than a block's list stride, repeated (a miscounted vmcnt shows as run-to-run differences)."""
    kw = _kw(cfg_params, 3.0)
    old = _fresh(sd, monkeypatch, dtype, ST_QKV_WS="0", ST_SMALL_GRID="0")
    new = _fresh(sd, monkeypatch, dtype, ST_QKV_WS="1", ST_QKV_WS_MIN_TILES="1", ST_SMALL_GRID="0")
    cases = [(2, 64, [64, 33]), (1, 65, [65]), (3, 253, [253, 252, 1]), (2, 700, [700, 255]), (4, 1000, [1000, 873, 640, 377]),
             (9, 130, [130, 7, 129, 64, 65, 1, 128, 100, 130]), (32, 1000, None),
             (65, 2000, None)]      # 130 items > 64 x the list stride the grid rule would pick: stride from the 64-bit mask, grid > 256 blocks
    for B, T, lengths in cases:
        inp = make_inputs(B, T, seed=40 + T, lengths=lengths) if lengths else make_inputs(B, T, seed=0, ragged=True)
        ref = _solve(old, inp, 2, "euler", kw)
        for _ in range(3):
            out = _solve(new, inp, 2, "euler", kw)
            assert torch.equal(out, ref), (B, T, float((out - ref).abs().max()))
    # the planes themselves, through the debug capture (every frame computed: no tile skipping under capture)
    inp = make_inputs(3, 200, seed=41, lengths=[200, 131, 64])
    t = torch.tensor(0.4)
    args = (t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    got = {}
    for name, d in (("old", old), ("new", new)):
        eng = d.estimator.engine()
        eng.debug_capture(True)
        try:
            with torch.no_grad():
                d.estimator(*args)
            torch.cuda.synchronize()
            got[name] = {k: eng.debug_fetch(f"b{blk}.{k}") for blk in (0, 5) for k in ("q", "k", "vt")}
        finally:
            eng.debug_capture(False)
    for k in got["old"]:
        assert torch.equal(torch.as_tensor(got["old"][k]), torch.as_tensor(got["new"][k])), k


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_weight_stationary_out_projection_is_bit_identical_to_the_generic_tile(sd, cfg_params, monkeypatch, dtype):
    """oproj_ws_kernel (out projection + gate + residual + LayerNorm_2 + modulate as a persistent kernel: weight in registers,
    attention output / residual rows / per-item constants / frame mask all by LDS-DMA one 32-frame tile ahead, the residual stream
    updated in place) against conv_gemm2_kernel<EPI_RESGATE>: whole solves bit for bit -- T not a multiple of 32, ragged work
    lists with holes, a length-1 row, masks, more items than a block's list stride, repeated."""
    kw = _kw(cfg_params, 3.0)
    old = _fresh(sd, monkeypatch, dtype, ST_OPROJ_WS="0", ST_SMALL_GRID="0")
    new = _fresh(sd, monkeypatch, dtype, ST_OPROJ_WS="1", ST_OPROJ_WS_MIN_TILES="1", ST_SMALL_GRID="0")
    cases = [(2, 64, [64, 33]), (1, 65, [65]), (3, 253, [253, 252, 1]), (2, 700, [700, 255]), (4, 1000, [1000, 873, 640, 377]),
             (9, 130, [130, 7, 129, 64, 65, 1, 128, 100, 130]), (32, 1000, None),
             (65, 2000, None)]      # 130 items > 64 x the list stride the grid rule would pick: stride from the 64-bit mask, grid > 256 blocks
    for B, T, lengths in cases:
        inp = make_inputs(B, T, seed=40 + T, lengths=lengths) if lengths else make_inputs(B, T, seed=0, ragged=True)
        ref = _solve(old, inp, 2, "euler", kw)
        for _ in range(3):
            out = _solve(new, inp, 2, "euler", kw)
            assert torch.equal(out, ref), (B, T, float((out - ref).abs().max()))
    inp = make_inputs(3, 400, seed=61, lengths=[400, 399, 17])
    t = torch.tensor([0.1, 0.5, 0.9])
    args = (t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    with torch.no_grad():
        assert torch.equal(new.estimator(*args), old.estimator(*args))


def test_winograd_fused_ffn_matches_the_direct_kernel_within_its_rounding(sd, cfg_params, monkeypatch):
    """ffn_wino_kernel (the f16 default on big grids: F(2,3) along the frame axis -- raw pair-interleaved rows, transformed operands
    formed with packed f16 adds, the fourth weight plane derived in registers, wave-private weight rings with per-position wait
    counts) against the direct fused kernel (ST_FUSED_FFN=1, itself bit-identical to the two-kernel path): NOT bit-identical by
    construction -- its operands are rounded sums -- but within a few 1e-4 of the solve's displacement (the oracle-level cost is
    +0.6e-4, tools/parity_c2.py), at tile-edge lengths, ragged with tile skipping, a length-1 row, the headline shape, and
    bit-REPRODUCIBLE run to run (a miscounted vmcnt or a missing barrier shows as run-to-run differences).  bf16 engines keep
    the direct kernel whatever the variable says."""
    kw = _kw(cfg_params, 3.0)
    old = _fresh(sd, monkeypatch, "f16", ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="1", ST_SMALL_GRID="0")
    new = _fresh(sd, monkeypatch, "f16", ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="3", ST_SMALL_GRID="0")
    worst = 0.0
    for B, T, lengths in ((2, 126, [126, 100]), (1, 127, [127]), (3, 253, [253, 252, 1]), (2, 700, [700, 255]), (4, 1000, [1000, 873, 640, 377]),
                          (32, 1000, None)):
        inp = make_inputs(B, T, seed=60 + T, lengths=lengths) if lengths else make_inputs(B, T, seed=0, ragged=True)
        ref = _solve(old, inp, 4, "euler", kw)
        out = _solve(new, inp, 4, "euler", kw)
        for _ in range(2):
            assert torch.equal(_solve(new, inp, 4, "euler", kw), out), (B, T)
        assert torch.isfinite(out).all()
        pad = ~inp["mask"].bool().expand_as(out)
        assert torch.equal(out[pad], inp["z"][pad])                     # padded frames untouched, exactly
        d = float((out - ref).abs().max() / (ref - inp["z"]).abs().max())
        worst = max(worst, d)
        assert 0.0 < d < 5e-4, (B, T, d)                                # different arithmetic (not 0), same function
    print(f"Winograd vs direct fused FFN, solve displacement: worst {worst:.2e}")
    # one evaluation with a per-item t
    inp = make_inputs(3, 400, seed=61, lengths=[400, 399, 17])
    t = torch.tensor([0.1, 0.5, 0.9])
    args = (t.cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    with torch.no_grad():
        a, b = new.estimator(*args), old.estimator(*args)
    assert float((a - b).abs().max() / b.abs().max()) < 5e-4
    # bf16: the variable is ignored (packed f16 adds form the operands)
    b1 = _fresh(sd, monkeypatch, "bf16", ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="1", ST_SMALL_GRID="0")
    b3 = _fresh(sd, monkeypatch, "bf16", ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN="3", ST_SMALL_GRID="0")
    inp = make_inputs(2, 300, seed=5, lengths=[300, 211])
    assert torch.equal(_solve(b3, inp, 2, "euler", kw), _solve(b1, inp, 2, "euler", kw))


@pytest.mark.parametrize("F", [256, 512, 768, 2048])
def test_winograd_fused_ffn_other_filter_widths(cfg_params, monkeypatch, F):
    """1, 2, 3 and 8 chunks of 256 intermediate channels instead of the 31M model's 4 (stream length, bias area, `last chunk` waits;
    one chunk: the first chunk is also the last)."""
    from stabletts_amd.flow_matching import CFMDecoder
    cfg = oracle.DecoderConfig(filter_channels=F)
    sdf = oracle.make_state_dict(777, cfg)
    kw = _kw(cfg_params, 2.0)
    decs = []
    for fused in ("1", "3"):
        for k, v in dict(ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN=fused, ST_SMALL_GRID="0").items():
            monkeypatch.setenv(k, v)
        d = CFMDecoder(128, 128, 256, 128, F, 4, 6, 3, 0.1, 256)
        d.estimator.load_state_dict(sdf)
        d = d.to("cuda:0"); d.estimator.engine()
        decs.append(d)
    for k in ("ST_BIG_MIN_BLOCKS", "ST_FUSED_FFN", "ST_SMALL_GRID"):
        monkeypatch.delenv(k)
    inp = make_inputs(3, 380, seed=33, lengths=[380, 251, 127])
    ref = _solve(decs[0], inp, 2, "euler", kw)
    out = _solve(decs[1], inp, 2, "euler", kw)
    assert torch.equal(_solve(decs[1], inp, 2, "euler", kw), out)
    d = float((out - ref).abs().max() / (ref - inp["z"]).abs().max())
    assert 0.0 < d < 5e-4, d


@pytest.mark.parametrize("F", [512, 768, 2048])
def test_fused_ffn_other_filter_widths(cfg_params, monkeypatch, F):
    """The fused kernel walks the intermediate in 256-channel chunks (2, 3, 8 of them here instead of the 31M model's 4); its
    weight stream, bias area and wait counts depend on the chunk count only through `last chunk`.  Bit-identical to the
    two-kernel path, ragged, repeated."""
    from stabletts_amd.flow_matching import CFMDecoder
    cfg = oracle.DecoderConfig(filter_channels=F)
    sdf = oracle.make_state_dict(777, cfg)
    kw = _kw(cfg_params, 2.0)
    decs = []
    for fused in ("0", "1"):
        for k, v in dict(ST_BIG_MIN_BLOCKS="1", ST_FUSED_FFN=fused, ST_SMALL_GRID="0").items():
            monkeypatch.setenv(k, v)
        d = CFMDecoder(128, 128, 256, 128, F, 4, 6, 3, 0.1, 256)
        d.estimator.load_state_dict(sdf)
        d = d.to("cuda:0"); d.estimator.engine()
        decs.append(d)
    for k in ("ST_BIG_MIN_BLOCKS", "ST_FUSED_FFN", "ST_SMALL_GRID"):
        monkeypatch.delenv(k)
    inp = make_inputs(3, 380, seed=33, lengths=[380, 251, 127])
    ref = _solve(decs[0], inp, 2, "euler", kw)
    for _ in range(2):
        assert torch.equal(_solve(decs[1], inp, 2, "euler", kw), ref)


def test_small_grid_variants_match_the_plain_kernels(sd, cfg_params, monkeypatch):
    """Split-K convolutions (+ row-wise finish kernel), 64-frame tiles and the key-split attention kernel change only
    the fp32 summation order: a small solve with them (default) and without (ST_SMALL_GRID=0) must agree to fp32
    rounding amplified through the 16-bit operands -- far inside the parity gate -- and repeat bit for bit."""
    kw = _kw(cfg_params, 3.0)
    plain = _fresh(sd, monkeypatch, "f16", ST_SMALL_GRID="0")
    small = _fresh(sd, monkeypatch, "f16")
    for B, T, lengths in ((1, 500, [500]), (2, 130, [130, 77]), (1, 31, [31])):
        inp = make_inputs(B, T, seed=70 + T, lengths=lengths)
        a, b = _solve(plain, inp, 4, "euler", kw), _solve(small, inp, 4, "euler", kw)
        assert torch.equal(_solve(small, inp, 4, "euler", kw), b)
        rel = float((a - b).abs().max() / a.abs().max())
        print(f"B={B} T={T}: small-grid vs plain {rel:.2e}")
        assert rel < 2e-4


def test_graph_replay_survives_rope_table_growth(dec, cfg_params, monkeypatch):
    """the RoPE tables are reallocated when T grows; instantiated graphs that baked the old pointers
    into the QKV kernel arguments must be dropped.  Sequence: capture a graph at T=200, then a longer-T solve with a
    smaller batch (workspace does not grow, tables do), then replay the first signature."""
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="bf16")
    d.estimator.load_state_dict(dec.estimator.state_dict())
    d = d.cuda()                                 # fresh engine: rope_T starts at 0
    kw = _kw(cfg_params, 2.0)
    a = make_inputs(8, 200, seed=90, lengths=[200, 190, 180, 170, 160, 150, 140, 130])
    b = make_inputs(1, 500, seed=91)
    monkeypatch.delenv("ST_HIP_GRAPH", raising=False)
    ref_a, ref_b = _solve(d, a, 3, "euler", kw), None
    monkeypatch.setenv("ST_HIP_GRAPH", "1")
    for _ in range(3):
        assert torch.equal(_solve(d, a, 3, "euler", kw), ref_a)       # eager, capture, replay (rope_T = 256)
    ref_b = _solve(d, b, 3, "euler", kw)                              # T=500 -> tables reallocated
    for _ in range(3):
        assert torch.equal(_solve(d, a, 3, "euler", kw), ref_a)       # must not replay a graph with dangling pointers
    assert torch.equal(_solve(d, b, 3, "euler", kw), ref_b)


def test_inputs_on_the_wrong_device_raise(dec):
    inp = make_inputs(2, 16, seed=1)
    with pytest.raises(ValueError, match="is on cpu"):
        dec(inp["mu"], inp["mask"].cuda(), 2, 1.0, inp["c"].cuda(), "euler")
    with pytest.raises(ValueError, match="is on cpu"):
        dec.estimator(torch.tensor(0.1).cuda(), inp["z"].cuda(), inp["mask"], inp["mu"].cuda(), inp["c"].cuda())


def test_release_engine_frees_and_rebuilds(sd):
    """release_engine(): the handle is destroyed (device bytes go back to the driver), the next call builds a fresh engine with the same results."""
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    d.estimator.load_state_dict(sd)
    d = d.cuda()
    inp = make_inputs(2, 48, seed=4, lengths=[48, 31])
    args = (torch.tensor(0.7).cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    a = d.estimator(*args)
    old = d.estimator.engine()
    d.estimator.release_engine()
    assert old.handle is None and d.estimator._engine is None
    d.estimator.release_engine()                          # idempotent
    b = d.estimator(*args)
    assert d.estimator.engine() is not old and torch.equal(a, b)


def test_sync_weights_after_data_writes(sd):
    """Writes through ``p.data`` do not bump the version counter (EMA / weight-swap utilities): sync_weights()."""
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="f16")
    d.estimator.load_state_dict(sd)
    d = d.cuda()
    inp = make_inputs(1, 32, seed=2)
    args = (torch.tensor(0.2).cuda(), inp["z"].cuda(), inp["mask"].cuda(), inp["mu"].cuda(), inp["c"].cuda())
    a = d.estimator(*args)
    d.estimator.final_proj.bias.data.add_(1.0)          # invisible to the version counter
    d.estimator.sync_weights()
    b = d.estimator(*args)
    assert float((b - a).mean()) == pytest.approx(1.0, abs=1e-3)
    d.estimator.load_state_dict(sd)                      # load_state_dict is picked up without sync_weights()
    c = d.estimator(*args)
    assert torch.equal(c, a)


def test_two_ranks_sharing_one_gpu_equal_single_process(tmp_path):
    """SURVEY section 4 item 6 / BASELINE config 4 in miniature: 2 processes (torch.distributed.run, gloo for the
    bookkeeping, BENCH_SHARE_GPU=1 so both use the one visible GPU) shard 12 ragged utterances with
    sharding.assign_batches, solve their batches natively, and rank 0 gathers; the result must be bitwise equal to one
    process solving the same batches.  No data-path collective is involved."""
    out = tmp_path / "shard.pt"
    env = dict(os.environ, BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "shard_solve.py"), "--out", str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_solve.py"), "--out", str(tmp_path / "one.pt")],
                        env=dict(os.environ), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    want = torch.load(tmp_path / "one.pt")
    assert got["world"] == 2 and want["world"] == 1
    assert sorted(got["mel"]) == sorted(want["mel"]) == list(range(12))
    for i in range(12):
        assert torch.equal(got["mel"][i], want["mel"][i]), i
    assert got["imbalance"][0] < 1.5


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's plain command form): bench.py
    re-executes itself under torch.distributed.run, one rank per GPU (BENCH_SHARE_GPU=1: both ranks on the one visible
    GPU), and rank 0 prints exactly one JSON line for the whole job -- the config-2 headline (weak scaling) plus the two
    multi-rank legs: "ragged" (config 4: ONE utterance set cut into equal-cost buckets, strong scaling) and "train_ddp"
    (config 5: DistributedDataParallel around compute_loss; gloo here because the ranks share a device, RCCL otherwise).
    Reduced leg sizes so that the test stays short."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--ragged-utterances", "48", "--train-batch", "4"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["dtype"] == "f16"
    assert j["config"]["global_batch"] == 64 and j["value"] > 0 and j["roofline"]["frac"] > 0
    rg = j["ragged"]
    assert rg["scaling"] == "strong" and rg["value"] > 0 and rg["sharding"]["utterances"] == 48
    assert rg["sharding"]["imbalance_max_over_mean"] <= 1.05 and len(rg["sharding"]["batches_per_rank"]) == 2
    assert sum(n for bs in rg["sharding"]["batches_per_rank"] for n in bs) == 48
    td = j["train_ddp"]
    assert td["ranks"] == 2 and td["backend"] == "gloo" and td["ms_per_step"] > 0 and td["allreduce_bytes_per_step"] > 80e6
    assert td["mean_loss_last_step"] == td["mean_loss_last_step"]          # finite


def test_bench_ragged_headline_is_one_workload_in_equal_cost_buckets():
    """`bench.py --ragged` (config 4 as the headline): N = 1 solves the SAME utterance set an 8-GPU run shards -- all of its buckets
    back to back inside the timed region -- so the driver's 1 -> 8 curve is strong scaling of one workload."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ragged", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--ragged-utterances", "96"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 1 and j["config"]["global_batch"] == 96
    assert len(j["sharding"]["batches_per_rank"][0]) == 3 and sum(j["sharding"]["batches_per_rank"][0]) == 96
    assert j["value"] > 0 and j["sharding"]["valid_frames"] < 96 * 1000


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_ragged_tile_skipping_and_qkv_two_block_tiles_are_bitwise_neutral(sd, cfg_params, monkeypatch, dtype):
    """Round-3 launch-level changes must not change a single bit of the result:
      * ST_RAGGED_SKIP (default on): frame tiles past an utterance's last needed frame (last valid + 4, the reach of the
        reference's pad leak) are not computed -- valid frames, and the padded frames (= z), must equal the run that computes
        every tile;
      * ST_QKV_RC1 (default on): the fused q/k/v projection on 256 x 128 tiles with one weight buffer (two blocks per CU)
        accumulates every output element over the same k order as the 256 x 256 tile.
    Ragged batch at T = 1000 with lengths that leave one, two and three whole tiles unused, CFG on, 3 Euler steps."""
    kw = _kw(cfg_params, 3.0)
    lengths = [1000, 997, 760, 759, 508, 505, 300, 254, 251, 1, 640, 900]
    inp = make_inputs(len(lengths), 1000, seed=95, lengths=lengths)
    base = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1")
    ref = _solve(base, inp, 3, "euler", kw)
    pad = ~inp["mask"].bool().expand_as(ref)
    assert torch.equal(ref[pad], inp["z"][pad])
    for env in (dict(ST_RAGGED_SKIP="0"), dict(ST_QKV_RC1="0"), dict(ST_RAGGED_SKIP="0", ST_QKV_RC1="0")):
        other = _fresh(sd, monkeypatch, dtype, ST_BIG_MIN_BLOCKS="1", **env)
        out = _solve(other, inp, 3, "euler", kw)
        assert torch.equal(out, ref), env
    # ... and a second, differently shaped solve on the same engine (the arena is re-laid-out and zeroed), then the first again
    inp2 = make_inputs(3, 700, seed=96, lengths=[700, 333, 90])
    _solve(base, inp2, 2, "euler", kw)
    assert torch.equal(_solve(base, inp, 3, "euler", kw), ref)


def test_nonfinite_guard_and_recovery(sd, cfg_params):
    """(a) f16 operands overflow at 65504 where the fp32 reference does not -- the engine must SAY so: the boundary
    kernel raises a host-visible flag on NaN / Inf outputs (st_output_status; CFMDecoder(check_finite=True) raises with the remedy);
    (b) a call that produced NaN must not poison later calls of the same layout: ragged tile skipping leaves stale frames that only
    don't-care positions read, but 0 x NaN = NaN -- after a flagged call the engine re-zeroes its workspace by itself."""
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, check_finite=True)
    d.estimator.load_state_dict(sd)
    d = d.to("cuda:0")
    kw = _kw(cfg_params, 3.0)
    inp = make_inputs(4, 300, seed=12, lengths=[300, 180, 120, 40])          # ragged: tiles past 180 / 120 / 40 (+4) are skipped
    good = _solve(d, inp, 2, "euler", kw)
    eng = d.estimator.engine()
    assert not eng.output_nonfinite(torch.cuda.current_stream().cuda_stream)
    bad = dict(inp)
    bad["mu"] = inp["mu"].clone(); bad["mu"][:, :, :] = float("nan")          # every frame, padded ones included
    with pytest.raises(FloatingPointError, match="bf16"):
        _solve(d, bad, 2, "euler", kw)
    again = _solve(d, inp, 2, "euler", kw)                                    # same layout, clean inputs
    assert torch.isfinite(again).all() and torch.equal(again, good)
    # an activation beyond f16's range: finite in the fp32 reference, Inf / NaN with f16 operands -> flagged; bf16 carries it
    big = dict(inp); big["mu"] = inp["mu"] * 3e4
    with pytest.raises(FloatingPointError):
        _solve(d, big, 2, "euler", kw)
    d16 = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256, operand_dtype="bf16", check_finite=True)
    d16.estimator.load_state_dict(sd)
    d16 = d16.to("cuda:0")
    assert torch.isfinite(_solve(d16, big, 2, "euler", kw)).all()


def test_repack_after_rebind_needs_finalize(sd):
    """st_bind_param moves the engine to other fp32 tensors; a st_repack without st_finalize would leave
    instantiated graphs and the recorded re-pack jobs pointing at the old ones.  It now fails with ST_ERR_STATE, and binding drops
    the graphs."""
    from stabletts_amd import _lib
    from stabletts_amd.flow_matching import CFMDecoder
    d = CFMDecoder(128, 128, 256, 128, 1024, 4, 6, 3, 0.1, 256)
    d.estimator.load_state_dict(sd)
    d = d.to("cuda:0")
    eng = d.estimator.engine()
    stream = torch.cuda.current_stream().cuda_stream
    eng.repack(stream)                                              # in-place update path: fine
    name, p = next(iter(d.estimator.named_parameters()))
    other = p.detach().clone()
    shape = (ctypes.c_int64 * other.dim())(*other.shape)
    assert eng.lib.st_bind_param(eng.handle, name.encode(), ctypes.c_void_p(other.data_ptr()), shape, other.dim()) == 0
    with pytest.raises(_lib.NativeError, match="st_finalize"):
        eng.repack(stream)
    assert eng.lib.st_finalize(eng.handle) == 0
    eng.repack(stream)
