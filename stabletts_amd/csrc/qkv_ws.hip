// Fused q / k / v projection + RoPE (diffusion_transformer.py:59-61,74-75,180-198) as a WEIGHT-STATIONARY, persistent kernel.
//
// The projection is a 1x1 convolution 256 -> 768 with K = 256: 25 GFLOP against 136 MB of HBM traffic per layer launch at the
// headline size -- bound by memory (22 us at the achievable HBM rate; 10 us of matrix time).  On the generic conv tile
// (conv_gemm2_kernel<EPI_QKV>, 256 channels x 128 frames, ONE weight buffer, two blocks per CU) it takes 47 us: every block
// walks 4 stages of K = 64 and each stage exposes one LDS-DMA round trip for its 32-KiB weight tile, the 3 x 512 blocks run as
// three lock-step rounds and each round ends in an epilogue during which nothing is loaded.
//
// Here the WEIGHTS live in registers and the activations stream:
//   * a block = 8 waves owns ONE plane (q, k or v: 256 output channels); wave w owns channels 32 w .. 32 w + 32 and keeps their
//     16 fragments (K = 256 = 16 k-steps x 4 registers = 64 VGPRs) for the whole launch, loaded as 16 coalesced 1-KiB pieces from
//     a fragment-ordered copy of the weight (launch_pack_qkv_frag).  No weight byte crosses LDS.
//   * the block walks a list of 64-frame activation tiles (one frame-tile position tf, items lane, lane + L, lane + 2L ...): a
//     tile = 64 frames x 256 channels x 2 B = 32 KiB arrives by LDS-DMA into one of two slots (4 chunk images of 64 rows x 128 B
//     with the source-side XOR swizzle of the conv kernels: conflict-free ds_read_b128 B-fragments), one tile ahead of the
//     MFMAs; every wave reads the whole tile (its 32 channels x 64 frames: 32 MFMAs per tile, 1.0 fragment reads per MFMA).
//   * because tf is fixed per block, the RoPE cos / sin rows of its 64 frames are loop invariants, parked in LDS once (8 KB).
//   * epilogue per tile: RoPE + q scaling in the accumulator registers (same expressions as g2_epilogue_qkv: results are
//     bit-identical to the generic tile, tests/test_gpu_engine.py), one 16-bit LDS image per tile (two buffers), full-line (128 B) rows out:
//     q / k [item][H][T][64], v^T [item][H][64][Tp] in the P.V operand's key order (attention.hip).
// Per tile and wave the counts are FIXED -- 4 LDS-DMA pieces in (rows past T and tiles past the end of the list read the zero
// page), 4 row stores out (rows past T go to a sink) -- so `s_waitcnt vmcnt(4)` at the top of a tile retires exactly "everything
// up to this tile's pieces" (gfx950 counts loads and stores on the one in-order counter).  ONE barrier per tile:
//   iteration i:  wait(tile i) . barrier . issue tile i+1 . 4 stores of tile i-1 . 32 x (2 reads + 2 MFMAs) . image i
//   RAW  tile i is read after the barrier of iteration i, which every wave passes after its own pieces of tile i have landed;
//        image i-1 is read (by the stores) after that barrier too, and every wave wrote its part of it before arriving there
//   WAR  ring slot (i+1) % 2 held tile i-1, whose last read precedes each wave's arrival at the barrier of iteration i;
//        image buffer i % 2 is rewritten in iteration i, after that barrier, and its last readers were the stores of tile i-2,
//        issued (row data in registers) in iteration i-1
// so the stores of a tile, the LDS-DMA of the next and the epilogue arithmetic of one wave run beside the MFMAs of the others.
// Grid: 3 planes x G groups, G = tiles_f x L with 3 G <= 256 (one block per CU, 140 KB of LDS); consecutive block ids of the
// XCD-aware numbering are the three planes of one group, so the same activation tile is fetched into one L2 three times in a row.
#include "common.h"
#include "launch.h"
#include <cstdlib>
#include <type_traits>

#ifndef ST_PRIO_QWS
#define ST_PRIO_QWS 0      // 1: s_setprio 1 around the MFMA clusters (the round-4 form; without it the solve is 0.2-0.5 % faster per kernel family, paired: profiles/r05_ab_setprio.txt)
#endif

namespace st {

constexpr int kQwsTile = 64 * 512, kQwsRing = 2, kQwsPitch = 144, kQwsImage = 256 * kQwsPitch;      // 32,768 / 36,864 B
// cos, sin rows of the block's 64 frames, 16 floats each at a pitch of 20: with 16 the epilogue's float4 reads (lane = frame) are
// 4-way bank conflicts (16 instead of 4 LDS cycles per ds_read_b128; SQ_LDS_BANK_CONFLICT exceeded the kernel's busy LDS cycles)
constexpr int kQwsRopePitch = 20, kQwsRope = 2 * 64 * kQwsRopePitch * 4;
constexpr int kQwsLds = kQwsRing * kQwsTile + 2 * kQwsImage + kQwsRope;                                // 149,504 B

#define ST_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// VAR: developer ablations, compile-time (tools/micro/qkv_bench.hip; results are garbage unless 0 or 64): 1 = plain stores, 2 = no
// stores, 4 = LDS-DMA of the first two tiles only, 8 = no MFMAs, 16 = no image writes (+ no RoPE), 32 = one barrier per tile,
// 64 = s_memtime stamps per loop segment into g.dbg.  (As a run-time argument the same switches cost the loop its schedule and
// 12 bytes of scratch per lane.)
template <class P, int VAR>
__global__ __launch_bounds__(512, 1)
void qkv_ws_kernel(const ConvGemmArgs g, int L) {
    constexpr int var = VAR;
    using vec8 = typename P::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int T = g.T, H = g.n_heads, Tp = g.Tp;
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    const int tiles_f = (T + 63) >> 6;

    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= 3 * tiles_f * L) return;
    const int plane = lin % 3, grp = lin / 3;
    const int tf = grp % tiles_f, first = grp / tiles_f;
    const int t0 = tf * 64;

    // the block's work list: items first, first + L, ... whose tile tf is needed (ragged batches: t_lim), as a 64-bit mask built
    // with ONE vector load before the loop -- a load inside the loop would put a compiler-generated vmcnt(0) into the pipeline
    unsigned long long todo;
    {
        const int n = first + lane * L;
        bool need = n < g.n_items;
        if (need && g.t_lim) need = t0 < g.t_lim[n % g.t_lim_mod];
        todo = __ballot(need);
    }
    if (todo == 0) return;
    auto pop_item = [&]() {       // next item of the list, g.n_items when it is exhausted (wave-uniform scalar arithmetic)
        if (todo == 0) return g.n_items;
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        return first + j * L;
    };
    int ncur = pop_item(), n1 = pop_item();

    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)smem);
    const unsigned char* zeros = (const unsigned char*)g.zeros;
    unsigned char* stage0 = smem + kQwsRing * kQwsTile;      // two image buffers

    // ---- LDS-DMA of one activation tile: 32 pieces of 8 rows x 128 B; wave w moves pieces 4 w .. 4 w + 3 = rows 32 (w & 1) ..
    // + 32 of channel chunk w >> 1
    const int chunk = wave >> 1;
    unsigned voff[4]; bool vrow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = ((wave & 1) * 4 + k) * 8 + (lane >> 3);
        vrow[k] = t0 + row < T;
        voff[k] = (unsigned)((t0 + row) * 512 + chunk * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
    auto issue_tile = [&](int n, int slot) {
        const bool unit = n < g.n_items;
        const unsigned char* hb = (const unsigned char*)g.a0 + (size_t)((unit ? n : 0) % g.a0_mod) * T * 512;
        const unsigned dst = lds0 + (unsigned)(slot * kQwsTile + chunk * 8192 + (wave & 1) * 4096);
#pragma unroll
        for (int k = 0; k < 4; ++k) glds16bo((unit && vrow[k]) ? hb + voff[k] : zeros, dst + k * 1024);
    };
    issue_tile(ncur, 0);

    // ---- the wave's weights: 16 fragments (lane (channel l31, hi) holds K slots ks*16 + hi*8 .. +8) from the fragment-ordered copy
    // of the packed q/k/v weight (launch_pack_qkv_frag: [plane][wave][k-step][lane][8]) -- 16 coalesced 1-KiB loads per wave; read
    // straight from the row-major weight the same fragments are 32 scattered 32-byte pieces per instruction (prologue 3.5 -> 1 us)
    vec8 wf[16];
    {
        const unsigned char* wfrag = (const unsigned char*)g.w_frag + ((size_t)(plane * 8 + wave) * 16 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ks] = as_vec8<P>(*(const uint4*)(wfrag + ks * 1024));
    }
    // bias as the C operand of a tile's first MFMAs.  q / k: D = W . X^T, lane = frame, registers = channels; v: D = X . W^T (the
    // SAME two fragments with the operand roles swapped), lane = channel, registers = frames -- 4 consecutive frames per register
    // group, i.e. 8-byte pieces of the [channel][frame] image instead of 2-byte ones
    const bool vplane = plane == 2;
    f32x16_t bt;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) {
            if (vplane) { const float b1 = g.bias[plane * 256 + wave * 32 + l31]; bv = make_float4(b1, b1, b1, b1); }
            else bv = *(const float4*)(g.bias + plane * 256 + wave * 32 + 8 * q4 + 4 * hi);
        }
        bt[4 * q4 + 0] = bv.x; bt[4 * q4 + 1] = bv.y; bt[4 * q4 + 2] = bv.z; bt[4 * q4 + 3] = bv.w;
    }
    const int head = wave >> 1, half = wave & 1;      // q / k: the wave owns head dims 32 half .. + 32 of head `head`
    const bool rope = plane < 2 && half == 0;         // partial RoPE: pairs (d, d + 16), d < 16 -- all inside the first half
    // RoPE rows of the block's 64 frames (tf is fixed per block): [frame][16] cos, then sin, in LDS -- read back by ds_read in the
    // epilogue (32 registers per lane otherwise; an in-loop global load would count in vmcnt)
    float* ropeT = (float*)(smem + kQwsRing * kQwsTile + 2 * kQwsImage);
    if (plane < 2) {
        const int fl = (tid & 255) >> 2, q = tid & 3;
        const int tl = t0 + fl < T ? t0 + fl : T - 1;
        const float4 v = *(const float4*)((tid < 256 ? g.rope_cos : g.rope_sin) + (size_t)tl * 16 + 4 * q);
        *(float4*)(ropeT + (tid < 256 ? 0 : 64 * kQwsRopePitch) + fl * kQwsRopePitch + 4 * q) = v;
    }
    const float sc = plane == 0 ? g.qscale : 1.0f;
    // B-fragment offsets inside a chunk image (row = frame l31 of fragment 0; fragment 1 = + 32 rows: same swizzle term)
    unsigned radr[4];
#pragma unroll
    for (int ksl = 0; ksl < 4; ++ksl) radr[ksl] = (unsigned)(l31 * 128 + (((ksl * 2 + hi) ^ ((l31 >> 1) & 7)) << 4));
    unsigned char* sink = (unsigned char*)g.sink + (size_t)(blockIdx.x & 63) * 1024 + lane * 16;
    // every ordinary load of this kernel has been issued: retire them HERE, where hipcc can see it (opaque uses), so that its
    // scoreboard is empty inside the loop -- otherwise it guards the first use of each weight fragment with its own
    // `s_waitcnt vmcnt(15 - ks)`, which inside the loop would wait for the LDS-DMA pieces just issued
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(wf[ks]));
    asm volatile("" : "+v"(bt));
    // var & 64: s_memtime stamps per loop segment -> g.dbg[block][wave 0 / 4][8] = top wait, barrier A, DMA issue, reads + MFMAs,
    // epilogue (image), barrier B, stores, prologue (developer tool: tools/micro/qkv_bench.hip)
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define QWS_STAMP(K) if constexpr ((var & 64)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm[K] += t_ - tlast; tlast = t_; }
    if constexpr ((var & 64)) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm[7] = t_ - tstart; tlast = t_; }
    // the whole tile loop exists twice, once per operand orientation (two orientations inside one loop body cost 160 B of scratch)
    auto run_tiles = [&](auto vtag) {
    constexpr bool V = decltype(vtag)::value;
    // rows of the image of tile `n_of` (buffer `ib`) out: 4 full-line stores per wave, always issued (rows outside the tensor -> sink)
    auto store_rows = [&](int n_of, int ib) {
        unsigned char* stage = stage0 + ib * kQwsImage;
        const int ncur = n_of;
        const int rsub = lane >> 3, seg = lane & 7;
        uint4 rv[4]; unsigned char* rp[4];
        if constexpr (!V) {
            unsigned char* dst = (unsigned char*)(plane == 0 ? g.q : g.k);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rowid = (k * 8 + wave) * 8 + rsub;
                const int hd = rowid >> 6, f = rowid & 63;
                rv[k] = *(const uint4*)(stage + rowid * kQwsPitch + seg * 16);
                unsigned char* p = dst + (((size_t)ncur * H + hd) * T + t0 + f) * 128 + seg * 16;
                rp[k] = t0 + f < T ? p : sink;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = (k * 8 + wave) * 8 + rsub;
                rv[k] = *(const uint4*)(stage + ch * kQwsPitch + seg * 16);
                const int tcol = t0 + seg * 8;
                unsigned char* p = (unsigned char*)g.vt + ((((size_t)ncur * H + (ch >> 6)) * 64 + (ch & 63)) * Tp + tcol) * 2;
                rp[k] = tcol < Tp ? p : sink;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr ((var & 2)) asm volatile("" :: "v"(rv[k].x), "v"(rv[k].y), "v"(rv[k].z), "v"(rv[k].w));
            else if constexpr ((var & 1)) *(uint4*)rp[k] = rv[k];
            else store_row16(rp[k], rv[k]);
        }
    };
    int slot = 0, nprev = g.n_items, last_ib = 0;
    for (int i = 0; ; ++i) {
        // ---- this tile's pieces have landed (everything but the last 4 operations of this wave: the previous tile's stores), everyone's have
        __builtin_amdgcn_sched_barrier(0);
        if (i <= 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (i = 0: also the weights; the RoPE rows are in LDS.  i = 1: nothing younger than tile 1's pieces yet)
        else if constexpr ((var & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                         // younger than tile i's pieces: the 4 stores of tile i-2
        QWS_STAMP(0)
        ST_RAW_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        QWS_STAMP(1)
        if constexpr (!(var & 4)) issue_tile(n1, slot ^ 1); else issue_tile(g.n_items, slot ^ 1);
        QWS_STAMP(2)
        __builtin_amdgcn_sched_barrier(0);
        if (i > 0) store_rows(nprev, (i - 1) & 1);      // the previous tile's image is complete: every wave wrote its part before the barrier
        __builtin_amdgcn_sched_barrier(0);
        QWS_STAMP(6)
        unsigned char* stage = stage0 + (i & 1) * kQwsImage;
        // ---- 32 channels x 64 frames x K 256: acc[b] = bias + W . X^T, k-steps in order (the generic tile's order)
        f32x16_t acc[2];
        {
            // the 8 B-fragments of chunk c + 1 are read while the 8 MFMAs of chunk c run (two fragment sets: hipcc left to itself
            // reads one fragment, waits for it, issues one MFMA -- an exposed LDS latency per MFMA)
            const unsigned base = lds0 + (unsigned)(slot * kQwsTile);
            unsigned ad[4];
#pragma unroll
            for (int ksl = 0; ksl < 4; ++ksl) ad[ksl] = base + radr[ksl];
            vec8 bf[2][4][2];
            auto load_chunk = [&](int c, int set) {
#pragma unroll
                for (int ksl = 0; ksl < 4; ++ksl) {
                    bf[set][ksl][0] = as_vec8<P>(lds_read16(ad[ksl] + c * 8192));
                    bf[set][ksl][1] = as_vec8<P>(lds_read16(ad[ksl] + c * 8192 + 4096));
                }
            };
            {
                load_chunk(0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3) load_chunk(c + 1, (c + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ST_PRIO_QWS) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ksl = 0; ksl < 4; ++ksl)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if constexpr ((var & 8)) { asm volatile("" :: "v"(bf[c & 1][ksl][b])); if (c == 0 && ksl == 0) acc[b] = bt; }
                            else if constexpr (V) {
                                if (c == 0 && ksl == 0) acc[b] = P::mfma(bf[0][0][b], wf[0], bt);
                                else acc[b] = P::mfma(bf[c & 1][ksl][b], wf[c * 4 + ksl], acc[b]);
                            } else {
                                if (c == 0 && ksl == 0) acc[b] = P::mfma(wf[0], bf[0][0][b], bt);
                                else acc[b] = P::mfma(wf[c * 4 + ksl], bf[c & 1][ksl][b], acc[b]);
                            }
                        }
                    if constexpr (ST_PRIO_QWS) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue into the 16-bit image
        QWS_STAMP(3)
        if constexpr ((var & 16)) asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
        else if constexpr (!V) {      // q / k: image [head][frame][64] (pitch 144 B)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int fl = b * 32 + l31;
                f32x16_t r = acc[b];
                if (rope) {
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const float4 cs = *(const float4*)(ropeT + fl * kQwsRopePitch + 8 * q4 + 4 * hi);
                        const float4 sn = *(const float4*)(ropeT + 64 * kQwsRopePitch + fl * kQwsRopePitch + 8 * q4 + 4 * hi);
                        const float cc[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x1 = r[4 * q4 + e], x2 = r[4 * (q4 + 2) + e];
                            rope_rot(x1, x2, cc[e], ss[e]);
                            r[4 * q4 + e] = x1; r[4 * (q4 + 2) + e] = x2;
                        }
                    }
                }
                unsigned char* row = stage + (head * 64 + fl) * kQwsPitch + half * 64 + 4 * hi * 2;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(uint2*)(row + 16 * q4) = scale_pack4<P>(r[4 * q4 + 0], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3], sc);
            }
        } else {              // v: image [channel][frame] in the P.V operand's key order (bits 2 <-> 3 of the frame index inside every 16)
            const bool tail = t0 + 64 > T;        // only the item's last tile has frames past T (stored as zeros)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    // registers 4 q4 .. + 3 = frames b*32 + 8 q4 + 4 hi + (0..3); position = frame with bits 2 and 3 swapped
                    const int f0 = b * 32 + 8 * q4 + 4 * hi;
                    const int pos = b * 32 + (q4 >> 1) * 16 + hi * 8 + (q4 & 1) * 4;
                    float v0 = acc[b][4 * q4 + 0], v1 = acc[b][4 * q4 + 1], v2 = acc[b][4 * q4 + 2], v3 = acc[b][4 * q4 + 3];
                    if (tail) {
                        v0 = t0 + f0 + 0 < T ? v0 : 0.0f; v1 = t0 + f0 + 1 < T ? v1 : 0.0f;
                        v2 = t0 + f0 + 2 < T ? v2 : 0.0f; v3 = t0 + f0 + 3 < T ? v3 : 0.0f;
                    }
                    *(uint2*)(stage + (wave * 32 + l31) * kQwsPitch + pos * 2) = pack4<P>(v0, v1, v2, v3);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QWS_STAMP(4)
        nprev = ncur; last_ib = i & 1;
        if (n1 >= g.n_items) break;
        ncur = n1; n1 = pop_item();
        slot ^= 1;
    }
    // the last tile's rows
    __builtin_amdgcn_sched_barrier(0);
    ST_RAW_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    store_rows(nprev, last_ib);
    };
    if (vplane) run_tiles(std::true_type{}); else run_tiles(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA piece may land after the block has given its LDS back
    if constexpr (var & 64) if (g.dbg && lane == 0 && (wave & 3) == 0 && blockIdx.x < 64) {
        unsigned long long* d = g.dbg + (size_t)(blockIdx.x * 2 + (wave >> 2)) * 8;
        for (int k = 0; k < 8; ++k) d[k] = tm[k];
    }
#undef QWS_STAMP
}

#if defined(ST_QWS_VAR) && !defined(ST_DEVTOOLS)
#error "ST_QWS_VAR (ablation builds: results are garbage) needs -DST_DEVTOOLS"
#endif
#ifndef ST_QWS_VAR
#define ST_QWS_VAR 0
#endif

template <class P>
static hipError_t launch_qkv_ws_t(const ConvGemmArgs& a, hipStream_t s) {
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)qkv_ws_kernel<P, ST_QWS_VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, kQwsLds);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    if (!a.zeros || !a.sink || !a.w_frag || a.cout != 768 || a.c0 != 256 || a.c1 || a.c2 || a.n_heads != 4 || !a.q || !a.k || !a.vt ||
        !a.rope_cos || !a.rope_sin || a.Tp < ((a.T + 63) & ~63) || a.w_item_stride || a.ksplit > 1) return hipErrorInvalidValue;
    const int tiles_f = (a.T + 63) / 64;
    int L = 85 / tiles_f;
    if (L < 1) L = 1;
    if (L < (a.n_items + 63) / 64) L = (a.n_items + 63) / 64;      // a block's work list is a 64-bit mask
    if (L > a.n_items) L = a.n_items;
    const int grid = ((3 * tiles_f * L + 7) / 8) * 8;
    hipLaunchKernelGGL((qkv_ws_kernel<P, ST_QWS_VAR>), dim3(grid), dim3(512), kQwsLds, s, a, L);
    return hipGetLastError();
}

hipError_t launch_qkv_ws(int dtype, const ConvGemmArgs& a, hipStream_t s) {
    return dtype == DT_BF16 ? launch_qkv_ws_t<OpBF16>(a, s) : launch_qkv_ws_t<OpF16>(a, s);
}

}  // namespace st
