// Attention out-projection + gate + residual + LayerNorm_2 + adaLN modulate (diffusion_transformer.py:65,111,112,26) as a
// WEIGHT-STATIONARY, persistent kernel -- the second member of the qkv_ws.hip family (read that header first: work lists, the
// fragment-ordered weight copy, the counted vmcnt protocol with a FIXED number of LDS-DMA pieces and row stores per tile and wave).
//
// The op is a 1x1 convolution 256 -> 256 (8.4 GFLOP per layer launch at the headline size) around 198 MB of HBM traffic: the fp32
// residual stream in and out, the 16-bit attention output in, the 16-bit FFN operand out.  On the generic 256 x 256 tile it runs at
// 4.3 TB/s in lock-step rounds (LDS-DMA prologue, a K loop during which almost nothing is fetched, four epilogue passes); its cost
// to the solve is 3.1-3.7 ms.  Here a persistent block = 8 waves keeps the whole 256 x 256 weight in registers
// (wave w: output channels 32 w .. + 32, 16 fragments = 64 VGPRs) and EVERYTHING a tile needs arrives by LDS-DMA one tile ahead:
//   per 32-frame tile:  attention output 16 KiB (4 chunk images, swizzled: the B fragments)      2 pieces per wave
//                       residual rows x_1 32 x 1 KiB fp32                                        4 pieces per wave
//                       the item's gate / shift / scale rows 3 x 1 KiB (waves 0..2), the frame mask of the 32 frames (wave 3, 4-byte
//                       pieces: the row is only dword-aligned), zero page -> sink KiB (waves 4..7)      1 piece per wave
//   iteration i:  wait(tile i) . barrier A . issue tile i+2 (7 pieces) . 16 x (read + MFMA) . x_2 = x_1 + gate ((acc + b) mask) written
//                 INTO the staged residual rows (lane = frame; they double as the transposition stage, 260-float pitch) . barrier B .
//                 row walk: 4 rows per wave (lane = 4 channels) -- x_2 -> store (1 KiB), LayerNorm over the 256 channels (DPP /
//                 permlane sums), modulate, mask -> 16-bit store (512 B)                           8 stores per wave
//   THREE ring slots, two tiles (106 KB) in flight per CU: with one tile ahead the kernel sat at the generic tile's 4.4 TB/s
//   (54 KB in flight / 2.7 us of loaded latency = 20 GB/s per CU).  `s_waitcnt vmcnt(23)` at the top retires everything but
//   stores i-2, pieces i+1, stores i-1; RAW / WAR as in qkv_ws.hip (ring slot (i+2) % 3 held tile i-1, last read in the row walk of
//   iteration i-1, before each wave's arrival at barrier A of iteration i).
// No ordinary load sits inside the loop (even a wave-uniform one is issued as a vector load and guarded by a compiler-generated
// vmcnt(0) that would wait for the pieces just issued): the items' frame limits are parked in LDS up front.  The row arithmetic is g2_rows'
// (EPI_RESGATE + LayerNorm branch, conv_gemm2_impl.h) expression for expression: results are bit-identical to the generic tile.
#include "common.h"
#include "launch.h"
#include <type_traits>

#ifndef ST_PRIO_OWS
#define ST_PRIO_OWS 0      // 1: s_setprio 1 around the MFMA clusters (the round-4 form; without it the solve is 0.2-0.5 % faster per kernel family, paired: profiles/r05_ab_setprio.txt)
#endif

namespace st {

constexpr int kOwsAo = 32 * 512, kOwsXPitch = 1040, kOwsX = 32 * kOwsXPitch, kOwsConst = 3 * 1024 + 256;
constexpr int kOwsSlot = kOwsAo + kOwsX + kOwsConst;      // operand tile, residual rows (260-float pitch), gate / shift / scale rows, 32 mask values: 52,992 B
constexpr int kOwsRing = 3;
constexpr int kOwsLds = kOwsRing * kOwsSlot + 1024 + 256;      // + one sink KiB + the items' frame limits: 160,256 B

#define ST_RAW_BARRIER() asm volatile("s_barrier" ::: "memory")

// Developer ablations (-DST_DEVTOOLS -DST_OWS_VAR=bits; results are garbage unless 0 or 1): 1 = write-back instead of write-through row
// stores, 2 = the fp32 x_2 rows go to the sink (no residual write traffic), 4 = the 16-bit operand rows go to the sink, 8 = the residual
// rows are not fetched (zero page), 16 = the attention-output tile is not fetched, 32 = no MFMAs.
#if defined(ST_OWS_VAR) && !defined(ST_DEVTOOLS)
#error "ST_OWS_VAR (ablation builds: results are garbage) needs -DST_DEVTOOLS"
#endif
#ifndef ST_OWS_VAR
#define ST_OWS_VAR 0
#endif
constexpr int kOwsVar = ST_OWS_VAR;

template <class P>
__global__ __launch_bounds__(512, 1)
void oproj_ws_kernel(const ConvGemmArgs g, int L) {
    using vec8 = typename P::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int T = g.T;
    const int tiles_f = (T + 31) >> 5;

    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= tiles_f * L) return;
    const int tf = lin % tiles_f, first = lin / tiles_f;
    const int t0 = tf * 32;

    unsigned long long todo;      // the block's work list: items first, first + L, ... whose tile tf is needed
    {
        const int n = first + lane * L;
        bool need = n < g.n_items;
        if (need && g.t_lim) need = t0 < g.t_lim[n % g.t_lim_mod];
        todo = __ballot(need);
    }
    if (todo == 0) return;
    auto pop_item = [&]() {
        if (todo == 0) return g.n_items;
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        return first + j * L;
    };
    int ncur = pop_item(), n1 = pop_item(), n2 = pop_item();

    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)smem);
    const unsigned char* zeros = (const unsigned char*)g.zeros;
    // last needed frame of the block's items, parked in LDS (entry j = item first + j L): a load inside the loop -- even a
    // wave-uniform one, hipcc issues it as a vector load -- would put a compiler-generated vmcnt(0) into the pipeline
    int* tlimT = (int*)(smem + kOwsRing * kOwsSlot + 1024);
    if (wave == 0) {
        const int n = first + lane * L;
        int tl = T;
        if (n < g.n_items && g.t_lim) tl = min(T, g.t_lim[n % g.t_lim_mod]);
        tlimT[lane] = tl;
    }

    // ---- LDS-DMA of one tile: 2 operand pieces (chunk w >> 1, rows 16 (w & 1) + 8 k ..), 4 residual rows (4 w + k), 1 constant piece
    unsigned voffA[2]; bool vrowA[2]; unsigned voffX[4]; bool vrowX[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = (wave & 1) * 16 + k * 8 + (lane >> 3);
        vrowA[k] = t0 + row < T;
        voffA[k] = (unsigned)((t0 + row) * 512 + (wave >> 1) * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = wave * 4 + k;
        vrowX[k] = t0 + row < T;
        voffX[k] = (unsigned)((t0 + row) * 1024 + lane * 16);
    }
    auto issue_tile = [&](int n, int slot) {
        const bool unit = n < g.n_items;
        const int nn = unit ? n : 0;
        const unsigned char* ab = (const unsigned char*)g.a0 + (size_t)(nn % g.a0_mod) * T * 512;
        const unsigned char* xb = (const unsigned char*)g.out32 + (size_t)nn * T * 1024;
        const unsigned base = lds0 + (unsigned)(slot * kOwsSlot);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            glds16bo((unit && vrowA[k] && !(kOwsVar & 16)) ? ab + voffA[k] : zeros, base + (unsigned)((wave >> 1) * 4096 + (wave & 1) * 2048 + k * 1024));
#pragma unroll
        for (int k = 0; k < 4; ++k) glds16bo((unit && vrowX[k] && !(kOwsVar & 8)) ? xb + voffX[k] : zeros, base + (unsigned)(kOwsAo + (wave * 4 + k) * kOwsXPitch));
        {   // waves 0, 1, 2: the item's gate / adaLN shift / adaLN scale rows; wave 3: the frame mask of the tile's 32 frames (a row
            // that is only dword-aligned: 4-byte pieces, frames past T clamped like the generic epilogue); waves 4..7: zero page -> sink
            const float* ad = g.ln_ada + (size_t)nn * g.ln_ada_stride;
            if (wave == 3) {
                const int t = t0 + (lane & 31);
                const float* mp = g.mask ? g.mask + (size_t)(nn % g.mask_mod) * T + (t < T ? t : T - 1) : nullptr;
                glds4bo((unit && mp && lane < 32) ? (const void*)mp : (const void*)zeros, base + (unsigned)(kOwsAo + kOwsX + 3 * 1024));
            } else {
                const float* src = wave == 0 ? g.gate + (size_t)nn * g.gate_stride : wave == 1 ? ad + g.ln_shift_off : ad + g.ln_scale_off;
                const unsigned char* p = (unit && wave < 3) ? (const unsigned char*)src + lane * 16 : zeros;
                glds16bo(p, wave < 3 ? base + (unsigned)(kOwsAo + kOwsX + wave * 1024) : lds0 + (unsigned)(kOwsRing * kOwsSlot));
            }
        }
    };
    issue_tile(ncur, 0);
    issue_tile(n1, 1);

    // ---- the wave's weights (fragment-ordered copy, plane 0) and constants
    vec8 wf[16];
    {
        const unsigned char* wfrag = (const unsigned char*)g.w_frag + ((size_t)wave * 16 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ks] = as_vec8<P>(*(const uint4*)(wfrag + ks * 1024));
    }
    const int ch = lane * 4;      // the lane's channels in the row walk
    float4 bias4[4];              // bias of the lane's accumulator channels 32 w + 8 q4 + 4 hi .. + 3
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) bias4[q4] = g.bias ? *(const float4*)(g.bias + wave * 32 + 8 * q4 + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned radr[4];
#pragma unroll
    for (int ksl = 0; ksl < 4; ++ksl) radr[ksl] = (unsigned)(l31 * 128 + (((ksl * 2 + hi) ^ ((l31 >> 1) & 7)) << 4));
    unsigned char* sink = (unsigned char*)g.sink + (size_t)(blockIdx.x & 63) * 1024 + lane * 16;
    const bool hasmask = g.mask != nullptr;
    const bool mout = g.ln_mask_out;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(wf[ks]));      // every ordinary load retired before the loop (qkv_ws.hip)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) asm volatile("" : "+v"(bias4[q4].x), "+v"(bias4[q4].y), "+v"(bias4[q4].z), "+v"(bias4[q4].w));

    int slot = 0;
    for (int i = 0; ; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (tile 1's pieces too: once)
        else if (i == 1) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");      // younger than tile 1's pieces: tile 2's 7 pieces, the 8 stores of tile 0
        else asm volatile("s_waitcnt vmcnt(23)" ::: "memory");                  // younger than tile i's pieces: stores i-2, pieces i+1, stores i-1
        ST_RAW_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        {
            int s2 = slot + 2; if (s2 >= kOwsRing) s2 -= kOwsRing;
            issue_tile(n2, s2);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned base = lds0 + (unsigned)(slot * kOwsSlot);
        unsigned char* slotp = smem + slot * kOwsSlot;
        // ---- 32 channels x 32 frames x K 256, k-steps in order (the generic tile's order), accumulator from zero like g2_init_acc
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        {
            unsigned ad[4];
#pragma unroll
            for (int ksl = 0; ksl < 4; ++ksl) ad[ksl] = base + radr[ksl];
            vec8 bf[2][4];
            auto load_chunk = [&](int c, int set) {
#pragma unroll
                for (int ksl = 0; ksl < 4; ++ksl) bf[set][ksl] = as_vec8<P>(lds_read16(ad[ksl] + c * 4096));
            };
            load_chunk(0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 3) load_chunk(c + 1, (c + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST_PRIO_OWS) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ksl = 0; ksl < 4; ++ksl) { if constexpr (kOwsVar & 32) asm volatile("" :: "v"(bf[c & 1][ksl])); else acc = P::mfma(wf[c * 4 + ksl], bf[c & 1][ksl], acc); }
                if constexpr (ST_PRIO_OWS) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the residual update IN the staged rows: x_2[frame][channel] = x_1 + gate ((acc + b) mask); lane = frame l31, registers =
        // channels 8 q4 + 4 hi + e of the wave's 32 -- the residual rows (260-float pitch) double as the transposition stage
        {
            const float* gateT = (const float*)(slotp + kOwsAo + kOwsX);
            const float m = hasmask ? ((const float*)(slotp + kOwsAo + kOwsX + 3 * 1024))[l31] : 1.0f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c0 = wave * 32 + 8 * q4 + 4 * hi;
                float* xp = (float*)(slotp + kOwsAo + l31 * kOwsXPitch) + c0;
                const float4 xin = *(const float4*)xp;
                const float4 gt = *(const float4*)(gateT + c0);
                float4 v;
                v.x = xin.x + gt.x * ((acc[4 * q4 + 0] + bias4[q4].x) * m); v.y = xin.y + gt.y * ((acc[4 * q4 + 1] + bias4[q4].y) * m);
                v.z = xin.z + gt.z * ((acc[4 * q4 + 2] + bias4[q4].z) * m); v.w = xin.w + gt.w * ((acc[4 * q4 + 3] + bias4[q4].w) * m);
                *(float4*)xp = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ST_RAW_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        // ---- row walk: rows 4 w .. 4 w + 3 of the tile, lane = channels 4 lane .. + 3 (g2_rows: the LayerNorm branch of EPI_RESGATE)
        {
            const unsigned char* cst = slotp + kOwsAo + kOwsX;
            const float4 sh = *(const float4*)(cst + 1024 + lane * 16);
            const float4 sc = *(const float4*)(cst + 2048 + lane * 16);
            const int tlim = tlimT[(ncur - first) / L];
            const float* mtile = (const float*)(cst + 3 * 1024);
            float4 v[4]; float m[4]; bool ok[4]; int tt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = wave * 4 + u;
                tt[u] = t0 + r; ok[u] = t0 + r < tlim;
                m[u] = hasmask ? mtile[r] : 1.0f;
                v[u] = *(const float4*)(slotp + kOwsAo + r * kOwsXPitch + lane * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float* p = g.out32 + ((size_t)ncur * T + tt[u]) * 256 + ch;
                void* dst = (ok[u] && !(kOwsVar & 2)) ? (void*)p : (void*)sink;
                if constexpr (kOwsVar & 1) *(float4*)dst = v[u]; else store_row16(dst, v[u]);
            }
            float mean[4], var[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) mean[u] = wave_sum(v[u].x + v[u].y + v[u].z + v[u].w) * (1.0f / 256.0f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u].x -= mean[u]; v[u].y -= mean[u]; v[u].z -= mean[u]; v[u].w -= mean[u];
                var[u] = wave_sum(v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w) * (1.0f / 256.0f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float rs = 1.0f / sqrtf(var[u] + 1e-5f);
                const float mm = mout ? m[u] : 1.0f;
                const float h0 = (v[u].x * rs * (1.0f + sc.x) + sh.x) * mm, h1 = (v[u].y * rs * (1.0f + sc.y) + sh.y) * mm;
                const float h2 = (v[u].z * rs * (1.0f + sc.z) + sh.z) * mm, h3 = (v[u].w * rs * (1.0f + sc.w) + sh.w) * mm;
                unsigned char* p = (unsigned char*)g.ln_h16 + (((size_t)ncur * T + tt[u]) * 256 + ch) * 2;
                void* dst = (ok[u] && !(kOwsVar & 4)) ? (void*)p : (void*)sink;
                if constexpr (kOwsVar & 1) *(uint2*)dst = pack4<P>(h0, h1, h2, h3); else store_row8(dst, pack4<P>(h0, h1, h2, h3));
            }
        }
        if (n1 >= g.n_items) break;
        ncur = n1; n1 = n2; n2 = pop_item();
        if (++slot == kOwsRing) slot = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA piece may land after the block has given its LDS back
}

template <class P>
static hipError_t launch_oproj_ws_t(const ConvGemmArgs& a, hipStream_t s) {
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)oproj_ws_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, kOwsLds);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    if (!a.zeros || !a.sink || !a.w_frag || a.cout != 256 || a.c0 != 256 || a.c1 || a.c2 || !a.out32 || !a.ln_h16 || a.ln_film || !a.ln_ada ||
        !a.gate || a.out16 || a.res32 || a.branch32 || a.out32_readonly || a.add32 || a.w_item_stride || a.ksplit > 1) return hipErrorInvalidValue;
    const int tiles_f = (a.T + 31) / 32;
    // (block count: 192, 224 and 256 blocks measured the same inside the solve, paired -- profiles/r05_ab_attn_xcd_ws_blocks.txt)
    int L = 240 / tiles_f;
    if (L < 1) L = 1;
    if (L < (a.n_items + 63) / 64) L = (a.n_items + 63) / 64;      // a block's work list is a 64-bit mask
    if (L > a.n_items) L = a.n_items;
    const int grid = ((tiles_f * L + 7) / 8) * 8;
    hipLaunchKernelGGL((oproj_ws_kernel<P>), dim3(grid), dim3(512), kOwsLds, s, a, L);
    return hipGetLastError();
}

hipError_t launch_oproj_ws(int dtype, const ConvGemmArgs& a, hipStream_t s) {
    return dtype == DT_BF16 ? launch_oproj_ws_t<OpBF16>(a, s) : launch_oproj_ws_t<OpF16>(a, s);
}

}  // namespace st
