// Weight gradient of a Conv1d as ONE "TN" GEMM over the frame axis, reading both operands where the backward pass left
// them -- row-major [frame][channel] -- with no transposed copies (autograd counterpart of nn.Conv1d's weight / bias
// gradient as exercised by CFMDecoder.compute_loss, models/flow_matching.py:69-100):
//
//     dW[co][j][ci] = sum_{n, t} dY[n][t][co] * X[n][t + j - taps/2][ci]          (zero where the shifted frame leaves the item)
//
// M = cout, N = taps * cin (one tap per N tile), K = the frames of the items a block owns.  The MFMA wants 8 consecutive k per
// lane for both operands, but k (the frame) is the STRIDED axis of both tensors.  The tiles are therefore staged in LDS
// exactly as they lie in memory -- [frame][channel], 128-byte rows, by LDS-DMA -- and the fragments are read with gfx950's
// transposing LDS read: ds_read_b64_tr_b16 hands lane L = 4a + b of a 16-lane group element b of the 8-byte chunk addressed by
// lanes a, a + 4, a + 8, a + 12 (probed on hardware: tools/micro/tr_probe.hip).  With source lane 4j + a addressing
// (frame f0 + j, channels c0 + 4a .. + 4), lane L receives frames f0 .. f0 + 3 of channel c0 + L: a k-run of 4; two reads make
// the 8 k-slots of a 32x32x16 operand.  Rows are XOR-swizzled at 16-byte granularity by ((frame >> 1) & 1) << 2 so that the
// four frames x 64 bytes a half-wave touches fall into four different bank quarters (no conflicts).
//
// Block = 256 (cout) x 256 (one tap's cin slice) output tile, 8 waves of 128 x 64, K stages of 32 frames in FOUR 32-KB buffers:
// the LDS-DMA of stage s + 3 is issued while stage s is computed and retired by a counted `s_waitcnt vmcnt(8)` that leaves the
// two younger stages in flight (with two 64-frame buffers and a full drain per stage the loop ran at the DMA round trip:
// 2.4 us per 64 frames, 35 % of the MFMA rate).  One barrier per stage publishes the stage and frees the buffer read last.
// The fp32 partial tile of the block's items goes through the conv kernels' coalesced EPI_F32 epilogue into
// plane s of partial[S][taps * cin][cout]; launch_wgrad_reduce adds the planes in order (deterministic).
#include "conv_gemm2_impl.h"
#include "train_launch.h"
#include <cstring>

namespace st {

struct WgradTnArgs {
    const void* dy; int cout;                    // [items * T][cout], 16 bit
    const void* x0; int c0; const void* x1; int c1;   // [items * T][c0], [items * T][c1] (channel concat; c1 may be 0)
    int taps, n_items, T, cps;                   // cps: 32-frame chunks per split (K range of a block; chunks never straddle items)
    const void* zeros;
    float* part_b;                               // [S][cout] column sums of the block's dY rows (bias-gradient partials), or null
};

namespace {

template <int OFF>
__device__ __forceinline__ uint2 tr_read(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <class P>
__device__ __forceinline__ typename P::vec8 join8(uint2 lo, uint2 hi) {
    return as_vec8<P>(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// acc + the sum of a fragment's 8 values (fp32 accumulation, products with 1 are exact): 4 dot instructions beside the MFMAs
template <class P>
__device__ __forceinline__ float sum8(uint2 lo, uint2 hi, float acc);
template <>
__device__ __forceinline__ float sum8<OpF16>(uint2 lo, uint2 hi, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, lo.x), one, acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, lo.y), one, acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, hi.x), one, acc, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, hi.y), one, acc, false);
}
template <>
__device__ __forceinline__ float sum8<OpBF16>(uint2 lo, uint2 hi, float acc) {      // bf16 -> fp32 is a 16-bit shift
    const unsigned w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xffff0000u);
    return acc;
}

}  // namespace

template <class P>
__global__ __launch_bounds__(512, 1) void wgrad_tn_kernel(const WgradTnArgs w, const ConvGemmArgs g) {
    constexpr int BC = 256, BF = 256, WC = 2, WF = 4, FC = 4, FF = 2;
    constexpr int KF = 32, TILE = KF * 512, BUF = 2 * TILE, NBUF = 4;          // 32 frames x 256 channels per operand; A tile + B tile per buffer
    using vec8 = typename P::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int cin = w.c0 + w.c1;
    const int half = w.taps / 2;
    const int nblk_ci = (cin + BF - 1) / BF;
    const int tiles_n = w.taps * nblk_ci, tiles_m = w.cout / BC;
    const int nchunk = (w.T + KF - 1) / KF, kchunks = w.n_items * nchunk;
    const int S = (kchunks + w.cps - 1) / w.cps;
    const int total = S * tiles_n * tiles_m;
    const int per_xcd = gridDim.x >> 3;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int tm = lin % tiles_m;
    const int tn = (lin / tiles_m) % tiles_n;
    const int s = lin / (tiles_m * tiles_n);
    const int j = tn / nblk_ci, cb = (tn % nblk_ci) * BF;       // tap and first input channel of this N tile
    const int mb = tm * BC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wf = wave / WC;
    const int T = w.T;
    const unsigned char* zeros = (const unsigned char*)w.zeros;

    // ---- LDS-DMA: 16 + 16 pieces of 1 KiB (8 frames x 128 B of one 64-channel block) per stage, 2 + 2 per wave
    const int prow = lane >> 3;
    auto issue = [&](int item, int t0, int buf) {
        unsigned char* Ab = smem + buf * BUF;
        unsigned char* Bb = Ab + TILE;
        const size_t rbase = (size_t)item * T;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave * 2 + k;                  // channel block = piece >> 2, frame group = piece & 3
            const int cbk = piece >> 2, row = (piece & 3) * 8 + prow;
            const int sseg = (lane & 7) ^ (((row >> 1) & 1) << 2);
            {   // A: dY frames t0 + row, channels mb + cbk*64 + sseg*8 .. + 8
                const int t = t0 + row;
                const unsigned char* src = (t < T) ? (const unsigned char*)w.dy + ((rbase + t) * w.cout + mb + cbk * 64 + sseg * 8) * 2 : zeros;
                glds16b(src, Ab + piece * 1024);
            }
            {   // B: X frames t0 + row + j - half of the same item, channels cb + cbk*64 + sseg*8 .. + 8 (x0 | x1)
                const int t = t0 + row + j - half;
                const int ch = cb + cbk * 64;
                const unsigned char* src = zeros;
                if (t >= 0 && t < T && ch < cin) {
                    if (ch < w.c0) src = (const unsigned char*)w.x0 + ((rbase + t) * w.c0 + ch + sseg * 8) * 2;
                    else           src = (const unsigned char*)w.x1 + ((rbase + t) * w.c1 + (ch - w.c0) + sseg * 8) * 2;
                }
                glds16b(src, Bb + piece * 1024);
            }
        }
    };

    // ---- fragment addresses (see the header): lane -> (16-lane group g, source index sI)
    const int gq = lane >> 4, sI = lane & 15;
    const int rlane = 8 * (gq >> 1) + (sI >> 2);                       // + 16 ks + 4 rd
    const unsigned swz = (unsigned)((sI >> 3) & 1) << 2;
    const unsigned seg0 = (unsigned)(2 * (gq & 1) + ((sI & 3) >> 1));
    const unsigned within = (unsigned)(8 * (sI & 1));
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    const unsigned rowb = (unsigned)rlane * 128u;
    // A operand (cout side): wave's 128 channels = channel blocks wc*2, wc*2+1; sub-block a: block a>>1, segment bit (a&1)<<2
    const unsigned aA0 = lds0 + (unsigned)wc * 8192u + rowb + ((seg0 ^ swz) << 4) + within;
    const unsigned aA1 = lds0 + (unsigned)wc * 8192u + rowb + (((seg0 | 4u) ^ swz) << 4) + within;
    // B operand (cin side): wave's 64 channels = channel block wf; sub-block b: segment bit b << 2
    const unsigned aB0 = lds0 + TILE + (unsigned)wf * 4096u + rowb + ((seg0 ^ swz) << 4) + within;
    const unsigned aB1 = lds0 + TILE + (unsigned)wf * 4096u + rowb + (((seg0 | 4u) ^ swz) << 4) + within;

    f32x16_t acc[FC][FF];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FF; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // bias-gradient partials as a by-product: the blocks of the first N tile (tap 0, first cin slice) see every dY row of their K
    // range exactly once; their two wf == 0 waves (one per 128-channel half) add up the A fragments they feed to the MFMAs.
    // Lane l of fragment a holds 8 frames of channel (l & 31); lanes l and l ^ 32 hold the two frame halves of a k-step.
    const bool do_bias = w.part_b != nullptr && tn == 0 && wf == 0;       // wave-uniform
    float csum[FC] = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const unsigned bo = (unsigned)buf * BUF;
#define ST_TN_KSTEP(KS)                                                                                             \
        {                                                                                                           \
            uint2 al[FC], ah[FC], bl[FF], bh[FF];                                                                   \
            al[0] = tr_read<(KS) * 2048>(aA0 + bo);        ah[0] = tr_read<(KS) * 2048 + 512>(aA0 + bo);            \
            al[1] = tr_read<(KS) * 2048>(aA1 + bo);        ah[1] = tr_read<(KS) * 2048 + 512>(aA1 + bo);            \
            al[2] = tr_read<4096 + (KS) * 2048>(aA0 + bo); ah[2] = tr_read<4096 + (KS) * 2048 + 512>(aA0 + bo);     \
            al[3] = tr_read<4096 + (KS) * 2048>(aA1 + bo); ah[3] = tr_read<4096 + (KS) * 2048 + 512>(aA1 + bo);     \
            bl[0] = tr_read<(KS) * 2048>(aB0 + bo);        bh[0] = tr_read<(KS) * 2048 + 512>(aB0 + bo);            \
            bl[1] = tr_read<(KS) * 2048>(aB1 + bo);        bh[1] = tr_read<(KS) * 2048 + 512>(aB1 + bo);            \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                     \
                         : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]), \
                           "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]) :: "memory");                          \
            vec8 bf0 = join8<P>(bl[0], bh[0]), bf1 = join8<P>(bl[1], bh[1]);                                        \
            if (do_bias) { _Pragma("unroll") for (int a = 0; a < FC; ++a) csum[a] = sum8<P>(al[a], ah[a], csum[a]); }   \
            _Pragma("unroll") for (int a = 0; a < FC; ++a) {                                                        \
                const vec8 af = join8<P>(al[a], ah[a]);                                                             \
                acc[a][0] = P::mfma(af, bf0, acc[a][0]);                                                            \
                acc[a][1] = P::mfma(af, bf1, acc[a][1]);                                                            \
            }                                                                                                       \
        }
        ST_TN_KSTEP(0) ST_TN_KSTEP(1)
#undef ST_TN_KSTEP
    };

    // ---- K loop over this block's range of 32-frame chunks (item = chunk / chunks per item)
    const int k0 = s * w.cps, nstage = min(kchunks, k0 + w.cps) - k0;
    auto issue_stage = [&](int st) { const int kx = k0 + st; issue(kx / nchunk, (kx % nchunk) * KF, st % NBUF); };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
        if (p < nstage) issue_stage(p);
    for (int st = 0; st < nstage; ++st) {
        // stage st has landed when at most the pieces of the younger stages in flight (4 per wave and stage) are outstanding
        const int younger = min(NBUF - 2, nstage - 1 - st);
#ifdef ST_TN_NO_DMA      // ablation builds (tools/ab_wgrad.sh): the loop without its DMA / without its reads + MFMAs
        ST_DMA_WAIT(0);
#else
        if (younger >= 2) ST_DMA_WAIT(8); else if (younger == 1) ST_DMA_WAIT(4); else ST_DMA_WAIT(0);
#endif
        __syncthreads();      // every wave's pieces of stage st are in LDS; every wave is done reading buffer (st - 1) % NBUF
#ifndef ST_TN_NO_DMA
        if (st + NBUF - 1 < nstage) issue_stage(st + NBUF - 1);
#endif
#ifndef ST_TN_NO_MMA
        compute(st % NBUF);
#endif
    }
    __syncthreads();
    if (do_bias) {
#pragma unroll
        for (int a = 0; a < FC; ++a) {
            const float v = csum[a] + __shfl_xor(csum[a], 32);
            if (lane < 32) w.part_b[(size_t)s * w.cout + mb + (wc * 2 + (a >> 1)) * 64 + (a & 1) * 32 + lane] = v;
        }
    }

    // partial tile -> plane s of partial[S][taps*cin][cout] (rows = this tap's input channels cb .. cb + 256 of cin)
    const int fvalid = min(BF, cin - cb);
    g2_epilogue<P, EPI_F32, BC, BF, WC, WF>(acc, (float*)smem, g, s, j * cin + cb, fvalid, mb, wave, lane);
}

template <class P>
static hipError_t launch_wgrad_tn_t(const WgradTnArgs& w, float* partial, int S, hipStream_t s) {
    constexpr int LDS = 4 * 32768;      // NBUF x (A tile + B tile)
    static bool attr_done_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_done_dev[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_tn_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_done_dev[dev_] = true;
    }
    const int cin = w.c0 + w.c1;
    if (!w.zeros || !partial || (w.cout % 256) || (w.c0 & 63) || (w.c1 & 63) || cin < 64 || w.cps < 1 || (w.taps != 1 && w.taps != 3)) return hipErrorInvalidValue;
    if (w.c1 && (w.c0 % 256)) return hipErrorInvalidValue;      // an N tile never straddles the two sources
    ConvGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.out32 = partial; g.cout = w.cout; g.T = w.taps * cin; g.n_items = S; g.mask_mod = 1; g.a0_mod = 1; g.a1_mod = 1;
    const int tiles = w.taps * ((cin + 255) / 256) * (w.cout / 256);
    const int total = tiles * S;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL((wgrad_tn_kernel<P>), dim3(grid), dim3(512), LDS, s, w, g);
    return hipGetLastError();
}

hipError_t launch_wgrad_tn(int dtype, const void* dy, int cout, const void* x0, int c0, const void* x1, int c1, int taps,
                           int n_items, int T, int cps, const void* zeros, float* partial, float* part_b, hipStream_t s) {
    WgradTnArgs w;
    w.part_b = part_b;
    w.dy = dy; w.cout = cout; w.x0 = x0; w.c0 = c0; w.x1 = x1; w.c1 = c1; w.taps = taps; w.n_items = n_items; w.T = T; w.cps = cps;
    w.zeros = zeros;
    const int kchunks = n_items * ((T + 31) / 32);
    const int S = (kchunks + cps - 1) / cps;
    return dtype == DT_BF16 ? launch_wgrad_tn_t<OpBF16>(w, partial, S, s) : launch_wgrad_tn_t<OpF16>(w, partial, S, s);
}

// part_b[rowblock][co] = sum over the block's 64 rows of dY[r][co] (the bias-gradient partials launch_bias_reduce adds up):
// block = 64 rows x 64 channels, thread = 8 channels (16 bytes) of one of 32 row pairs
template <class P>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const typename P::elem* dy, int cout, int64_t R, float* part_b) {
    __shared__ float red[32][64 + 1];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int ch0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t r = r0 + ty + 32 * u;
        if (r < R) {
            const typename P::vec8 x = as_vec8<P>(*(const uint4*)(dy + (size_t)r * cout + ch0 + tx * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = v[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x];
        part_b[(size_t)blockIdx.x * cout + ch0 + threadIdx.x] = t;
    }
}

hipError_t launch_colsum_rows(int dtype, const void* dy, int cout, int64_t R, float* part_b, hipStream_t s) {
    if (cout & 63) return hipErrorInvalidValue;
    dim3 grid((unsigned)((R + 63) / 64), (unsigned)(cout / 64));
    if (dtype == DT_BF16) hipLaunchKernelGGL((colsum_rows_kernel<OpBF16>), grid, dim3(256), 0, s, (const __bf16*)dy, cout, R, part_b);
    else                  hipLaunchKernelGGL((colsum_rows_kernel<OpF16>), grid, dim3(256), 0, s, (const _Float16*)dy, cout, R, part_b);
    return hipGetLastError();
}

}  // namespace st
