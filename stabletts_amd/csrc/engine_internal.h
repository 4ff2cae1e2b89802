// Host-side internals shared by engine.cpp (inference: parameter store, packing, solve loop) and engine_train.cpp
// (training: forward that keeps activations + backward).  Not part of the C ABI.
#pragma once
#include "../../include/stabletts_hip.h"
#include "launch.h"

#include <map>
#include <string>
#include <vector>

namespace sthost {

enum ProfClass {
    PC_PREP = 0, PC_PRENET, PC_INPROJ, PC_FILM_LN1, PC_QKV, PC_ATTN, PC_OPROJ, PC_LN2, PC_FFN1, PC_FFN2,
    PC_LSC, PC_FINAL, PC_ODE, PC_TRAIN_FWD, PC_TRAIN_BWD, PC_COUNT
};

struct Param {
    std::vector<int64_t> shape;
    float* dev = nullptr;
    bool loaded = false;
    bool borrowed = false;     // st_bind_param: dev is the caller's tensor, never freed here
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Conv {            // packed 16-bit weights [cout][taps][cin] + fp32 bias
    void* w = nullptr;
    float* bias = nullptr;
    int cout = 0, cin = 0, taps = 0;
    bool split = false;  // cin = 3 x the reference's: [W_hi | W_hi | W_lo] for a split-precision operand [x_hi | x_lo | x_hi]
};

struct Captured { void* dev = nullptr; int64_t n = 0; bool is16 = false; };

struct ProfEvent { int cls; hipEvent_t a, b; double flops; };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct TrainState;     // engine_train.cpp
struct VocosState;     // engine_vocos.cpp

}  // namespace sthost

constexpr int kMaxParts = 4;
constexpr int kSplitKMax = 16;
constexpr size_t kSplitKBytes = 32u << 20;

struct st_engine {
    st_config cfg{};
    int device = 0;
    int dt = st::DT_BF16;
    int M = 0, Mp = 0, C = 0, F = 0, H = 0, L = 0, K = 0, G = 0;
    int kind = 0;                       // 0: CFM decoder estimator, 1: TextEncoder (same DiT block kernels), 2: Vocos vocoder
    int n_vocab = 0;
    // parameter-name prefix of DiT block i: estimator.py:13,79 "blocks.i.block." / text_encoder.py:25 "encoder.i."
    std::string blk(int i) const {
        return kind == 0 ? "blocks." + std::to_string(i) + ".block." : "encoder." + std::to_string(i) + ".";
    }
    std::map<std::string, sthost::Param> params;
    bool finalized = false;
    // Re-pack job lists (launch.h: PackJob): the first pack runs its individual launches and records them; later re-packs of
    // the same parameter tensors are ONE launch per list.  Dropped whenever a parameter pointer may have changed.
    struct PackList { std::vector<st::PackJob> jobs; st::PackJob* dev = nullptr; unsigned nblocks = 0; bool ready = false; bool recording = false; };
    PackList pk_fwd, pk_T;
    bool packed_once = false;          // pack_all has allocated the packed-weight buffers (st_repack re-uses them)
    std::string err;
    int64_t weight_bytes = 0;

    // packed weights
    std::vector<sthost::Conv> pre;              // 3 prenet convs
    sthost::Conv inx, inc, fin;                 // in_proj x-part / cond-part, final_proj
    std::vector<sthost::Conv> lsc, qkv, oproj, ffn1, ffn2;
    std::vector<void*> oproj_frag;      // per block: the out-projection weight in fragment order (oproj_ws.hip)
    std::vector<void*> qkv_frag;        // per block: the q/k/v weight in fragment order (qkv_ws.hip); empty if unsupported
    std::vector<void*> ffn_stream;      // per block: conv_1 + conv_2 weights as the fused FFN kernel's stream (ffn_fused.h); empty if unsupported
    std::vector<void*> owned;           // device allocations to free

    float* rope_cos = nullptr; float* rope_sin = nullptr; int rope_T = 0;
    void* sink = nullptr;               // 64 KiB of scratch: store target of rows outside the tensor (qkv_ws.hip)
    void* zeros = nullptr;              // 256 zero bytes: halo source of the LDS-DMA conv path

    // workspace arena
    char* ws = nullptr; size_t ws_cap = 0;
    uint64_t ws_sig = 0;                // layout signature of what the arena last held (see arena_fresh)

    // debug / profile
    bool capture = false;
    std::map<std::string, sthost::Captured> caps;
    bool prof = false;
    uint64_t prof_mask = ~0ull;
    int prof_stride = 1;
    int64_t prof_seen[sthost::PC_COUNT] = {0};
    std::vector<sthost::ProfEvent> evs;
    std::vector<hipEvent_t> ev_pool;
    int64_t prof_launches[sthost::PC_COUNT] = {0};
    double prof_ms[sthost::PC_COUNT] = {0};
    double prof_flops[sthost::PC_COUNT] = {0};

    int64_t last_nfe = 0, last_steps = 0, last_rejects = 0;   // statistics of the last solve
    // Non-finite guard: the boundary kernel that writes a call's output sets *status_host (host-mapped, no synchronisation) when a
    // value is NaN / Inf -- an f16 operand beyond 65504, a bad input.  st_output_status reads it after a stream sync; the next call
    // reads it without one and, if set, re-zeroes the arena (ragged tile skipping leaves stale frames that are only ever read by
    // don't-care positions, but 0 x NaN would leak: arena_fresh).
    int* status_host = nullptr; int* status_dev = nullptr;
    bool arena_poisoned() { if (status_host && *status_host) { *status_host = 0; ws_sig = 0; return true; } return false; }

    // tile policy: 256x256 tiles only when the launch has at least this many of them (~3/4 block per CU); read once
    // from ST_BIG_MIN_BLOCKS at st_create (tests force either tile family with it)
    void* adams_buf = nullptr; size_t adams_bytes = 0;     // extra state buffers of the implicit Adams solver (allocated at its first use)
    int big_min_blocks = 192;
    unsigned skip_mask = 0;             // developer tool, -DST_DEVTOOLS builds only (ST_SKIP_CLASSES, bit = profile class): launches of these classes of run_estimator are NOT issued -- what a
                                        // class costs the solve with its parts overlapping on four streams (tools/ab_engines.py); results are garbage
    int qkv_ws_min_tiles = 400;         // ... when the launch has at least this many 64-frame tiles (>= 5 per persistent block)
    int qkv_ws = 1;                     // fused q/k/v projection of big grids as the weight-stationary persistent kernel (qkv_ws.hip; default);
                                        // ST_QKV_WS=0: the generic conv tile
    int oproj_ws = 1;                   // out projection of big grids as the weight-stationary persistent kernel (oproj_ws.hip; default); ST_OPROJ_WS=0: the generic 256 x 256 tile
    int oproj_ws_min_tiles = 1000;      // ... from this many 32-frame tiles per launch
    int qkv_rc1 = 1;                    // fused q/k/v projection of big grids on 256 x 128 tiles with one weight buffer, two blocks per CU (ST_QKV_RC1=0: A/B)
    int ragged_skip = 1;                // conv / attention launches skip frame tiles past an utterance's last needed frame (ST_RAGGED_SKIP=0: A/B)
    int fused_ffn = -1;                 // FFN of big grids as ONE kernel, the intermediate kept in LDS.  -1 = default = 1: the direct kernel on 32x32x16 MFMA
                                        // (ffn_fused.h, bit-identical to the two-kernel path); ST_FUSED_FFN=3 (opt-in, f16 only): Winograd F(2,3) along the frames
                                        // (ffn_wino.h: a third fewer MFMAs, -1.45 % per solve, NOT bit-identical -- rounded sums as operands, +20 % error, half the
                                        // f16 range for the intermediate); ST_FUSED_FFN=0: two kernels.  Read at st_create.
    int phased = 1;                     // k = 3 convs on 256-wide tiles use the phased K loop (conv_gemm_phased.h); ST_PHASED=0: A/B runs
    int splitk_max = kSplitKMax, splitk_min_stages = 4;
    int small_tiles = 256;              // conv launches of <= this many 128x128 tiles use the 64-frame tile variants (0: never)
    int splitk_target = 256;            // split-K: blocks a small conv launch is brought up to (gemm())
    int attn_split = 0;                 // st_set_option("attention_precision", 1): q and k as hi + lo operand pairs, scores from three products (attention.hip: SPLIT)
    unsigned* lse_cells = nullptr;      // kLseCells cells 64 B apart: max over attention rows of the log2-sum-exp since the last st_attention_stats (order-preserving int bits)
    int attn_small_blocks = 32;         // attention launches of <= this many 256-query blocks use the key-split kernel
    float* kpart = nullptr;             // split-K partial planes [ks][items][T][256] fp32
    size_t kpart_bytes = 0;
    int conc = 1;                       // solve parts in flight on separate streams (their launches share the chip)
    hipStream_t sx[kMaxParts] = {};     // streams of solve parts 1.. (part 0 runs on the caller's stream)
    hipEvent_t ev_fork = nullptr, ev_joinx[kMaxParts] = {};

    // HIP-graph replay of the fixed-grid solve body (ST_HIP_GRAPH=1): one instantiated graph per solve signature
    struct SolveGraph {
        int B, T, n_steps, solver, use_cfg; float cfg_strength; const char* ws; int parts; int seen; hipGraphExec_t exec;
    };
    std::vector<SolveGraph> graphs;
    hipStream_t gstream = nullptr;      // capture stream
    void drop_graphs() {
        for (auto& g : graphs) if (g.exec) hipGraphExecDestroy(g.exec);
        graphs.clear();
    }

    // training (engine_train.cpp): transposed dgrad weights, saved activations, gradient buffers
    sthost::TrainState* train = nullptr;
    sthost::VocosState* voc = nullptr;  // kind == 2 (engine_vocos.cpp)

    int fail(int code, const std::string& msg) { err = msg; return code; }
};


namespace sthost {

#define HIPCHK(e, call)                                                                         \
    do {                                                                                        \
        hipError_t _err = (call);                                                               \
        if (_err != hipSuccess)                                                                 \
            return (e)->fail(ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_err));  \
    } while (0)

int dev_alloc(st_engine* e, void** p, size_t bytes);
const float* P(st_engine* e, const std::string& name);
hipError_t gemm(st_engine* e, int taps, int epi, const st::ConvGemmArgs& a, hipStream_t s);
bool gemm_is_phased(const st_engine* e, int taps, const st::ConvGemmArgs& a);   // gemm() would run the 256 x 254 phased kernel (the tile that has EPI_SILU)
void prof_collect(st_engine* e);
void capture(st_engine* e, const std::string& name, const void* dev, int64_t n, bool is16, hipStream_t s);
int ensure_ws(st_engine* e, size_t bytes);
// Ragged batches leave frame tiles past an utterance's end uncomputed; whatever the arena holds there is only ever read by
// don't-care positions (masked keys, frames whose results are multiplied by the mask), but it must be FINITE: 0 x NaN would
// leak.  Stale values of the SAME layout are real activations of an earlier call (finite); when the layout changes (other B, T,
// CFG, part count, entry point) the bytes would be re-interpreted under another type, so the used range is zeroed once, on `s`.
int arena_fresh(st_engine* e, uint64_t sig, size_t used_bytes, hipStream_t s);
int ensure_rope(st_engine* e, int T, hipStream_t s);
int check_ready(st_engine* e, int B, int T);
extern std::string g_create_error;
int vocos_finalize(st_engine* e);
int pack_all(st_engine* e, hipStream_t s);
// recorder / replayer of a PackList: begin -> the pk_* calls (individual launch + record) -> end (upload); replay = one launch
bool pk_replay(st_engine* e, st_engine::PackList& L, hipStream_t s, int* rc);      // true: the list was replayed (or failed: *rc)
void pk_begin(st_engine::PackList& L);
int pk_end(st_engine* e, st_engine::PackList& L, hipStream_t s);
void pk_drop(st_engine* e);                                                          // parameter pointers may have changed
int pk_weight(st_engine* e, st_engine::PackList& L, const float* src, int cout, int cin_total, int K, int ci_off, int ci_cnt, void* dst,
              int row_off, int cin_p, int col_off, int slice_w, int lo, hipStream_t s);
int pk_weight_t(st_engine* e, st_engine::PackList& L, const float* src, int cout, int cin_total, int taps, int ci_off, int ci_cnt, void* dst,
                int cin_p, int ld, int col_off, hipStream_t s);
int pk_copy(st_engine* e, st_engine::PackList& L, float* dst, const float* src, int n, hipStream_t s);            // (re)packs every 16-bit weight; allocates on the first call only
void vocos_destroy(st_engine* e);

// HIP-event bracket around the launches of one kernel class (st_profile_*)
struct ProfScope {
    st_engine* e; hipStream_t s; int idx = -1;
    ProfScope(st_engine* e_, hipStream_t s_, int cls, double flops);
    ~ProfScope();
};

// engine_train.cpp
int train_prepare(st_engine* e, hipStream_t s);       // packs the transposed (dgrad) weights if the parameters changed
void train_invalidate(st_engine* e);                  // called by st_finalize
void train_destroy(st_engine* e);
int64_t train_bytes(const st_engine* e);              // device bytes held by the training state
int64_t train_grad_layout(const st_engine* e, std::map<std::string, int64_t>* offs);   // flat parameter-gradient layout (64-byte aligned slices); returns the total

}  // namespace sthost
