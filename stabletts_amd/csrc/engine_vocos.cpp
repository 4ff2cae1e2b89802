// Vocos vocoder behind the C ABI (st_create_vocoder / st_vocos_forward): parameter table, weight packing and the
// launch sequence.  Reference: vocoders/vocos/models/model.py:11-20, backbone.py:21-56, module.py:16-46,
// head.py:17-117; config.py:4-19,46-50.  The five GEMM shapes (embed as an im2col GEMM, pwconv1 + GELU, pwconv2 +
// layer scale + residual, head) run on the implicit-GEMM kernels of the decoder over the FLATTENED rows of the batch
// (taps = 1: rows are independent); everything between them is in vocos_kernels.hip.
#include "engine_internal.h"
#include "vocos_launch.h"

#include <cstring>
#include <string>

using namespace st;
using namespace sthost;

namespace sthost {

struct VocosState {
    st_vocos_config cfg{};
    Conv embed, head;
    std::vector<Conv> pw1, pw2;
};

static std::string vblk(int i) { return "backbone.convnext." + std::to_string(i) + "."; }

void vocos_build_params(st_engine* e, const st_vocos_config& c) {
    auto expect = [&](const std::string& n, std::vector<int64_t> shape) { Param p; p.shape = std::move(shape); e->params[n] = p; };
    const int64_t C = c.dim, F = c.intermediate_dim, M = c.input_channels;
    expect("backbone.embed.weight", {C, M, 7}); expect("backbone.embed.bias", {C});                 // backbone.py:28
    expect("backbone.norm.weight", {C}); expect("backbone.norm.bias", {C});                        // :29
    for (int i = 0; i < c.num_layers; ++i) {                                                       // module.py:22-31
        const std::string p = vblk(i);
        expect(p + "dwconv.weight", {C, 1, 7}); expect(p + "dwconv.bias", {C});
        expect(p + "norm.weight", {C}); expect(p + "norm.bias", {C});
        expect(p + "pwconv1.weight", {F, C}); expect(p + "pwconv1.bias", {F});
        expect(p + "pwconv2.weight", {C, F}); expect(p + "pwconv2.bias", {C});
        expect(p + "gamma", {C});
    }
    expect("backbone.final_layer_norm.weight", {C}); expect("backbone.final_layer_norm.bias", {C});   // backbone.py:41
    expect("head.out.weight", {c.n_fft + 2, C}); expect("head.out.bias", {c.n_fft + 2});            // head.py:88-89
    expect("head.istft.window", {c.n_fft});                                                         // head.py:27-28
}

// st_finalize of a vocoder handle: 16-bit GEMM weights.  Linear weights (out, in) are k = 1 convolutions; the k = 7
// embed convolution packs as [cout][tap][cin] = the K order of launch_voc_im2col7's rows.  The head's 2050 output
// rows are laid out as two planes of kVocHeadPlane rows (log-magnitude 0..1024, phase 0..1024, zero rows between).
int vocos_finalize(st_engine* e) {
    VocosState* v = e->voc;
    const st_vocos_config& c = v->cfg;
    const int C = c.dim, F = c.intermediate_dim, M = c.input_channels, L = c.num_layers;
    hipStream_t s = nullptr;
    auto pack = [&](Conv& cv, const std::string& wname, const std::string& bname, int cout, int cin, int taps) -> int {
        cv.cout = cout; cv.cin = cin; cv.taps = taps; cv.split = false;
        int rc = dev_alloc(e, &cv.w, (size_t)cout * taps * cin * 2); if (rc) return rc;
        HIPCHK(e, launch_pack_weight(e->dt, P(e, wname), cout, cin, taps, 0, cin, cv.w, 0, cin, 0, cin, 0, s));
        rc = dev_alloc(e, (void**)&cv.bias, (size_t)cout * 4); if (rc) return rc;
        HIPCHK(e, hipMemcpyAsync(cv.bias, P(e, bname), (size_t)cout * 4, hipMemcpyDeviceToDevice, s));
        return ST_OK;
    };
    // the pointwise convs of the ConvNeXt blocks with SPLIT WEIGHTS (round 6): K = [x | x] against [W_hi | W_lo] -- the activation operand is read
    // twice, no producer changes; twice the MFMA work of the backbone's GEMMs.  Takes the weights' share out of the waveform's 16-bit error
    // (it sat 3 % under its 1e-3 gate).
    auto pack_wsplit = [&](Conv& cv, const std::string& wname, const std::string& bname, int cout, int cin) -> int {
        cv.cout = cout; cv.cin = 2 * cin; cv.taps = 1; cv.split = false;
        int rc = dev_alloc(e, &cv.w, (size_t)cout * 2 * cin * 2); if (rc) return rc;
        for (int k2 = 0; k2 < 2; ++k2)
            HIPCHK(e, launch_pack_weight(e->dt, P(e, wname), cout, cin, 1, 0, cin, cv.w, 0, 2 * cin, k2 * cin, cin, k2 == 1, s));
        rc = dev_alloc(e, (void**)&cv.bias, (size_t)cout * 4); if (rc) return rc;
        HIPCHK(e, hipMemcpyAsync(cv.bias, P(e, bname), (size_t)cout * 4, hipMemcpyDeviceToDevice, s));
        return ST_OK;
    };
    int rc;
    if ((rc = pack(v->embed, "backbone.embed.weight", "backbone.embed.bias", C, M, 7))) return rc;
    v->embed.cin = 7 * M; v->embed.taps = 1;        // consumed as a k = 1 GEMM over the im2col rows
    v->pw1.assign(L, Conv()); v->pw2.assign(L, Conv());
    for (int i = 0; i < L; ++i) {
        if ((rc = pack_wsplit(v->pw1[i], vblk(i) + "pwconv1.weight", vblk(i) + "pwconv1.bias", F, C))) return rc;
        if ((rc = pack_wsplit(v->pw2[i], vblk(i) + "pwconv2.weight", vblk(i) + "pwconv2.bias", C, F))) return rc;
    }
    {
        Conv& h = v->head;
        const int bins = c.n_fft / 2 + 1, planes = 2 * kVocHeadPlane;
        // split-precision operands (round 5), as for the decoder's in_proj / final_proj: the head's output x feeds exp(x) -- an absolute
        // error of x is a RELATIVE error of the magnitude -- so the 16-bit rounding of its operands reached the waveform un-attenuated
        // (f16: hidden 4.8e-4 -> audio 1.1e-3).  K = [h_hi | h_lo | h_hi] against [W_hi | W_hi | W_lo]: 3x the MFMA work of one small GEMM.
        h.cout = planes; h.cin = 3 * C; h.taps = 1; h.split = true;
        if ((rc = dev_alloc(e, &h.w, (size_t)planes * 3 * C * 2))) return rc;
        if ((rc = dev_alloc(e, (void**)&h.bias, (size_t)planes * 4))) return rc;
        HIPCHK(e, hipMemsetAsync(h.w, 0, (size_t)planes * 3 * C * 2, s));
        HIPCHK(e, hipMemsetAsync(h.bias, 0, (size_t)planes * 4, s));
        const float* W = P(e, "head.out.weight"); const float* Bv = P(e, "head.out.bias");
        for (int part = 0; part < 2; ++part) {       // head.py:104: mag, p = x.chunk(2, dim=1)
            for (int k3 = 0; k3 < 3; ++k3)
                HIPCHK(e, launch_pack_weight(e->dt, W + (size_t)part * bins * C, bins, C, 1, 0, C, h.w, part * kVocHeadPlane, 3 * C, k3 * C, C, k3 == 2, s));
            HIPCHK(e, hipMemcpyAsync(h.bias + part * kVocHeadPlane, Bv + (size_t)part * bins, (size_t)bins * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    HIPCHK(e, hipDeviceSynchronize());
    e->finalized = true;
    return ST_OK;
}

void vocos_destroy(st_engine* e) { delete e->voc; e->voc = nullptr; }

}  // namespace sthost

extern "C" {

int st_create_vocoder(const st_vocos_config* cfg, int device, st_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return ST_ERR_INVALID; }
    auto bad = [&](const char* m, int code) { g_create_error = m; return code; };
    if (cfg->num_layers < 1 || cfg->num_layers > 32) return bad("num_layers must be in [1, 32]", ST_ERR_INVALID);
    if (cfg->input_channels < 1 || cfg->dim < 1 || cfg->intermediate_dim < 1) return bad("channel counts must be positive", ST_ERR_INVALID);
    if (cfg->operand_dtype != ST_OPERAND_BF16 && cfg->operand_dtype != ST_OPERAND_F16) return bad("operand_dtype", ST_ERR_INVALID);
    // limits of this native build
    if (cfg->dim != kVocDim) return bad("native vocoder kernels are built for dim == 512", ST_ERR_UNSUPPORTED);
    if (cfg->n_fft != kVocNfft || cfg->hop_length != kVocHop) return bad("native ISTFT is built for n_fft == 2048, hop_length == 512", ST_ERR_UNSUPPORTED);
    if (cfg->input_channels % 64 != 0 || cfg->input_channels > 192) return bad("input_channels must be 64, 128 or 192", ST_ERR_UNSUPPORTED);
    if (cfg->intermediate_dim % 256 != 0) return bad("intermediate_dim must be a multiple of 256", ST_ERR_UNSUPPORTED);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return bad("no such HIP device", ST_ERR_HIP);
    if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice failed", ST_ERR_HIP);
    st_engine* e = new st_engine();
    e->device = device; e->kind = 2;
    e->dt = cfg->operand_dtype == ST_OPERAND_BF16 ? DT_BF16 : DT_F16;
    e->voc = new VocosState();
    e->voc->cfg = *cfg;
    e->splitk_target = 0; e->attn_small_blocks = 0;      // decoder-only small-grid paths
    vocos_build_params(e, *cfg);
    if (hipMalloc(&e->zeros, 256) != hipSuccess || hipMemset(e->zeros, 0, 256) != hipSuccess) {
        vocos_destroy(e); delete e;
        return bad("hipMalloc failed", ST_ERR_HIP);
    }
    *out = e;
    return ST_OK;
}

int st_vocos_forward(st_engine* e, const float* mel, float* audio, int B, int T, void* stream) {
    if (!e) return ST_ERR_INVALID;
    if (e->kind != 2) return e->fail(ST_ERR_STATE, "this handle is not a vocoder (st_create_vocoder)");
    if (!e->finalized) return e->fail(ST_ERR_STATE, "st_finalize() has not been called after loading parameters");
    if (!mel || !audio) return e->fail(ST_ERR_INVALID, "null tensor pointer");
    if (B < 1 || T < 1) return e->fail(ST_ERR_INVALID, "B and T must be >= 1");
    VocosState* v = e->voc;
    const st_vocos_config& c = v->cfg;
    const int C = c.dim, F = c.intermediate_dim, M = c.input_channels, L = c.num_layers;
    const int64_t R = (int64_t)B * T;
    if (R * 2 * kVocHeadPlane >= ((int64_t)1 << 31)) return e->fail(ST_ERR_INVALID, "B*T too large for 32-bit row indexing");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;

    // workspace: im2col rows, fp32 residual stream, 16-bit operands, head output, windowed frames
    size_t off = 0;
    auto want = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_a16 = want((size_t)R * 7 * M * 2), o_x = want((size_t)R * C * 4), o_h16 = want((size_t)R * C * 2), o_h16lo = want((size_t)R * C * 2);
    const size_t o_u16 = want((size_t)R * F * 2), o_head = want((size_t)R * 2 * kVocHeadPlane * 4), o_fr = want((size_t)R * kVocNfft * 4);
    int rc = ensure_ws(e, off); if (rc) return rc;
    void* a16 = e->ws + o_a16; float* x = (float*)(e->ws + o_x); void* h16 = e->ws + o_h16; void* h16lo = e->ws + o_h16lo; void* u16 = e->ws + o_u16;
    float* head = (float*)(e->ws + o_head); float* frames = (float*)(e->ws + o_fr);

    auto args = [&](const Conv& cv) {
        ConvGemmArgs a; memset(&a, 0, sizeof(a));
        a.w = cv.w; a.bias = cv.bias; a.cout = cv.cout; a.T = (int)R; a.n_items = 1;     // flattened rows
        a.a0_mod = 1; a.a1_mod = 1; a.mask_mod = 1; a.zeros = e->zeros;
        return a;
    };
    const bool cap = e->capture;
    {   // embed (backbone.py:51) + LayerNorm (:52)
        ProfScope ps(e, s, PC_PRENET, 2.0 * R * C * 7.0 * M);
        HIPCHK(e, launch_voc_im2col7(e->dt, mel, B, M, T, a16, s));
        ConvGemmArgs a = args(v->embed); a.a0 = a16; a.c0 = 7 * M; a.out32 = x;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
        HIPCHK(e, launch_voc_ln(e->dt, x, P(e, "backbone.norm.weight"), P(e, "backbone.norm.bias"), R, x, nullptr, nullptr, s));
    }
    if (cap) capture(e, "voc.embed", x, R * C, false, s);
    for (int i = 0; i < L; ++i) {       // ConvNeXtBlock.forward (module.py:33-46)
        const std::string p = vblk(i);
        {
            ProfScope ps(e, s, PC_FILM_LN1, 0);
            HIPCHK(e, launch_voc_dwconv_ln(e->dt, x, P(e, p + "dwconv.weight"), P(e, p + "dwconv.bias"), P(e, p + "norm.weight"),
                                           P(e, p + "norm.bias"), B, T, h16, s));
        }
        {
            ProfScope ps(e, s, PC_FFN1, 2.0 * R * C * (double)F);
            ConvGemmArgs a = args(v->pw1[i]); a.a0 = h16; a.c0 = C; a.a1 = h16; a.c1 = C; a.out16 = u16;      // [x | x] . [W_hi | W_lo]
            HIPCHK(e, gemm(e, 1, EPI_GELU16, a, s));
        }
        {   // pwconv2, layer scale, residual (:40-45): x += gamma * (W u + b)
            ProfScope ps(e, s, PC_FFN2, 2.0 * R * C * (double)F);
            ConvGemmArgs a = args(v->pw2[i]); a.a0 = u16; a.c0 = F; a.a1 = u16; a.c1 = F; a.out32 = x; a.gate = P(e, p + "gamma"); a.gate_stride = 0;
            HIPCHK(e, gemm(e, 1, EPI_RESGATE, a, s));
        }
        if (cap) capture(e, "voc.block" + std::to_string(i), x, R * C, false, s);
    }
    {   // final LayerNorm (backbone.py:55) + head projection (head.py:103)
        ProfScope ps(e, s, PC_FINAL, 2.0 * R * C * (double)(c.n_fft + 2));
        HIPCHK(e, launch_voc_ln(e->dt, x, P(e, "backbone.final_layer_norm.weight"), P(e, "backbone.final_layer_norm.bias"), R,
                                cap ? x : nullptr, h16, h16lo, s));
        ConvGemmArgs a = args(v->head); a.a0 = h16; a.c0 = C; a.a1 = h16lo; a.c1 = C; a.c2 = C; a.out32 = head;
        HIPCHK(e, gemm(e, 1, EPI_F32, a, s));
    }
    if (cap) { capture(e, "voc.hidden", x, R * C, false, s); capture(e, "voc.head_out", head, R * 2 * kVocHeadPlane, false, s); }
    {   // ISTFT (head.py:104-116)
        ProfScope ps(e, s, PC_ODE, 0);
        HIPCHK(e, launch_voc_spec_ifft(head, P(e, "head.istft.window"), R, frames, s));
        HIPCHK(e, launch_voc_overlap_add(frames, P(e, "head.istft.window"), B, T, audio, s));
    }
    return ST_OK;
}

}  // extern "C"
