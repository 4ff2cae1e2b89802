// Host-side check of the weight-layout index maps of stabletts_amd/csrc/common.h (compiled by tests/test_pack_index.py with
// hipcc --cuda-host-only; no GPU): ffn_stream_index (both stages) and qkv_frag_index must be bijections onto
// their buffers, and every source element must be hit exactly once; ffn_wino_index likewise (three planes per tap triple).
#include <cstdio>
#include <vector>
#include "../../stabletts_amd/csrc/common.h"

int main() {
    int bad = 0;
    for (int F : {256, 512, 1024, 2048})
        {
            const size_t n = (size_t)F * 256 * 3;
            std::vector<unsigned char> dst(2 * n, 0);
            for (int stage = 0; stage < 2; ++stage) {
                std::vector<unsigned char> src(n, 0);
                for (size_t idx = 0; idx < n; ++idx) {
                    size_t so, dof;
                    st::ffn_stream_index(idx, stage, F, &so, &dof);
                    if (so >= n || dof >= 2 * n) { ++bad; continue; }
                    ++src[so]; ++dst[dof];
                }
                for (size_t i = 0; i < n; ++i) bad += src[i] != 1;
            }
            for (size_t i = 0; i < 2 * n; ++i) bad += dst[i] != 1;
        }
    {
        std::vector<unsigned char> dst(3 * 256 * 256, 0);
        for (int plane = 0; plane < 3; ++plane)
            for (int co = 0; co < 256; ++co)
                for (int ci = 0; ci < 256; ++ci) {
                    const size_t d = st::qkv_frag_index(plane, co, ci);
                    if (d >= dst.size()) { ++bad; continue; }
                    ++dst[d];
                    // a wave's fragment (plane, co / 32, ci / 16) is one contiguous KiB, lane-linear
                    const size_t frag = d >> 9, lane = (d >> 3) & 63, e = d & 7;
                    bad += frag != (size_t)((plane * 8 + co / 32) * 16 + ci / 16) || lane != (size_t)(((ci >> 3) & 1) * 32 + (co & 31)) || e != (size_t)(ci & 7);
                }
        for (unsigned char c : dst) bad += c != 1;
    }
    // ffn_wino_index (the Winograd fused FFN's stream): per stage every (row, input channel) tap triple is used for exactly the three
    // planes, the two stages fill the stream exactly once, a wave's three fragments of a k-step are consecutive KiB, lane-linear
    for (int F : {256, 512, 1024, 2048}) {
        const size_t n = (size_t)F * 256 * 3;
        std::vector<unsigned char> dst(2 * n, 0);
        for (int stage = 0; stage < 2; ++stage) {
            std::vector<unsigned char> planes(n / 3, 0);
            for (size_t idx = 0; idx < n; ++idx) {
                size_t so, dof; int pl;
                st::ffn_wino_index(idx, stage, F, &so, &dof, &pl);
                if (so % 3 || so + 2 >= n || dof >= 2 * n || pl < 0 || pl > 2) { ++bad; continue; }
                planes[so / 3] |= (unsigned char)(1 << pl); ++dst[dof];
                const size_t frag = dof >> 9;                       // = (((chunk * 2 + stage) * 16 + k-step) * 8 + wave) * 3 + plane
                bad += (int)(frag % 3) != pl || (int)((frag / 3 / 8 / 16) & 1) != stage;
            }
            for (unsigned char c : planes) bad += c != 7;
        }
        for (unsigned char c : dst) bad += c != 1;
        const float g3[3] = {1.0f, 2.0f, 4.0f};
        bad += st::ffn_wino_plane(g3, 0) != 1.0f || st::ffn_wino_plane(g3, 1) != 3.5f || st::ffn_wino_plane(g3, 2) != 4.0f;
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
