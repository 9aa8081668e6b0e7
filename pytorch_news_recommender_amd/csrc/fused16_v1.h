// Geometry shared by the forward (fused16_v1.hip) and the backward (fused16_v1_bwd.hip) of nrms_v1's news encoder in fp16 mode.
#pragma once
#include "fused16.h"

namespace nrms {

struct V1Geom {
    int d, h, dk, hv, dkv;      // model width, attention heads, d_k; output blocks ("virtual heads") and their width
    int kl, lo;                 // k-step of the leftover features, leftover features per head (d_k - 48, >= 0)
    int n_head_tiles;           // 6 h
};

__host__ __device__ inline V1Geom v1_geom(int d, int h) {
    V1Geom g;
    g.d = d; g.h = h; g.dk = d / h;
    g.hv = 10; g.dkv = d / 10;
    g.kl = 3 * h;
    g.lo = g.dk > 48 ? g.dk - 48 : 0;
    g.n_head_tiles = 6 * h;
    return g;
}

// column (0 .. 319) of attn feature fh of head hd in the stored / operand order of attn16
__host__ __device__ inline int v1_attn_col(const V1Geom& g, int hd, int fh) {
    if (fh < 48) {
        const int t = fh & 15;
        const int p = 8 * ((t >> 2) & 1) + (((t >> 3) << 2) | (t & 3));         // P16 order inside a k-step (acc_frag's order)
        return 16 * (3 * hd + (fh >> 4)) + p;
    }
    return 16 * g.kl + g.lo * hd + (fh - 48);
}
constexpr int V1_ONES_COL = 16 * 19;        // k-step 19, position 0: the ones column (bias of W_O)


}  // namespace nrms
