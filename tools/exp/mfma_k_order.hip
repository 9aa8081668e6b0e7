// Experiment: is the fp32 result of two chained v_mfma_f32_32x32x16_f16 (K = 32) independent of WHERE in k a block of nonzero
// products sits, when every other product is an exact zero?  (Needed for titles placed at different row offsets of a 32-row
// tile to produce bit-identical attention sums.)   hipcc --offload-arch=gfx950 -O2 -o mfma_k_order mfma_k_order.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D[i][j] = sum_k A[i][k] B[k][j], K = 32 as two MFMAs; A row i from a[i][32], B column j from b[j][32]
__global__ void kern(const _Float16* a, const _Float16* b, float* d) {
    const int lane = threadIdx.x, l32 = lane & 31, hh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < 2; ++s) {
        h8 af, bf;
        for (int e = 0; e < 8; ++e) { af[e] = a[l32 * 32 + 16 * s + 8 * hh + e]; bf[e] = b[l32 * 32 + 16 * s + 8 * hh + e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) d[(8 * (r >> 2) + 4 * hh + (r & 3)) * 32 + l32] = acc[r];
}

int main() {
    const int R = 12;                     // nonzero products per dot product
    _Float16 ha[32 * 32], hb[32 * 32];
    float ref[32 * 32], out[32 * 32];
    _Float16 *da, *db; float* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof out);
    srand(1);
    float va[32][R], vb[32][R];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < R; ++k) { va[i][k] = (rand() / (float)RAND_MAX - 0.5f); vb[i][k] = (rand() / (float)RAND_MAX - 0.5f) * 3.f; }
    for (int off = 0; off <= 32 - R; ++off) {
        memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < R; ++k) { ha[i * 32 + off + k] = (_Float16)va[i][k]; hb[i * 32 + off + k] = (_Float16)vb[i][k]; }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(out, dd, sizeof out, hipMemcpyDeviceToHost);
        if (off == 0) memcpy(ref, out, sizeof out);
        int diff = 0; float worst = 0.f;
        for (int i = 0; i < 1024; ++i) if (memcmp(&out[i], &ref[i], 4)) { ++diff; float e = out[i] - ref[i]; if (e < 0) e = -e; if (e > worst) worst = e; }
        printf("offset %2d: %4d of 1024 results differ from offset 0 (max |diff| %.3g)\n", off, diff, worst);
    }
    return 0;
}
