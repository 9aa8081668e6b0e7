#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the real NT stage shape: 19 tiles; per tile 2 ds_read_b128 (hi, lo fragments, used as B operands) then 6 MFMAs.
// SHAPE 0: 16x16x32 (6 per tile)   SHAPE 1: 32x32x16 (3 per tile, same flops)
// READS: ds_read_b128 per tile (0, 2);  PREF: fragments of tile t+1 are read before the MFMAs of tile t
template <int SHAPE, int READS, int PREF>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 19 * 16 * 32 + 64];
    f32x4 acc[2][19]; f32x16 acc2[10];
    for (int i = 0; i < 19; ++i) { acc[0][i] = f32x4{0, 0, 0, 0}; acc[1][i] = acc[0][i]; }
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) acc2[i][j] = 0.f;
    for (int i = threadIdx.x; i < 2 * 19 * 16 * 32; i += 512) lds[i] = (__bf16)(i * 1e-4f);
    __syncthreads();
    bf16x8 a0, a1;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)(threadIdx.x * 1e-3f); a1[i] = (__bf16)(i * 0.5f); }
    const int lane = threadIdx.x & 63;
    const __bf16* frag = lds + lane * 8;
    for (int it = 0; it < iters; ++it) {
        bf16x8 bh = a0, bl = a1, nh, nl;
        if (READS && PREF) { bh = *reinterpret_cast<const bf16x8*>(frag); bl = *reinterpret_cast<const bf16x8*>(frag + 19 * 512); }
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            if (READS && PREF && t + 1 < 19) { nh = *reinterpret_cast<const bf16x8*>(frag + (t + 1) * 512); nl = *reinterpret_cast<const bf16x8*>(frag + (19 + t + 1) * 512); }
            if (READS && !PREF) { bh = *reinterpret_cast<const bf16x8*>(frag + t * 512); bl = *reinterpret_cast<const bf16x8*>(frag + (19 + t) * 512); }
            if (SHAPE == 0) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bl, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bl, acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh, acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh, acc[1][t], 0, 0, 0);
            } else {
                acc2[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc2[t >> 1], 0, 0, 0);
                acc2[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc2[t >> 1], 0, 0, 0);
                acc2[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc2[t >> 1], 0, 0, 0);
            }
            if (READS && PREF) { bh = nh; bl = nl; }
        }
        __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 19; ++i) s += acc[0][i][0] + acc[1][i][1] + acc2[i >> 1][i & 15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, int READS, int PREF> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<SHAPE, READS, PREF><<<256, 512>>>(out, 10);
    hipEventRecord(e0); k<SHAPE, READS, PREF><<<256, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %8.1f ns/iter\n", name, ms * 1e6 / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0, 0>("16x16x32: 19 x 6 MFMA, operands in registers", out);
    run<0, 2, 0>("16x16x32: + 2 ds_read_b128 per tile, read-then-use", out);
    run<0, 2, 1>("16x16x32: + 2 ds_read_b128 per tile, one tile ahead", out);
    run<1, 0, 0>("32x32x16: 19 x 3 MFMA, operands in registers", out);
    run<1, 2, 0>("32x32x16: + 2 ds_read_b128 per tile, read-then-use", out);
    run<1, 2, 1>("32x32x16: + 2 ds_read_b128 per tile, one tile ahead", out);
    return 0;
}
