#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 16x16x32 (114 per phase), SHAPE 1: 32x32x16 (57 per phase: same flops)
// FILL 0: none, 1: 130 v_fma, 2: 130 v_cvt_pk_bf16 + shifts (the split), 3: 38 ds_read_b128, 4: interleaved fma (1 per MFMA, same wave)
template <int SHAPE, int FILL>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    f32x4 acc[16]; f32x16 acc2[4]; float v[16];
    for (int i = 0; i < 16; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = threadIdx.x * 1e-3f + i; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc2[i][j] = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 0.5f); }
    f32x4 ld = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (FILL == 4) {
#pragma unroll
            for (int i = 0; i < 114; ++i) {
                if (SHAPE == 0) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 15], 0, 0, 0);
                else if (i & 1) acc2[(i >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[(i >> 1) & 3], 0, 0, 0);
                v[i & 15] = __builtin_fmaf(v[i & 15], c, 1.0f);
            }
        } else {
            if (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 114; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 15], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 57; ++i) acc2[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[i & 3], 0, 0, 0);
            }
            if (FILL == 1) {
#pragma unroll
                for (int i = 0; i < 130; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], c, 1.0f);
            }
            if (FILL == 2) {
#pragma unroll
                for (int i = 0; i < 65; ++i) {
                    const __bf16 h = (__bf16)v[i & 15];
                    v[i & 15] = v[i & 15] - (float)h + c;
                }
            }
            if (FILL == 3) {
#pragma unroll
                for (int i = 0; i < 38; ++i) ld += *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + i * 256) & 8188));
            }
        }
        __syncthreads();
    }
    float s = ld[0] + ld[1] + ld[2] + ld[3];
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + acc2[i & 3][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, int FILL> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<SHAPE, FILL><<<256, 512>>>(out, 10, 1.0001f);
    hipEventRecord(e0); k<SHAPE, FILL><<<256, 512>>>(out, iters, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f ns/iter\n", name, ms * 1e6 / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0>("16x16x32 x114", out);
    run<0, 1>("16x16x32 x114 + 130 fma after", out);
    run<0, 4>("16x16x32 x114 + 114 fma interleaved", out);
    run<0, 2>("16x16x32 x114 + 65 bf16 split (cvt,sub)", out);
    run<0, 3>("16x16x32 x114 + 38 ds_read_b128", out);
    run<1, 0>("32x32x16 x57", out);
    run<1, 1>("32x32x16 x57 + 130 fma after", out);
    run<1, 4>("32x32x16 x57 + 114 fma interleaved", out);
    run<1, 2>("32x32x16 x57 + 65 bf16 split", out);
    run<1, 3>("32x32x16 x57 + 38 ds_read_b128", out);
    return 0;
}
