// micro-benchmark: does a SIMD overlap one wave's MFMAs with its partner wave's VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM>
__device__ __forceinline__ void mfma_phase(f32x4 (&acc)[16], const bf16x8& a, const bf16x8& b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 15], 0, 0, 0);
}
template <int NV>
__device__ __forceinline__ void valu_phase(float (&v)[16], float c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], c, 1.0f);
}

// MODE 0: every wave does [MFMA x NM][VALU x NV] per iteration (lockstep through the barrier)
// MODE 1: waves 0-3 do 2 x MFMA phase, waves 4-7 do 2 x VALU phase (same total work)
// MODE 2: MODE 0 without the barrier
// MODE 3: MFMA only (all waves), MODE 4: VALU only (all waves)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c) {
    f32x4 acc[16]; float v[16];
    for (int i = 0; i < 16; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = threadIdx.x * 1e-3f + i; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 0.5f); }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) { mfma_phase<114>(acc, a, b); valu_phase<130>(v, c); }
        if (MODE == 1) { if (wave < 4) { mfma_phase<114>(acc, a, b); mfma_phase<114>(acc, a, b); } else { valu_phase<130>(v, c); valu_phase<130>(v, c); } }
        if (MODE == 3) mfma_phase<114>(acc, a, b);
        if (MODE == 4) valu_phase<130>(v, c);
        if (MODE != 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<256, 512>>>(out, 10, 1.0001f);
    hipEventRecord(e0); k<MODE><<<256, 512>>>(out, iters, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f ns/iter\n", name, ms * 1e6 / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<3>("MFMA x114 only (8 waves)", out);
    run<4>("VALU x130 only (8 waves)", out);
    run<0>("both, every wave, barrier (lockstep)", out);
    run<2>("both, every wave, no barrier", out);
    run<1>("specialised: 4 MFMA waves + 4 VALU waves", out);
    return 0;
}
