// VALU issue-rate microbenchmark for gfx950: independent v_add_f32 / v_mul_f32 / v_pk_add_f32 /
// DPP chains, W waves per SIMD.  Prints wave-instructions per cycle per SIMD at an assumed clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, int iters)
{
    float a[8];
    float2_ p[4];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) p[i] = float2_{a[2 * i], a[2 * i + 1]};
    const float c = out[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] + c;              // 8 independent v_add_f32
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] * c;              // v_mul_f32
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = p[i] + float2_{c, c};   // 4 independent v_pk_add_f32 (8 flops)
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i)                                 // DPP-fused subtract
                    a[i] = a[i] - __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[(i + 1) & 7]), 0x130, 0xF, 0xF, true));
            } else if (MODE == 4) {
                a[0] = a[0] + c; a[0] = a[0] * c; a[0] = a[0] + c; a[0] = a[0] * c;   // one dependent chain
                a[0] = a[0] + c; a[0] = a[0] * c; a[0] = a[0] + c; a[0] = a[0] * c;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    if (s == 123.456f) out[1] = s;
}
template <int MODE> void run(const char *name, float *d, int per)
{
    const int iters = 4000;
    for (int W : {1, 2, 3, 4, 8}) {
        const int blocks = 1024 * W;   // 256 CUs x 4 SIMDs x W
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 8 * per * W;   // per SIMD
        printf("%-22s W=%d  %.3f ms  %.3f wave-instr/ns/SIMD  (%.2f cycles/instr at 2.4 GHz)\n", name, W, ms,
               instr / (ms * 1e6), 2.4 * ms * 1e6 / instr);
    }
}
int main()
{
    float *d; hipMalloc(&d, 64); float h[2] = {1.0000001f, 0}; hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    run<0>("v_add_f32 x8 indep", d, 8);
    run<1>("v_mul_f32 x8 indep", d, 8);
    run<2>("v_pk_add_f32 x4 indep", d, 4);
    run<3>("v_sub_f32_dpp x8", d, 8);
    run<4>("dependent add/mul", d, 8);
    return 0;
}
