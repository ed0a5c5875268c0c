// Which instruction forms reach the ~2.3-cycle issue rate?  (cycles per wave64 VALU instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, int iters, float cs)
{
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f; }
    const float cv = out[threadIdx.x & 1];       // VGPR coefficient
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {          // 8 independent muls, SGPR coefficient (kernel argument)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] * cs;
#pragma unroll
                for (int i = 0; i < 8; ++i) b[i] = b[i] * cs;
            } else if (MODE == 1) {   // same with a VGPR coefficient
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] * cv;
#pragma unroll
                for (int i = 0; i < 8; ++i) b[i] = b[i] * cv;
            } else if (MODE == 2) {   // the kernel's pattern, element-interleaved: sub, mul(SGPR), sub x 4 chains + 4 more
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = a[i + 4] - a[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = b[i] * cs;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = a[i] - b[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i + 4] = a[(i + 1) & 3] - a[i];
            } else if (MODE == 3) {   // same with VGPR coefficient
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = a[i + 4] - a[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = b[i] * cv;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = a[i] - b[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i + 4] = a[(i + 1) & 3] - a[i];
            } else if (MODE == 4) {   // chain-serial order as hipcc's default scheduler emits it (sub,mul,sub per element)
#pragma unroll
                for (int i = 0; i < 4; ++i) { float t = a[i + 4] - a[i]; t = t * cs; a[i] = a[i] - t; b[i + 4] = a[(i + 1) & 3] - a[i]; }
                asm volatile("" ::: "memory");
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    if (s == 123.456f) out[1] = s;
}
template <int MODE> void run(const char *name, float *d)
{
    const int iters = 4000;
    for (int W : {1, 3, 8}) {
        const int blocks = 1024 * W;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0000001f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 4 * 16 * W;
        printf("%-44s W=%d  %.2f cycles/instr at 2.4 GHz\n", name, W, 2.4 * ms * 1e6 / instr);
    }
}
int main()
{
    float *d; (void)hipMalloc(&d, 64); float h[2] = {1.0000001f, 1.0000001f}; (void)hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    run<0>("16 indep v_mul, SGPR coefficient", d);
    run<1>("16 indep v_mul, VGPR coefficient", d);
    run<2>("stencil pattern interleaved, SGPR coef", d);
    run<3>("stencil pattern interleaved, VGPR coef", d);
    run<4>("stencil pattern chain-serial, SGPR coef", d);
    return 0;
}
