// How does an isolated DPP instruction slow a SIMD that several waves share?  Wall-clock ns per 64-instruction
// block per wave-slot (W waves per SIMD), for blocks that contain k DPP forms in different arrangements.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench_dpp tools/ubench_dpp.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define P "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n"
#define PL "v_add_f32 %7, %8, %7\n"
#define D "v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DA ".p2align 3\n" D   /* the DPP form on an 8-byte boundary (s_nop padding when needed) */
#define E64 "v_add_f32_e64 %7, %8, %7\n"   /* a plain add in the 8-byte VOP3 encoding */
#define G0 P PL        /* 8 plain */
#define G1 P D         /* 7 plain + 1 DPP */
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, float cs, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = cs + threadIdx.x * 1e-9f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(G0 G0 G0 G0 G0 G0 G0 G0 OPS);
        if (KIND == 1) asm volatile(G1 G0 G0 G0 G0 G0 G0 G0 OPS);                      // 1 per 64
        if (KIND == 2) asm volatile(G1 G0 G0 G0 G1 G0 G0 G0 OPS);                      // 1 per 32
        if (KIND == 3) asm volatile(G1 G0 G1 G0 G1 G0 G1 G0 OPS);                      // 1 per 16
        if (KIND == 4) asm volatile(G1 G1 G1 G1 G1 G1 G1 G1 OPS);                      // 1 per 8
        if (KIND == 5) asm volatile(P D D G0 G0 G0 G0 G0 G0 PL OPS);                   // 2 adjacent per 64
        if (KIND == 6) asm volatile(P D D D D G0 G0 G0 G0 G0 G0 PL PL PL OPS);         // 4 adjacent per 64
        if (KIND == 7) asm volatile(P D D D D G0 G0 G0 P D D D D G0 G0 PL PL PL PL PL PL P OPS);   // 2 x 4 adjacent per 64 (+-)
        if (KIND == 9) asm volatile(P DA G0 P DA G0 P DA G0 P DA G0 OPS);              // 1 per 16, every DPP 8-byte aligned
        if (KIND == 10) asm volatile(P E64 G0 P E64 G0 P E64 G0 P E64 G0 OPS);          // 1 per 16 in the VOP3 encoding (8 bytes, no DPP)
        if (KIND == 11) asm volatile(P DA P DA P DA P DA P DA P DA P DA P DA OPS);      // 1 per 8, aligned
        if (KIND == 8) asm volatile(G1 G0 G0 G1 G0 G0 G1 G0 OPS);                      // 3 per 64 (one per ~22: the pass kernel's spacing)
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[0] = s;
}

template <int KIND> void run(const char *name, float *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-44s", name);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 40000 / w;
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, d, 1.0000001f, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, d, 1.0000001f, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %7.1f", ms * 1e6 / ((double)iters * w));
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    printf("# wall-clock ns per 64-instruction block per SIMD; columns: W = 1, 2, 4, 8 waves per SIMD\n");
    run<0>("64 plain", d);
    run<1>("1 DPP per 64", d);
    run<2>("1 DPP per 32", d);
    run<8>("1 DPP per ~22 (3 per 64)", d);
    run<3>("1 DPP per 16", d);
    run<4>("1 DPP per 8", d);
    run<9>("1 DPP per 16, each on an 8-byte boundary", d);
    run<11>("1 DPP per 8, each on an 8-byte boundary", d);
    run<10>("1 VOP3-encoded plain add per 16", d);
    run<5>("2 adjacent DPP per 64", d);
    run<6>("4 adjacent DPP per 64", d);
    run<7>("2 groups of 4 adjacent DPP per ~64", d);
    return 0;
}
