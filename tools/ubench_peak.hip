// Peak VALU issue rate of one SIMD, by WALL CLOCK (HIP events over >= 1 ms kernels), so that no
// assumption about the s_memtime tick or about where the waves are placed enters: W blocks of 256
// threads per CU (= W waves on every SIMD, all resident), each wave issuing ITER x 64 instructions.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench_peak tools/ubench_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float cs, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = cs + threadIdx.x * 1e-9f;
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)        // VGPR-only operands, 8 independent chains
            asm volatile(REP8("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 1)   // SGPR operand in every instruction
            asm volatile(REP8("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(cs));
        else if (KIND == 2)   // one dependent chain, VGPR-only
            asm volatile(REP64("v_add_f32 %0, %1, %0\n") : "+v"(a0) : "v"(c));
        else if (KIND == 3)   // fma, VGPR-only, 8 chains
            asm volatile(REP8("v_fma_f32 %0, %8, %0, %8\n v_fma_f32 %1, %8, %1, %8\n v_fma_f32 %2, %8, %2, %8\n v_fma_f32 %3, %8, %3, %8\n"
                              "v_fma_f32 %4, %8, %4, %8\n v_fma_f32 %5, %8, %5, %8\n v_fma_f32 %6, %8, %6, %8\n v_fma_f32 %7, %8, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 4)   // the pass kernel's mix: sub, mul by SGPR, sub -- chain-serial through one temp
            asm volatile(REP8("v_sub_f32 %0, %1, %2\n v_mul_f32 %0, %8, %0\n v_sub_f32 %3, %3, %0\n v_sub_f32 %0, %2, %1\n"
                              "v_mul_f32 %0, %8, %0\n v_add_f32 %4, %4, %0\n v_sub_f32 %0, %5, %6\n v_mul_f32 %0, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(cs));
        else if (KIND == 5)   // DPP form in every instruction
            asm volatile(REP8("v_add_f32_dpp %0, %8, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 6)   // every instruction a DPP form, row_shr:1
            asm volatile(REP8("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 7)   // every instruction a DPP form, wave_shr:1
            asm volatile(REP8("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 8)   // 1 of 8 DPP (wave_shr)
            asm volatile(REP8("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 9)   // 2 of 8 DPP, row_shr / row_shl (inside 16 lanes)
            asm volatile(REP8("v_add_f32_dpp %0, %8, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        else if (KIND == 10) {  // 8 plain + 2 ds_bpermute_b32 per group (results awaited at the end of the 64-block)
            float b0, b1;
            asm volatile(REP8("ds_bpermute_b32 %9, %11, %0\n v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "ds_bpermute_b32 %10, %11, %4\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) , "=&v"(b0), "=&v"(b1): "v"(c), "v"(addr));
            a7 += b0 * 0.f + b1 * 0.f;
        } else if (KIND == 11) {  // 8 plain + 2 ds_swizzle_b32 (rotate by 1 inside 32 lanes)
            float b0, b1;
            asm volatile(REP8("ds_swizzle_b32 %9, %0 offset:0xc020\n v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "ds_swizzle_b32 %10, %4 offset:0xc020\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) , "=&v"(b0), "=&v"(b1): "v"(c));
            a7 += b0 * 0.f + b1 * 0.f;
        } else if (KIND == 12)  // 2 of 8: v_mov_b32_dpp (a plain move through the DPP path)
            asm volatile(REP8("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                              "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char *name, float *d, unsigned long long *c)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-58s", name);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 40000 / w;     // 64 instructions each
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, d, c, 1.0000001f, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * w), dim3(256), 0, 0, d, c, 1.0000001f, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h = 0;
        (void)hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double ninst = 64.0 * iters * w;          // wave-instructions per SIMD
        printf("  %5.2f ns (%4.2f tick, %4.0f MHz)", ms * 1e6 / ninst, (double)h / (64.0 * iters) / w, (double)h / (ms * 1e3));
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    unsigned long long *c; (void)hipMalloc(&c, 4096 * sizeof(unsigned long long));
    printf("# wall-clock ns per wave64 VALU instruction per SIMD (s_memtime ticks per instruction per SIMD, s_memtime ticks per us); columns: W = 1, 2, 4, 8 waves per SIMD\n");
    run<0>("v_add_f32 v,v,v  8 chains", d, c);
    run<1>("v_add_f32 v,s,v  8 chains", d, c);
    run<2>("v_add_f32 v,v,v  1 chain", d, c);
    run<3>("v_fma_f32 v,v,v,v 8 chains", d, c);
    run<4>("sub / mul-by-SGPR / sub, chain-serial through one temp", d, c);
    run<5>("2 of 8 DPP (wave_shl / wave_shr)", d, c);
    run<6>("8 of 8 DPP row_shr:1", d, c);
    run<7>("8 of 8 DPP wave_shr:1", d, c);
    run<8>("1 of 8 DPP wave_shr:1", d, c);
    run<9>("2 of 8 DPP (row_shl / row_shr)", d, c);
    run<12>("2 of 8 v_mov_b32_dpp (wave_shl / wave_shr)", d, c);
    run<10>("8 plain + 2 ds_bpermute_b32 (ns per plain instruction)", d, c);
    run<11>("8 plain + 2 ds_swizzle_b32 rotate (ns per plain instruction)", d, c);
    return 0;
}
