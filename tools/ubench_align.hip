// Which 8-byte instruction encodings care where they start?  15 plain v_add_f32 + ONE instruction X, X placed on
// an 8-byte boundary (".p2align 3; X") or 4 bytes past one (".p2align 3; s_nop 0; X"), W waves per SIMD.
// Wall-clock ns per group of 16 per SIMD.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench_align tools/ubench_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define P15 "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n" \
            "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %0, %8, %0\n"
#define AL ".p2align 3\n"
#define MIS ".p2align 3\n s_nop 0\n"
#define ALN ".p2align 3\n s_nop 0\n s_nop 0\n"      /* aligned, with the same two extra s_nop as a control */
#define REP4(x) x x x x
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q) : "v"(c), "v"(lds_addr), "v"(gp) : "memory"
// operands: %0-%7 accumulators, %8 q (4 dwords), %9 c, %10 lds address, %11 global pointer
#undef P15
#define P15 "v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n" \
            "v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n v_add_f32 %0, %9, %0\n"

#define X_DPP   "v_add_f32_dpp %7, %9, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define X_DSR   "ds_read_b128 %8, %10\n"
#define X_DSW   "ds_write_b128 %10, %8\n"
#define X_GLD   "global_load_dwordx4 %8, %11, off\n"
#define X_LIT   "v_add_f32 %7, 0x3f800001, %7\n"
#define X_VOP3  "v_add_f32_e64 %7, %9, %7\n"
#define X_CND   "v_cndmask_b32_e64 %7, %7, %9, vcc\n"
#define X_SDWA  "v_add_f32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define X_MAD   "v_mad_u64_u32 %8, vcc, %9, %9, 0\n"

typedef float f4 __attribute__((ext_vector_type(4)));

#define KERNEL(NAME, PRE, X, TAIL)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float *out, const float *g, float cs, int iters)                      \
    {                                                                                                                   \
        __shared__ f4 sm[256];                                                                                         \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float c = cs + threadIdx.x * 1e-9f;                                                                            \
        f4 q = {a0, a1, a2, a3};                                                                                        \
        sm[threadIdx.x] = q;                                                                                            \
        const unsigned lds_addr = (unsigned)(threadIdx.x * 16);                                                         \
        const float *gp = g + threadIdx.x * 4;                                                                          \
        for (int i = 0; i < iters; ++i) asm volatile(REP4(P15 PRE X) TAIL OPS);                                         \
        const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q.x + q.y + q.z + q.w + sm[(threadIdx.x + 1) & 255].x;  \
        if (s == 123.456f) out[0] = s;                                                                                  \
    }

#define TRIO(N, X, TAIL) KERNEL(k_##N##_al, AL, X, TAIL) KERNEL(k_##N##_mis, MIS, X, TAIL) KERNEL(k_##N##_aln, ALN, X, TAIL)
TRIO(dpp, X_DPP, "")
TRIO(dsr, X_DSR, "s_waitcnt lgkmcnt(0)\n")
TRIO(dsw, X_DSW, "s_waitcnt lgkmcnt(0)\n")
TRIO(gld, X_GLD, "s_waitcnt vmcnt(0)\n")
TRIO(lit, X_LIT, "")
TRIO(vop3, X_VOP3, "")
TRIO(cnd, X_CND, "")
TRIO(sdwa, X_SDWA, "")

typedef void (*kern_t)(float *, const float *, float, int);
void run(const char *name, kern_t k, float *d, float *g)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-40s", name);
    for (int w : {1, 2, 4}) {
        const int iters = 20000 / w;
        hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, d, g, 1.0000001f, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, d, g, 1.0000001f, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %7.1f", ms * 1e6 / ((double)iters * w * 4));
    }
    printf("\n");
    fflush(stdout);
}
#define RUN3(N, label) run(label ", on a boundary", k_##N##_al, d, g); run(label ", 4 bytes past", k_##N##_mis, d, g); run(label ", on a boundary + 2 s_nop", k_##N##_aln, d, g);
int main()
{
    float *d, *g; (void)hipMalloc(&d, 64); (void)hipMalloc(&g, 1 << 16); (void)hipMemset(g, 0, 1 << 16);
    printf("# wall-clock ns per group (15 plain v_add_f32 + X) per SIMD; columns: W = 1, 2, 4 waves per SIMD (plain: 16 x 0.98 = 15.7 at W >= 2)\n");
    RUN3(dpp, "v_add_f32_dpp")
    RUN3(vop3, "v_add_f32_e64")
    RUN3(lit, "v_add_f32 + literal")
    RUN3(cnd, "v_cndmask_b32_e64")
    RUN3(sdwa, "v_add_f32_sdwa")
    RUN3(dsr, "ds_read_b128")
    RUN3(dsw, "ds_write_b128")
    RUN3(gld, "global_load_dwordx4")
    return 0;
}
