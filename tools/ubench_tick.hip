// What does the per-tick hand-off of the level-split pass cost?  A workgroup of NW waves forms a
// pipeline: per tick every wave reads one row (3 x 16 B per lane) from the LDS buffer the wave
// before it filled in the previous tick, runs NV VALU instructions on it, writes it to its own
// buffer and joins an s_barrier.  Variants: SYNC 0 = no barrier and private buffers (upper bound
// of what removing the barrier could give), 1 = one barrier per tick (what k_bulk_split does),
// 2 = one barrier per TWO rows (rows handed over in pairs).  Printed: shader cycles per tick per
// workgroup (s_memtime) for 1, 2, 4, (8) workgroups per CU.
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o tools/ubench_tick tools/ubench_tick.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct alignas(16) F4 { float v[4]; };

template <int NW, int NV, int SYNC>
__global__ __launch_bounds__(64 * NW) void k(float *out, unsigned long long *cyc, float cs, int ticks)
{
    __shared__ F4 buf[NW][2][2][3][64];     // [wave][parity][row of a pair][field][lane]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    F4 e, x, y;
    for (int v = 0; v < 4; ++v) { e.v[v] = lane * 0.001f + v; x.v[v] = 0.5f * v; y.v[v] = 0.25f * lane; }
    for (int p = 0; p < 2; ++p) for (int r = 0; r < 2; ++r) { buf[w][p][r][0][lane] = e; buf[w][p][r][1][lane] = x; buf[w][p][r][2][lane] = y; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int src = SYNC == 0 ? w : (w + NW - 1) % NW;
    constexpr int RPB = SYNC == 2 ? 2 : 1;      // rows per barrier
    for (int t = 0; t < ticks; t += RPB) {
        const int par = (t / RPB) & 1;
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
            e = buf[src][par ^ 1][r][0][lane];
            x = buf[src][par ^ 1][r][1][lane];
            y = buf[src][par ^ 1][r][2][lane];
#pragma unroll
            for (int n = 0; n < NV / 24; ++n) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {                 // 6 VALU per element, like one H update
                    x.v[v] = x.v[v] - cs * (e.v[(v + 1) & 3] - e.v[v]);
                    y.v[v] = y.v[v] + cs * (x.v[v] - e.v[v]);
                }
            }
            buf[w][par][r][0][lane] = e;
            buf[w][par][r][1][lane] = x;
            buf[w][par][r][2][lane] = y;
        }
        if (SYNC != 0) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int v = 0; v < 4; ++v) s += e.v[v] + x.v[v] + y.v[v];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NW, int NV, int SYNC> void run(const char *name, float *d, unsigned long long *c)
{
    const int ticks = 2000;
    std::vector<unsigned long long> h(4096);
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    float ms = 0;
    double mhz = 0;
    printf("%-52s", name);
    for (int per_cu : {1, 2, 4, 8}) {
        if (per_cu * NW > 32) { printf(" %8s", "-"); continue; }
        const int blocks = 256 * per_cu;
        hipLaunchKernelGGL((k<NW, NV, SYNC>), dim3(blocks), dim3(64 * NW), 0, 0, d, c, 1.0000001f, 64);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NW, NV, SYNC>), dim3(blocks), dim3(64 * NW), 0, 0, d, c, 1.0000001f, ticks);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), c, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.begin() + blocks);
        printf(" %8.0f", (double)h[blocks / 2] / ticks);
        mhz = (double)h[blocks / 2] / (ms * 1e3);
    }
    printf("   [ticks/us %.0f]\n", mhz);
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    unsigned long long *c; (void)hipMalloc(&c, 4096 * sizeof(unsigned long long));
    printf("# shader cycles per tick (one row through one wave), median workgroup; columns: workgroups per CU\n");
    printf("%-52s %8s %8s %8s %8s\n", "variant", "1/CU", "2/CU", "4/CU", "8/CU");
    run<4, 0, 0>("4 waves,   0 VALU/tick, no barrier", d, c);
    run<4, 0, 1>("4 waves,   0 VALU/tick, barrier per tick", d, c);
    run<4, 0, 2>("4 waves,   0 VALU/tick, barrier per 2 rows", d, c);
    run<4, 96, 0>("4 waves,  96 VALU/tick, no barrier", d, c);
    run<4, 96, 1>("4 waves,  96 VALU/tick, barrier per tick", d, c);
    run<4, 96, 2>("4 waves,  96 VALU/tick, barrier per 2 rows", d, c);
    run<4, 192, 0>("4 waves, 192 VALU/tick, no barrier", d, c);
    run<4, 192, 1>("4 waves, 192 VALU/tick, barrier per tick", d, c);
    run<4, 192, 2>("4 waves, 192 VALU/tick, barrier per 2 rows", d, c);
    run<8, 96, 0>("8 waves,  96 VALU/tick, no barrier", d, c);
    run<8, 96, 1>("8 waves,  96 VALU/tick, barrier per tick", d, c);
    run<8, 96, 2>("8 waves,  96 VALU/tick, barrier per 2 rows", d, c);
    run<2, 384, 0>("2 waves, 384 VALU/tick, no barrier", d, c);
    run<2, 384, 1>("2 waves, 384 VALU/tick, barrier per tick", d, c);
    run<1, 768, 0>("1 wave,  768 VALU/tick, no barrier", d, c);
    return 0;
}
