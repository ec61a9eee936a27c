// Micro-benchmark: can the matrix pipe, the LDS fragment reads and the LDS-DMA staging of conv3_kernel's K loop overlap on
// gfx950, and how much of the MFMA rate survives?  The real kernel's ablations add up instead of overlapping (DESIGN.md 5:
// main loop 440 us = 268 MFMA + 130 fragment reads + 40 staging); this reproduces the loop's instruction mix without the
// convolution: 2 workgroups x 8 waves per CU (4 waves per SIMD), per K-group and wave 36 ds_read_b128 + 36
// v_mfma_f32_32x32x16_bf16 (2 + 2 per step, 18 steps, fragments of step s+1 read while step s multiplies), 5 LDS-DMA
// instructions (1 KiB each, L2-resident source) for the next K-group, one s_waitcnt vmcnt(0) + barrier per K-group.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { uint4 q; bf16x8 v; };

constexpr int STEPS = 18;
// bit 0: MFMAs, bit 1: fragment reads, bit 2: LDS-DMA of the next K-group, bit 3: vmcnt(0) + barrier per K-group,
// bit 4: fragment reads NOT software-pipelined (read, wait, multiply)
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loop(const uint4* __restrict__ src, int kgroups, float* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 stages x 38 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const uint4* mysrc = src + ((blockIdx.x & 63) * 8 + wave) * 5 * 64 + lane;       // 2.5 MiB source window in all: L2-resident
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int stage = 0;
    for (int g = 0; g < kgroups; ++g) {
        if (MODE & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (MODE & 4) {
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(lds + (stage ^ 1) * 38 * 1024 + (wave * 5 + c) * 1024 % (38 * 1024));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(mysrc + c * 64) : "memory", "m0");
            }
        }
        const uint4* base = reinterpret_cast<const uint4*>(lds + stage * 38 * 1024) + lane;
        Frag a[2], b[2];                                   // ping-pong: fragments of step s+1 are read while step s multiplies
        if (MODE & 2) { a[0].q = base[0]; b[0].q = base[64]; }
        else { a[0].q = b[0].q = make_uint4(lane, g, 1, 2); a[1].q = b[1].q = make_uint4(lane, g, 3, 4); }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if ((MODE & 2) && !(MODE & 16) && s + 1 < STEPS) { a[nxt].q = base[(2 * (s + 1)) * 64]; b[nxt].q = base[(2 * (s + 1) + 1) * 64]; }
            if ((MODE & 2) && (MODE & 16) && s > 0) { a[cur].q = base[(2 * s) * 64]; b[cur].q = base[(2 * s + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 1) {
                acc[(s & 1) * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur].v, b[cur].v, acc[(s & 1) * 2 + 0], 0, 0, 0);
                acc[(s & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cur].v, a[cur].v, acc[(s & 1) * 2 + 1], 0, 0, 0);
            } else {
                acc[0][0] += __builtin_bit_cast(float, a[cur].q.x ^ b[cur].q.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage ^= 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s_ = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_ += acc[i][0] + acc[i][7];
    if (s_ == 123.456f) sink[0] = s_;
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

template <int MODE>
static void run(const char* what, const uint4* src, float* sink, unsigned long long* cyc, int cus) {
    constexpr int WAVES = 8;
    const int kgroups = 2000, grid = cus * 2, lds_bytes = 2 * 38 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds_bytes, 0, src, 20, sink, cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemset(cyc, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop<MODE, WAVES>), dim3(grid), dim3(64 * WAVES), lds_bytes, 0, src, kgroups, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c = 0;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double per_kg_wave = (double)c / ((double)grid * WAVES) / kgroups;       // s_memtime ticks (100 MHz x ?) - report wall time instead
    const double us_per_kg_pair = ms * 1e3 / kgroups;                               // one K-group of both workgroups of a CU
    const double mfma_us = 2.0 * WAVES * 2 * STEPS * 32 / 4 / 2.4e3;                // 72 MFMAs per SIMD-wave pair... at 2.4 GHz: pure pipe time per pair
    printf("%-64s %7.3f us per K-group pair per CU   (pure MFMA pipe time at 2.4 GHz: %.3f us -> MFMA utilisation %.2f)   [memtime/kg/wave %.0f]\n", what,
           us_per_kg_pair, mfma_us, (MODE & 1) ? mfma_us / us_per_kg_pair : 0.0, per_kg_wave);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint4* src; float* sink; unsigned long long* cyc;
    CK(hipMalloc(&src, 16 << 20)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(src, 0, 16 << 20));
    printf("%d CUs; 2 workgroups x 8 waves per CU; per wave and K-group: 36 MFMA 32x32x16 bf16, 36 ds_read_b128, 5 LDS-DMA KiB\n", cus);
    run<1>("MFMA only", src, sink, cyc, cus);
    run<2>("fragment reads only", src, sink, cyc, cus);
    run<3>("MFMA + fragment reads (pipelined)", src, sink, cyc, cus);
    run<19>("MFMA + fragment reads (read, wait, multiply)", src, sink, cyc, cus);
    run<4 + 8>("LDS-DMA + wait + barrier only", src, sink, cyc, cus);
    run<1 + 4 + 8>("MFMA + LDS-DMA + wait + barrier", src, sink, cyc, cus);
    run<2 + 4 + 8>("fragment reads + LDS-DMA + wait + barrier", src, sink, cyc, cus);
    run<1 + 2 + 8>("MFMA + fragment reads + barrier", src, sink, cyc, cus);
    run<1 + 2 + 4 + 8>("MFMA + fragment reads + LDS-DMA + wait + barrier (the kernel's loop)", src, sink, cyc, cus);
    return 0;
}
