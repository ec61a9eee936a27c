// Micro-benchmark: how many bytes per clock a CU can READ on MI355X, as a function of where the data lives and of how it is
// read.  Answers DESIGN.md section 5's open question: conv3_kernel stages its operands at ~7.5 B/clk per CU (4.2 TB/s chip-wide)
// and upfir_kernel reads its scratch at ~5.6 B/clk per CU - is that a ceiling of the machine or of those kernels?
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
// Every workgroup (256 threads) reads `span` bytes of a working set of `set_bytes` bytes over and over with 16-byte loads per
// lane (a wave instruction = 1 KiB contiguous), `DEPTH` loads in flight per wave.  Working sets: 256 KiB per workgroup window
// inside a set that is L2-sized (2 MiB per XCD), Infinity-Cache-sized (128 MiB) or HBM-sized (4 GiB).  Forms: plain
// global_load_dwordx4 into registers, and global_load_lds_dwordx4 (LDS-DMA, what conv3_kernel uses).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: registers; mode 1: LDS-DMA.  Each wave walks its workgroup's window with stride 4 KiB x (waves), DEPTH loads per wait.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void reader(const uint4* __restrict__ buf, unsigned long long set_items, unsigned long long window_items,
                                               int rounds, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[DEPTH * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // window of this workgroup: windows tile the set; workgroup k of XCD (k % 8) takes window k (so each XCD's L2 sees 1/8 of them)
    const unsigned long long windows = set_items / window_items;
    float acc = 0.0f;
    const unsigned long long per_wave = window_items / 4;             // items of this wave's quarter
    const unsigned long long steps = per_wave / 64;                   // 1-KiB instructions per pass
    for (int r = 0; r < rounds; ++r) {
        // round r: window (k + r * grid) mod windows - the same window every round when the set is small (cache-resident), a
        // walk through the whole set when it is large; k mod 8 (the XCD) is preserved, grid and windows being multiples of 8
        const unsigned long long w0 = (((unsigned long long)blockIdx.x + (unsigned long long)r * gridDim.x) % windows) * window_items;
        const uint4* base = buf + w0 + (unsigned long long)wave * per_wave + lane;
        for (unsigned long long s = 0; s < steps; s += DEPTH) {
            if (MODE == 0) {
                uint4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = base[(s + d) * 64];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) acc += __uint_as_float(v[d].x ^ v[d].y ^ v[d].z ^ v[d].w);
            } else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(lds + (d * 4 + wave) * 1024);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(base + (s + d) * 64) : "memory", "m0");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if (MODE == 1) acc = reinterpret_cast<float*>(lds)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}

// upfir_kernel's read pattern on its own: a (2H+1) x (2W+1) x C fp32 scratch, H = W = 256, C = 128 (row pitch 262 656 B).  A
// workgroup of 4 waves owns 8 input columns (19 scratch columns of 512 B) and 8 output rows (11 scratch rows); per scratch row
// every wave issues 5 loads of 16 B per lane, lanes 0-31 at column X0 - 1 + jj, lanes 32-63 two columns (1 KiB) further.
// PAIR = false: the same bytes, but every instruction reads 1 KiB contiguous (both half waves side by side).
template <bool PAIR>
__global__ __launch_bounds__(256) void upfir_pattern(const uint4* __restrict__ buf, int views, float* sink) {
    const int TH = 513, TW = 513, C16 = 32;                       // 16-byte items per scratch pixel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = 256 / 4, cblocks = 256 / 8;                // 4 two-row blocks per workgroup pass, 8 input columns per workgroup
    float acc = 0.0f;
    for (int wg = blockIdx.x; wg < views * groups * cblocks; wg += gridDim.x) {
        const int cb = wg % cblocks, bg = (wg / cblocks) % groups, n = wg / (cblocks * groups);
        const int X0 = 2 * (8 * cb + 2 * wave + (PAIR ? (lane >> 5) : 0));
        const uint4* vb = buf + (unsigned long long)n * TH * TW * C16;
        for (int r = 0; r < 11; ++r) {
            const int ty = min(max(2 * 4 * bg - 1 + r, 0), TH - 1);
            uint4 v[5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int tx = PAIR ? min(max(X0 - 1 + jj, 0), TW - 1) : min(max(X0 - 1 + 2 * jj + (lane >> 5), 0), TW - 1);
                v[jj] = vb[((unsigned long long)ty * TW + tx) * C16 + (lane & 31)];
            }
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) acc += __uint_as_float(v[jj].x ^ v[jj].y ^ v[jj].z ^ v[jj].w);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <bool PAIR>
static void run_upfir(const uint4* buf, float* sink, int cus) {
    const int views = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((upfir_pattern<PAIR>), dim3(cus * 8), dim3(256), 0, 0, buf, views, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double unique = 8.0 * 513 * 513 * 512;
    printf("upfir read pattern (%s): %.1f us for %.2f GB of scratch = %.2f TB/s of unique bytes (loads issued: %.2fx)\n",
           PAIR ? "half waves 1 KiB apart, as shipped" : "1 KiB contiguous per instruction", ms * 1e3, unique / 1e9, unique / (ms * 1e-3) / 1e12,
           PAIR ? 11.0 / 8 * 5 / 2 : 11.0 / 8 * 5 / 2);
}

template <int MODE, int DEPTH>
static void run(const char* what, const uint4* buf, unsigned long long set_bytes, unsigned long long window_bytes, int wgs_per_cu, int cus, float* sink,
                double clock_ghz) {
    const unsigned long long set_items = set_bytes / 16, window_items = window_bytes / 16;
    const int grid = cus * wgs_per_cu;
    // bytes per workgroup per launch ~ 64 MiB / (wgs per CU)
    int rounds = (int)((64ull << 20) / wgs_per_cu / window_bytes);
    if (rounds < 1) rounds = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((reader<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, set_items, window_items, rounds, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((reader<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, set_items, window_items, rounds, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * rounds * (double)window_bytes;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s set %8.1f MiB  window %6llu KiB  %d WG/CU  depth %2d : %6.2f TB/s = %5.1f B/clk per CU (at %.2f GHz)\n", what, set_bytes / 1048576.0,
           window_bytes >> 10, wgs_per_cu, DEPTH, tbs, bytes / (ms * 1e-3) / cus / (clock_ghz * 1e9), clock_ghz);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.2;                 // the in-run clock of the render bench; printed B/clk scale with it
    printf("%s, %d CUs\n", prop.name, cus);
    const unsigned long long total = 4ull << 30;
    uint4* buf; float* sink;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, total));
    struct { const char* name; unsigned long long set, window; } cases[] = {
        {"L2-resident (16 MiB / 8 XCDs)", 16ull << 20, 64ull << 10},
        {"L2-resident, 256 KiB windows", 16ull << 20, 256ull << 10},
        {"Infinity-Cache-resident", 128ull << 20, 256ull << 10},
        {"HBM (4 GiB)", 4ull << 30, 256ull << 10},
        {"HBM (4 GiB), 4 MiB windows", 4ull << 30, 4ull << 20},
    };
    run_upfir<true>(buf, sink, cus);
    run_upfir<false>(buf, sink, cus);
    for (auto& c : cases) {
        char nm[96];
        snprintf(nm, sizeof nm, "load -> VGPR, %s", c.name);
        run<0, 8>(nm, buf, c.set, c.window, 4, cus, sink, ghz);
        run<0, 16>(nm, buf, c.set, c.window, 8, cus, sink, ghz);
        snprintf(nm, sizeof nm, "LDS-DMA, %s", c.name);
        run<1, 8>(nm, buf, c.set, c.window, 2, cus, sink, ghz);
        run<1, 8>(nm, buf, c.set, c.window, 4, cus, sink, ghz);
    }
    return 0;
}
