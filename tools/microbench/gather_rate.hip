// Micro-benchmark: cost of one vector-memory wave-instruction on gfx950 as a function of the lane->address
// pattern (how many distinct 128-byte lines one instruction touches), the load width and the footprint
// (L1- / L2- / MALL-resident).  Answers "what bounds the tri-plane gather": DESIGN.md section 6.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// PATTERN: lines touched by one wave-instruction / bytes used per line
//  0: 64 lines x 16 B      1: 32 lines x 2x16 B (shipped render kernel)   2: 16 lines x 4x16 B
//  3: 8 lines x 128 B (8 consecutive lanes per line)   4: 8 consecutive lines (1 KiB contiguous)
//  5: one address for all lanes                        6: 8 lines x 128 B, lanes interleaved (lane&7 = line)
//  7: 16 quads x 64 B    8: quads Q and Q+8 read the two halves of one line (proposed render mapping)    9: 32 lane pairs x 32 B
template <int PATTERN, int WIDTH>
__global__ __launch_bounds__(256, 2) void gather(const char* __restrict__ base, uint32_t line_mask, int iters, float* out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t g, off;
    if (PATTERN == 0) { g = lane; off = 0; }
    else if (PATTERN == 1) { g = lane & 31; off = (lane >> 5) * 64; }
    else if (PATTERN == 2) { g = lane & 15; off = (lane >> 4) * 32; }
    else if (PATTERN == 3) { g = lane >> 3; off = (lane & 7) * 16; }
    else if (PATTERN == 4) { g = 0; off = lane * 16; }
    else if (PATTERN == 5) { g = 0; off = 0; }
    else if (PATTERN == 6) { g = lane & 7; off = (lane >> 3) * 16; }
    else if (PATTERN == 7) { g = lane >> 2; off = (lane & 3) * 16; }                       // 16 quads, 64 B each, 16 lines
    else if (PATTERN == 8) { g = (lane >> 2) & 7; off = (lane >> 5) * 64 + (lane & 3) * 16; } // quads Q, Q+8: halves of one line
    else { g = lane >> 1; off = (lane & 1) * 16; }                                         // 32 pairs, 32 B each
    const uint32_t gsalt = g * 0x9E3779B9u;
    uint32_t state = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            state = state * 1664525u + 1013904223u;
            uint32_t h = (state ^ gsalt) * 0x85EBCA6Bu;
            h ^= h >> 15;
            uint32_t line = h & line_mask;
            if (PATTERN == 4) line &= ~7u;
            // patterns 1/2 use a sub-offset that walks the lane's slice of the line like the 4 (2) loads of a tap
            uint32_t sub = (PATTERN == 1) ? (u & 3) * 16 : (PATTERN == 2) ? (u & 1) * 16 : (PATTERN == 0) ? (u & 7) * 16 : 0;
            const char* p = base + (size_t)line * 128 + off + sub;
            if (WIDTH == 4) v[u] = *(const f4*)p;
            else if (WIDTH == 2) { f2 t = *(const f2*)p; v[u] = f4{t.x, t.y, 0.f, 0.f}; }
            else { float t = *(const float*)p; v[u] = f4{t, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int PATTERN, int WIDTH>
static void run(const char* name, const char* base, size_t footprint, float* out, int cus, double ghz)
{
    const int blocks = cus * 2 * 4, iters = 2048;
    const uint32_t mask = (uint32_t)(footprint / 128 - 1);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<PATTERN, WIDTH><<<blocks, 256>>>(base, mask, 64, out);
    CK(hipEventRecord(a));
    gather<PATTERN, WIDTH><<<blocks, 256>>>(base, mask, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instrs_per_cu = (double)blocks * 4 * iters * 8 / cus;
    const double ns = ms * 1e6 / instrs_per_cu;
    const double lane_bytes = 64.0 * 4 * WIDTH;
    printf("%-44s w=%d  footprint %8zu KiB  %7.2f ns/instr/CU  %6.1f clk@%.1fGHz  %8.1f GB/s lane bytes\n", name, WIDTH, footprint >> 10,
           ns, ns * ghz, ghz, lane_bytes * instrs_per_cu * cus / (ms * 1e-3) * 1e-9);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    printf("%s  CUs %d  clock %.2f GHz\n", prop.name, cus, ghz);
    const size_t max_fp = 64u << 20;
    char* base; float* out;
    CK(hipMalloc(&base, max_fp + 4096)); CK(hipMalloc(&out, 64));
    std::vector<float> h(max_fp / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
    CK(hipMemcpy(base, h.data(), max_fp, hipMemcpyHostToDevice));
    const size_t fps[] = {8u << 10, 2u << 20, 32u << 20};
    for (size_t fp : fps) {
        run<0, 4>("64 lines x 16 B", base, fp, out, cus, ghz);
        run<1, 4>("32 lines x 2x16 B (shipped kernel)", base, fp, out, cus, ghz);
        run<2, 4>("16 lines x 4x16 B", base, fp, out, cus, ghz);
        run<3, 4>("8 lines x 128 B (8 adjacent lanes / line)", base, fp, out, cus, ghz);
        run<6, 4>("8 lines x 128 B (lanes interleaved)", base, fp, out, cus, ghz);
        run<7, 4>("16 quads x 64 B (16 lines)", base, fp, out, cus, ghz);
        run<8, 4>("8 lines, quad Q / Q+8 = halves", base, fp, out, cus, ghz);
        run<9, 4>("32 lane pairs x 32 B", base, fp, out, cus, ghz);
        run<4, 4>("1 KiB contiguous", base, fp, out, cus, ghz);
        run<5, 4>("single address", base, fp, out, cus, ghz);
        run<1, 2>("32 lines x 2x8 B", base, fp, out, cus, ghz);
        run<1, 1>("32 lines x 2x4 B", base, fp, out, cus, ghz);
        run<3, 2>("8 lines, 8 lanes x 8 B", base, fp, out, cus, ghz);
        run<3, 1>("8 lines, 8 lanes x 4 B", base, fp, out, cus, ghz);
        run<4, 1>("256 B contiguous (dword)", base, fp, out, cus, ghz);
    }
    return 0;
}
