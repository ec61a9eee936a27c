// Reproducer attempt for profiles/experiments/r02_lane_mask.md: a 64-bit lane mask written by two VALU compares, combined on the
// scalar unit (s_and_b64) and consumed by v_cndmask, in a kernel that allocates ~240 VGPRs and runs at two waves per SIMD with
// MFMA, LDS and global-store traffic around the sequence.  The same select is also computed without the scalar unit
// ((c1 ? a : 0) * (c2 ? 1 : 0), each compare consumed by its own v_cndmask); any lane where the two differ is counted.
//   hipcc --offload-arch=gfx950 -O3 -o lane_mask lane_mask.hip && ./lane_mask [nops between v_cmp and s_and: 0..7]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NOPS, bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void repro(const float* __restrict__ in, float4* __restrict__ out,
                                                                                      unsigned* __restrict__ bad, int iters, unsigned W, unsigned H) {
    __shared__ float lds[64 * 65];
    const int lane = threadIdx.x;
    float pad[BIG ? 160 : 4];                       // register ballast: kept live across the loop
#pragma unroll
    for (int i = 0; i < (BIG ? 160 : 4); ++i) pad[i] = in[(i * 64 + lane) & 4095];
    f32x16 acc = {0};
    bf16x8 A, B;
#pragma unroll
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * (lane & 7) + i); B[i] = (__bf16)(0.02f * i); }
    unsigned state = 1234567u + 7919u * (blockIdx.x * 64 + lane);
    unsigned mism = 0, lanes_bad = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const float u = (float)(state >> 8) * (1.0f / 16777216.0f) * 1.2f - 0.1f;       // a few percent of the samples leave [0, 1)
        state = state * 1664525u + 1013904223u;
        const float v = (float)(state >> 8) * (1.0f / 16777216.0f) * 1.2f - 0.1f;
        const float ix = u * (float)W - 0.5f, iy = v * (float)H - 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const float dx = ix - fx, ey = 1.0f - (iy - fy);
        const float dy = iy - fy, ex = 1.0f - dx;
        const unsigned x0 = (unsigned)(int)fx, x1 = x0 + 1u, y0 = (unsigned)(int)fy, y1 = y0 + 1u;
        const float p0 = ex * ey, p1 = dx * ey, p2 = ex * dy, p3 = dx * dy;
        // reference: every compare consumed by its own select
        const float ax0 = x0 < W ? ex : 0.0f, ax1 = x1 < W ? dx : 0.0f, ay0 = y0 < H ? ey : 0.0f, ay1 = y1 < H ? dy : 0.0f;
        const float r0 = ax0 * ay0, r1 = ax1 * ay0, r2 = ax0 * ay1, r3 = ax1 * ay1;
        // the form under test: the instruction sequence of tap_geometry()'s four weights as the compiler emitted it in
        // bwd_scatter_sorted_kernel<true, true> (profiles/experiments/r02_lane_mask.md), pinned in asm; NOPS s_nop 0 before the last select
        float w0 = p0, w1 = p1, w2 = p2, w3 = p3;
        unsigned long long mx1, my0, my1, mt;
        asm volatile("v_cmp_gt_u32_e64 %[my0], %[H], %[y0]\n\t"
                     "v_cmp_gt_u32_e64 %[my1], %[H], %[y1]\n\t"
                     "v_cmp_gt_u32_e32 vcc, %[W], %[x0]\n\t"
                     "v_cmp_gt_u32_e64 %[mx1], %[W], %[x1]\n\t"
                     "s_and_b64 %[mt], %[my1], vcc\n\t"
                     "s_and_b64 vcc, %[my0], vcc\n\t"
                     "v_cndmask_b32_e32 %[w0], 0, %[w0], vcc\n\t"
                     "s_and_b64 vcc, %[my1], %[mx1]\n\t"
                     "v_cndmask_b32_e64 %[w2], 0, %[w2], %[mt]\n\t"
                     "v_cndmask_b32_e32 %[w3], 0, %[w3], vcc\n\t"
                     "s_and_b64 vcc, %[my0], %[mx1]\n\t"
                     ".rept %[nops]\n\ts_nop 0\n\t.endr\n\t"
                     "v_cndmask_b32_e32 %[w1], 0, %[w1], vcc"
                     : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [mx1] "=&s"(mx1), [my0] "=&s"(my0), [my1] "=&s"(my1), [mt] "=&s"(mt)
                     : [W] "s"(W), [H] "s"(H), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [nops] "n"(NOPS)
                     : "vcc");
        const bool differ = w0 != r0 || w1 != r1 || w2 != r2 || w3 != r3;
        mism += differ ? 1u : 0u;
        if (w1 != r1) lanes_bad += 1u;
        const float w_mask = w1, w_sep = r1, p = w0 + w2 + w3;
        // traffic around it, as in the decoder-backward kernel: 16-byte stores, LDS exchange, matrix instructions
        out[((size_t)blockIdx.x * iters + it) * 64 + lane] = make_float4(w_mask, w_sep, p, pad[1]);
        lds[lane * 65 + (it & 63)] = w_mask;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
        pad[0] += lds[((lane + 1) & 63) * 65 + (it & 63)] * 1e-9f;
#pragma unroll
        for (int i = 0; i < (BIG ? 160 : 4); ++i) asm volatile("" : "+v"(pad[i]));      // the ballast stays in registers
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < (BIG ? 160 : 4); ++i) s += pad[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) out[0].x = s;
    bad[(blockIdx.x * 64 + lane) * 2] = mism;
    bad[(blockIdx.x * 64 + lane) * 2 + 1] = lanes_bad;
}

template <int NOPS, bool BIG>
static void run(const float* din, float4* dout, unsigned* dbad, int blocks, int iters) {
    CK(hipMemset(dbad, 0, sizeof(unsigned) * blocks * 128));
    hipLaunchKernelGGL((repro<NOPS, BIG>), dim3(blocks), dim3(64), 0, 0, din, dout, dbad, iters, 256u, 256u);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(blocks * 128);
    CK(hipMemcpy(h.data(), dbad, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long total = 0; unsigned long long per_group[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int l = 0; l < 64; ++l) { total += h[(b * 64 + l) * 2]; per_group[l >> 4] += h[(b * 64 + l) * 2]; }
    printf("registers %-5s nops %d: %llu mismatching selects of %llu (lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu)\n", BIG ? "~240" : "small", NOPS,
           total, (unsigned long long)blocks * 64 * iters, per_group[0], per_group[1], per_group[2], per_group[3]);
}

int main() {
    const int blocks = 256 * 8 * 4, iters = 512;      // four rounds of two waves per SIMD
    float* din; float4* dout; unsigned* dbad;
    CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, (size_t)blocks * iters * 64 * 16)); CK(hipMalloc(&dbad, sizeof(unsigned) * blocks * 128));
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * i;
    CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        run<0, true>(din, dout, dbad, blocks, iters);
        run<0, false>(din, dout, dbad, blocks, iters);
    }
    run<1, true>(din, dout, dbad, blocks, iters);
    run<2, true>(din, dout, dbad, blocks, iters);
    run<4, true>(din, dout, dbad, blocks, iters);
    return 0;
}
