// Round 4 (profiles/experiments/r04_bwd_lanes48_rootcause.md): stand-alone test of what the ISA-patch bisection of
// bwd_scatter_sorted_kernel<true, true> points at: a packed-fp32 instruction whose LOW result takes the HIGH register of its src1
// pair (op_sel:[0,1]) reads zeros in lanes 48..63 now and then when the SIMD's other wave keeps the matrix pipe busy.
// Every wave alternates bursts of v_mfma_f32_32x32x16_bf16 with bursts of the sequence under test (two waves per SIMD, 64-thread
// workgroups, 256 registers - the failing kernel's shape), so at any moment some SIMDs hold one wave of each kind.
//   form 0: v_cmp vcc; s_nop 1; v_cndmask hi, 0, fy, vcc; v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]      (the failing site)
//   form 1: the pk_mul alone (b's high register written long before)
//   form 2: as 0 with the operands commuted: v_pk_mul_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1]                    (the form that never failed)
//   form 3: as 0 with v_pk_add_f32
//   form 4: as 0 with default op_sel (lo * lo, hi * hi)
//   forms 5..12: which operand position / which half matters (see the SEQ list)
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel pk_opsel.hip && ./pk_opsel [form] [mfma burst] [test burst] [launches] [nops behind each mfma]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float seq1(float wx0, float wx1, float wy0, float fy, unsigned y1, unsigned H) {
    float r;
    // fixed registers: the failing site's own numbers (v[6:7] = x weights, v[10:11] = y weights, v[8:9] = result)
    asm volatile("v_mov_b32 v6, %1\n\tv_mov_b32 v7, %2\n\tv_mov_b32 v10, %3\n\tv_mov_b32 v11, %4\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                 "v_mov_b32 %0, v8"
                 : "=v"(r) : "v"(wx0), "v"(wx1), "v"(wy0), "v"(fy), "v"(y1), "s"(H) : "v6", "v7", "v8", "v9", "v10", "v11", "vcc");
    return r;
}
// (the instruction under test is pasted by the preprocessor: inline asm has no string operands).  v[12:13] = (wx1, wy0) is the addend of the fma forms.
#define SEQ(NAME, INSN, RES)                                                                                               \
    __device__ __forceinline__ float NAME(float wx0, float wx1, float wy0, float fy, unsigned y1, unsigned H) {           \
        float r;                                                                                                           \
        asm volatile("v_mov_b32 v6, %1\n\tv_mov_b32 v7, %2\n\tv_mov_b32 v10, %3\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\t"     \
                     "v_cmp_gt_u32_e32 vcc, %6, %5\n\t"                                                                    \
                     "s_nop 1\n\t"                                                                                         \
                     "v_cndmask_b32_e32 v11, 0, %4, vcc\n\t"                                                               \
                     "s_nop 7\n\ts_nop 7\n\t" INSN "\n\t"                                                               \
                     "v_mov_b32 %0, " RES                                                                                  \
                     : "=v"(r) : "v"(wx0), "v"(wx1), "v"(wy0), "v"(fy), "v"(y1), "s"(H)                                    \
                     : "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "vcc");                                         \
        return r;                                                                                                          \
    }
SEQ(seq0, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]", "v8")
SEQ(seq2, "v_pk_mul_f32 v[8:9], v[10:11], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]", "v8")
SEQ(seq3, "v_pk_add_f32 v[8:9], v[6:7], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]", "v8")
SEQ(seq4, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11]", "v8")
SEQ(seq5, "v_pk_fma_f32 v[8:9], v[6:7], v[10:11], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v8")     // lo = wx0 * fy + wx1
SEQ(seq6, "v_pk_fma_f32 v[8:9], v[6:7], v[10:11], v[12:13] op_sel:[0,0,1] op_sel_hi:[1,1,0]", "v8")     // lo = wx0 * wy0 + wy0 (src2 high)
SEQ(seq7, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel:[0,1]", "v8")                                    // lo = wx0 * fy, hi = wx1 * fy
SEQ(seq8, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel_hi:[1,0]", "v9")                                 // hi = wx1 * wy0 (src1 low for the high result)
SEQ(seq9, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel:[1,1] op_sel_hi:[0,0]", "v8")                    // lo = wx1 * fy
SEQ(seq10, "v_pk_fma_f32 v[8:9], v[6:7], v[10:11], v[12:13] op_sel:[1,0,0]", "v8")                       // lo = wx1 * wy0 + wx1 (src0 high: the library's commonest form)
SEQ(seq11, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v8, v9", "v8") // the failing instruction's HIGH result (wx1 * wy0)
SEQ(seq12, "v_pk_mul_f32 v[8:9], v[6:7], v[10:11]", "v9")                                                // default op_sel, high result wx1 * fy
SEQ(seq13, "v_pk_mul_f32 v[8:9], v[10:11], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]", "v8")                 // both sources the SAME pair (render_kernel's site): wy0 * fy
SEQ(seq14, "v_pk_fma_f32 v[8:9], v[6:7], v[10:11], v[12:13] op_sel:[0,1,1] op_sel_hi:[1,0,0]", "v8")    // lo = wx0 * fy + wy0
SEQ(seq15, "v_pk_mov_b32 v[8:9], v[6:7], v[10:11] op_sel:[1,0]", "v8")                                   // v_pk_mov_b32: lo = src0.hi (wx1): the library's two instances
SEQ(seq16, "v_pk_mov_b32 v[8:9], v[6:7], v[10:11] op_sel:[1,0]", "v9")                                   //               hi = src1.lo (wy0)
SEQ(seq17, "v_pk_mov_b32 v[8:9], v[6:7], v[10:11] op_sel:[0,1]", "v8")                                   //               lo = src0.lo (wx0)
SEQ(seq18, "v_pk_mov_b32 v[8:9], v[6:7], v[10:11] op_sel:[0,1]", "v9")                                   //               hi = src1.hi (fy)

template <int FORM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const float* __restrict__ in, unsigned* __restrict__ bad,
                                                                                      int rounds, int mfma_burst, int test_burst, unsigned H) {
    __shared__ float lds[64 * 68];                  // 17 KB per workgroup like the failing kernel (limits the CU to 9 workgroups)
    const int lane = threadIdx.x;
    lds[lane] = in[lane];
    f32x16 acc[4];
    bf16x8 A, B;
#pragma unroll
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * (lane & 7) + i); B[i] = (__bf16)(0.02f * i); }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    unsigned state = 1234567u + 7919u * (blockIdx.x * 64 + lane);
    unsigned mism = 0;
    // waves start their first burst at different points of the cycle
    for (int r = 0; r < rounds; ++r) {
        if (((r + blockIdx.x) & 1) == 0) {
            for (int m = 0; m < mfma_burst; ++m) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[k], 0, 0, 0);
#ifdef MFMA_PAD
                    asm volatile("s_nop 7\n\ts_nop 7");
#endif
                }
            }
        } else {
            for (int t = 0; t < test_burst; ++t) {
                state = state * 1664525u + 1013904223u;
                const float fx = (float)(state >> 8) * (1.0f / 16777216.0f) * 0.98f + 0.01f;
                state = state * 1664525u + 1013904223u;
                const float fy = (float)(state >> 8) * (1.0f / 16777216.0f) * 0.98f + 0.01f;
                const float wx0 = 1.0f - fx, wx1 = fx, wy0 = 1.0f - fy;
                const unsigned y1 = (state >> 3) % (H - 1u);                  // always valid: the select must pass fy through
                float got, want;
                switch (FORM) {       // `want` by plain one-result instructions (inline asm: the compiler must not pair them up)
                    case 0: got = seq0(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(fy)); break;
                    case 1: got = seq1(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(fy)); break;
                    case 2: got = seq2(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(fy)); break;
                    case 3: got = seq3(wx0, wx1, wy0, fy, y1, H); asm volatile("v_add_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(fy)); break;
                    case 4: got = seq4(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(wy0)); break;
                    case 5: got = seq5(wx0, wx1, wy0, fy, y1, H); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want) : "v"(wx0), "v"(fy), "v"(wx1)); break;
                    case 6: got = seq6(wx0, wx1, wy0, fy, y1, H); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want) : "v"(wx0), "v"(wy0), "v"(wy0)); break;
                    case 7: got = seq7(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx0), "v"(fy)); break;
                    case 8: got = seq8(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx1), "v"(wy0)); break;
                    case 9: got = seq9(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx1), "v"(fy)); break;
                    case 10: got = seq10(wx0, wx1, wy0, fy, y1, H); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want) : "v"(wx1), "v"(wy0), "v"(wx1)); break;
                    case 11: got = seq11(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx1), "v"(wy0)); break;
                    case 12: got = seq12(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wx1), "v"(fy)); break;
                    case 13: got = seq13(wx0, wx1, wy0, fy, y1, H); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(wy0), "v"(fy)); break;
                    case 14: got = seq14(wx0, wx1, wy0, fy, y1, H); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want) : "v"(wx0), "v"(fy), "v"(wy0)); break;
                    case 15: got = seq15(wx0, wx1, wy0, fy, y1, H); want = wx1; break;
                    case 16: got = seq16(wx0, wx1, wy0, fy, y1, H); want = wy0; break;
                    case 17: got = seq17(wx0, wx1, wy0, fy, y1, H); want = wx0; break;
                    default: got = seq18(wx0, wx1, wy0, fy, y1, H); want = fy; break;
                }
                if (__float_as_uint(got) != __float_as_uint(want)) {
                    ++mism;
                    atomicAdd(bad + 2 + lane, 1u);
                    if (got == 0.0f) atomicAdd(bad + 1, 1u);
                }
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 12345.678f) bad[70] = 1u;               // keeps the matrix work alive
    if (mism) atomicAdd(bad, mism);
}

int main(int argc, char** argv) {
    const int form = argc > 1 ? atoi(argv[1]) : 0, mb = argc > 2 ? atoi(argv[2]) : 8, tb = argc > 3 ? atoi(argv[3]) : 8;
    const int launches = argc > 4 ? atoi(argv[4]) : 20;
    float* in; unsigned* bad;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&bad, 128 * 4));
    std::vector<float> h(4096, 1.0f);
    CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 128 * 4));
    const int blocks = 256 * 8 * 24, rounds = 64;
    for (int l = 0; l < launches; ++l) {
#define LAUNCH(F) case F: hipLaunchKernelGGL(probe<F>, dim3(blocks), dim3(64), 0, 0, in, bad, rounds, mb, tb, 256u); break;
        switch (form) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15) LAUNCH(16) LAUNCH(17) default: hipLaunchKernelGGL(probe<18>, dim3(blocks), dim3(64), 0, 0, in, bad, rounds, mb, tb, 256u); }
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hb(128);
    CK(hipMemcpy(hb.data(), bad, 128 * 4, hipMemcpyDeviceToHost));
    const double tests = (double)launches * blocks * 64.0 * (rounds / 2) * tb;
    printf("form %d, mfma burst %d, test burst %d, %d launches: %u wrong of %.3g lane-results (%u of them exactly 0.0); by lane:", form, mb, tb, launches, hb[0], tests, hb[1]);
    for (int i = 0; i < 64; ++i) if (hb[2 + i]) printf(" %d:%u", i, hb[2 + i]);
    printf("\n");
    return 0;
}
