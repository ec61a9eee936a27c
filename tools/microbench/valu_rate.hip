// Micro-benchmark: vector-ALU issue rate on gfx950 as a function of the instruction class and of how many waves
// share a SIMD.  Answers "what is the VALU-issue ceiling of the render kernel" (DESIGN.md section 6): is a wave64
// v_fma_f32 4 cycles of its SIMD (so two waves per SIMD saturate it) or 2 (SIMD-32: two waves overlap)?  What do
// v_pk_fma_f32 / v_exp_f32 / DPP moves / ds_swizzle cost, alone and beside v_mfma_f32_32x32x16_bf16?
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Output: per mode and waves/SIMD, shader cycles per instruction per WAVE (s_memtime around the loop, median over
// waves) and per SIMD (cycles / (instructions of all waves of that SIMD)).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Mode { FMA = 0, PK_FMA, EXP, MOV_DPP, SWIZZLE, MFMA_ONLY, MFMA_FMA8, MFMA_FMA4, MFMA_PK4, FMA_EXP_MIX, PK_MUL, CVT_PK, PERM, LDS_READ128, LDS_ADD_F32, LDS_ADD_U32, LDS_RMW_F32, N_MODES };
static const char* mode_name[N_MODES] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_mov_b32 dpp quad_perm", "ds_swizzle_b32",
                                         "mfma_32x32x16_bf16 alone", "mfma + 8 v_fma_f32", "mfma + 4 v_fma_f32", "mfma + 4 v_pk_fma_f32",
                                         "6 v_fma + 2 v_exp", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_perm_b32", "ds_read_b128",
                                         "ds_add_f32 (64 banks)", "ds_add_u32 (64 banks)", "ds_read_b32+add+ds_write_b32"};
// instructions counted per loop iteration (the "unit" whose cycles are reported)
static const int per_iter[N_MODES] = {64, 64, 64, 64, 64, 16, 16 * 9, 16 * 5, 16 * 5, 64, 64, 64, 64, 32, 32, 32, 32};

template <int MODE>
__global__ __launch_bounds__(1024) void rate(float* out, int iters, unsigned long long* cycles) {
    __shared__ float lds[4096];
    float a[16];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[i], a[i + 8]};
    const float b = 1.0001f, c = 0.5f;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 A, B;
#pragma unroll
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * (threadIdx.x & 7) + i); B[i] = (__bf16)(0.02f * i); }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned u = threadIdx.x * 2654435761u;
    const float* lp = lds + (threadIdx.x & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == FMA) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == PK_FMA) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        } else if (MODE == PK_MUL) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        } else if (MODE == EXP) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == FMA_EXP_MIX) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[6]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[7]));
            }
        } else if (MODE == MOV_DPP) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        } else if (MODE == SWIZZLE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("ds_swizzle_b32 %0, %0 offset:0x80b1" : "+v"(a[i]));
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        } else if (MODE == CVT_PK) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == PERM) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(u));
        } else if (MODE == LDS_READ128) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp), "n"(i * 1024));
                    asm volatile("" :: "v"(v));
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        } else if (MODE == LDS_ADD_F32 || MODE == LDS_ADD_U32) {
            // lane = consecutive dword (conflict-free), rows differ per instruction and per wave
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const unsigned addr = (unsigned)(size_t)(lds + ((threadIdx.x >> 6) & 3) * 1024 + (i & 15) * 64 + (threadIdx.x & 63));
                if (MODE == LDS_ADD_F32) asm volatile("ds_add_f32 %0, %1" :: "v"(addr), "v"(a[i & 15]) : "memory");
                else asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(u) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (MODE == LDS_RMW_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const unsigned addr = (unsigned)(size_t)(lds + ((threadIdx.x >> 6) & 3) * 1024 + (i & 15) * 64 + (threadIdx.x & 63));
                float t;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
                t += a[i & 15];
                asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(t) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else {
            // 16 MFMAs per iteration on two accumulators, with k VALU fillers after each
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc0, 0, 0, 0);
                if (MODE == MFMA_FMA8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                } else if (MODE == MFMA_FMA4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                } else if (MODE == MFMA_PK4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + acc0[i] + acc1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(int cus, float* out, unsigned long long* dcyc) {
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int threads = 256 * waves_per_simd;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate<MODE>, dim3(cus), dim3(threads), 0, 0, out, iters, dcyc);    // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate<MODE>, dim3(cus), dim3(threads), 0, 0, out, iters, dcyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const int nw = cus * threads / 64;
        std::vector<unsigned long long> h(nw);
        CK(hipMemcpy(h.data(), dcyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[nw / 2];
        const double per_wave = med / ((double)iters * per_iter[MODE]);
        printf("%-28s waves/SIMD %d  cycles/instr/wave %6.2f  cycles/instr/SIMD %6.2f  (median wave %.0f cyc, kernel %.3f ms => %.2f GHz-equivalent)\n",
               mode_name[MODE], waves_per_simd, per_wave, per_wave / waves_per_simd, med, ms, med / (ms * 1e6));
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out; unsigned long long* dcyc;
    CK(hipMalloc(&out, 1024));
    CK(hipMalloc(&dcyc, sizeof(unsigned long long) * cus * 16));
    run<FMA>(cus, out, dcyc);
    run<PK_FMA>(cus, out, dcyc);
    run<PK_MUL>(cus, out, dcyc);
    run<EXP>(cus, out, dcyc);
    run<FMA_EXP_MIX>(cus, out, dcyc);
    run<MOV_DPP>(cus, out, dcyc);
    run<SWIZZLE>(cus, out, dcyc);
    run<CVT_PK>(cus, out, dcyc);
    run<PERM>(cus, out, dcyc);
    run<LDS_READ128>(cus, out, dcyc);
    run<LDS_ADD_F32>(cus, out, dcyc);
    run<LDS_ADD_U32>(cus, out, dcyc);
    run<LDS_RMW_F32>(cus, out, dcyc);
    run<MFMA_ONLY>(cus, out, dcyc);
    run<MFMA_FMA4>(cus, out, dcyc);
    run<MFMA_FMA8>(cus, out, dcyc);
    run<MFMA_PK4>(cus, out, dcyc);
    return 0;
}
