// VALU issue-rate probe for gfx950 (MI355X): cycles per wave64 vector instruction for the instruction mixes of the Jacobi row update
// (k_jacobi_round, csrc/stencil.hip), with one and two waves per SIMD.  Each test body is an inline-asm block repeated 8x inside a
// 2,000-iteration loop; s_memtime brackets the loop; the median over workgroups of wave 0's (t1 - t0) / instructions is printed.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/valu_probe tools/probes/valu_probe.hip     (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REP8(x) x x x x x x x x
#define DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPP_RSHR " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPP_RSHL " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

// 16 live VGPRs a[0..15] (allocated by the compiler as %0..%15), two SGPR-pair masks
#define OPERANDS                                                                                                                    \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),      \
      "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                                                 \
    : "s"(m0), "s"(m1)

#define BODY_INDEP4 "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %1, %5, %1\n\tv_add_f32_e32 %2, %6, %2\n\tv_add_f32_e32 %3, %7, %3\n\t"
#define BODY_DEP "v_add_f32_e32 %0, %4, %0\n\tv_add_f32_e32 %0, %5, %0\n\tv_add_f32_e32 %0, %6, %0\n\tv_add_f32_e32 %0, %7, %0\n\t"
#define BODY_DPP_WAVE "v_add_f32_dpp %0, %4, %0" DPP_SHR "v_add_f32_dpp %1, %5, %1" DPP_SHL "v_add_f32_dpp %2, %6, %2" DPP_SHR "v_add_f32_dpp %3, %7, %3" DPP_SHL
#define BODY_DPP_ROW "v_add_f32_dpp %0, %4, %0" DPP_RSHR "v_add_f32_dpp %1, %5, %1" DPP_RSHL "v_add_f32_dpp %2, %6, %2" DPP_RSHR "v_add_f32_dpp %3, %7, %3" DPP_RSHL
#define BODY_CNDMASK "v_cndmask_b32_e64 %0, %0, 0, %16\n\tv_cndmask_b32_e64 %1, %1, 0, %17\n\tv_cndmask_b32_e64 %2, %2, 0, %16\n\tv_cndmask_b32_e64 %3, %3, 0, %17\n\t"
#define BODY_MUL "v_mul_f32_e32 %0, 0x3e800000, %0\n\tv_mul_f32_e32 %1, 0x3e800000, %1\n\tv_mul_f32_e32 %2, 0x3e800000, %2\n\tv_mul_f32_e32 %3, 0x3e800000, %3\n\t"
#define BODY_FMA "v_fma_f32 %0, %4, %8, %0\n\tv_fma_f32 %1, %5, %9, %1\n\tv_fma_f32 %2, %6, %10, %2\n\tv_fma_f32 %3, %7, %11, %3\n\t"
#define BODY_PKADD "v_pk_add_f32 %[0:1], %[4:5], %[0:1]\n\t"
// the Jacobi row: t = %0..%3, up = %4..7, cur = %8..11, dn/dv = %12..15 (dn and dv share registers here: same instruction count)
#define ROW(S1, S2)                                                                                                                 \
    "v_add_f32_e32 %0, %4, %12\n\tv_add_f32_e32 %1, %5, %13\n\tv_add_f32_e32 %2, %6, %14\n\tv_add_f32_e32 %3, %7, %15\n\t"          \
    "v_add_f32_dpp %0, %11, %0" S1 "v_add_f32_e32 %1, %8, %1\n\tv_add_f32_e32 %2, %9, %2\n\tv_add_f32_e32 %3, %10, %3\n\t"          \
    "v_add_f32_e32 %0, %9, %0\n\tv_add_f32_e32 %1, %10, %1\n\tv_add_f32_e32 %2, %11, %2\n\tv_add_f32_dpp %3, %8, %3" S2             \
    "v_sub_f32_e32 %0, %0, %12\n\tv_sub_f32_e32 %1, %1, %13\n\tv_sub_f32_e32 %2, %2, %14\n\tv_sub_f32_e32 %3, %3, %15\n\t"          \
    "v_mul_f32_e32 %0, 0x3e800000, %0\n\tv_mul_f32_e32 %1, 0x3e800000, %1\n\tv_mul_f32_e32 %2, 0x3e800000, %2\n\t"                  \
    "v_mul_f32_e32 %3, 0x3e800000, %3\n\tv_cndmask_b32_e64 %0, %0, 0, %16\n\tv_cndmask_b32_e64 %3, %3, 0, %17\n\t"
#define BODY_ROW_WAVE ROW(DPP_SHR, DPP_SHL)
#define BODY_ROW_ROW ROW(DPP_RSHR, DPP_RSHL)
#define BODY_ROW_NODPP                                                                                                              \
    "v_add_f32_e32 %0, %4, %12\n\tv_add_f32_e32 %1, %5, %13\n\tv_add_f32_e32 %2, %6, %14\n\tv_add_f32_e32 %3, %7, %15\n\t"          \
    "v_add_f32_e32 %0, %11, %0\n\tv_add_f32_e32 %1, %8, %1\n\tv_add_f32_e32 %2, %9, %2\n\tv_add_f32_e32 %3, %10, %3\n\t"            \
    "v_add_f32_e32 %0, %9, %0\n\tv_add_f32_e32 %1, %10, %1\n\tv_add_f32_e32 %2, %11, %2\n\tv_add_f32_e32 %3, %8, %3\n\t"            \
    "v_sub_f32_e32 %0, %0, %12\n\tv_sub_f32_e32 %1, %1, %13\n\tv_sub_f32_e32 %2, %2, %14\n\tv_sub_f32_e32 %3, %3, %15\n\t"          \
    "v_mul_f32_e32 %0, 0x3e800000, %0\n\tv_mul_f32_e32 %1, 0x3e800000, %1\n\tv_mul_f32_e32 %2, 0x3e800000, %2\n\t"                  \
    "v_mul_f32_e32 %3, 0x3e800000, %3\n\tv_cndmask_b32_e64 %0, %0, 0, %16\n\tv_cndmask_b32_e64 %3, %3, 0, %17\n\t"

#define DEFINE_TEST(NAME, BODY)                                                                                                     \
    __global__ void NAME(float *out, long long *ticks, int iters) {                                                                 \
        float a[16];                                                                                                                \
        for (int i = 0; i < 16; ++i) a[i] = 1.0f + 0.001f * (threadIdx.x + i);                                                      \
        const unsigned long long m0 = 1ull, m1 = 1ull << 63;                                                                        \
        __syncthreads();                                                                                                            \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                                          \
        for (int it = 0; it < iters; ++it) { asm volatile(REP8(BODY) OPERANDS); }                                                   \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                                          \
        float s = 0.f;                                                                                                              \
        for (int i = 0; i < 16; ++i) s += a[i];                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                             \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                           \
    }

DEFINE_TEST(t_indep4, BODY_INDEP4)
DEFINE_TEST(t_dep, BODY_DEP)
DEFINE_TEST(t_dpp_wave, BODY_DPP_WAVE)
DEFINE_TEST(t_dpp_row, BODY_DPP_ROW)
DEFINE_TEST(t_cndmask, BODY_CNDMASK)
DEFINE_TEST(t_mul, BODY_MUL)
DEFINE_TEST(t_fma, BODY_FMA)
DEFINE_TEST(t_row_wave, BODY_ROW_WAVE)
DEFINE_TEST(t_row_row, BODY_ROW_ROW)
DEFINE_TEST(t_row_nodpp, BODY_ROW_NODPP)

typedef void (*kern_t)(float *, long long *, int);

int main() {
    struct T { const char *name; kern_t k; int n; } tests[] = {
        {"4 independent v_add_f32", t_indep4, 4}, {"dependent v_add_f32 chain", t_dep, 4}, {"v_add_f32_dpp wave_shr/shl", t_dpp_wave, 4},
        {"v_add_f32_dpp row_shr/shl", t_dpp_row, 4}, {"v_cndmask_b32_e64 (SGPR mask)", t_cndmask, 4}, {"v_mul_f32 literal", t_mul, 4},
        {"v_fma_f32 (4 indep)", t_fma, 4}, {"Jacobi row, wave-shift DPP (22)", t_row_wave, 22}, {"Jacobi row, row-shift DPP (22)", t_row_row, 22},
        {"Jacobi row, no DPP (22)", t_row_nodpp, 22}};
    float *out; long long *ticks;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&ticks, 256 * 16 * sizeof(long long));
    const int iters = 2000;
    printf("%-36s %10s %10s %10s   (cycles per wave-instruction, as seen by one wave; per SIMD = that / waves per SIMD)\n", "test", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto &t : tests) {
        printf("%-36s", t.name);
        for (int threads : {256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
            hipDeviceSynchronize();
            const int nw = 256 * threads / 64;
            std::vector<long long> h(nw);
            hipMemcpy(h.data(), ticks, nw * sizeof(long long), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf(" %10.2f", (double)h[nw / 2] / ((double)iters * 8 * t.n));
        }
        printf("\n");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
