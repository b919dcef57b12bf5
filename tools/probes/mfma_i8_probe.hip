// Probe of the v_mfma_i32_32x32x32_i8 operand layout on gfx950 with exact integer data (asymmetric A and B).
// Hypothesis H0: lane l (r = l&31, h = l>>5) holds A[r][k = 16h + j], B[k = 16h + j][r], j = 0..15.
// Hypothesis H1: two K=16 halves: k = 8h + j (j < 8) and k = 16 + 8h + (j - 8) (j >= 8).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const signed char *A, const signed char *B, int *C, int hyp) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    signed char a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        int kk = hyp == 0 ? 16 * h + j : (j < 8 ? 8 * h + j : 16 + 8 * h + (j - 8));
        a[j] = A[r * 32 + kk];
        b[j] = B[kk * 32 + r];
    }
    i32x4 av, bv;
    for (int q = 0; q < 4; ++q) {
        av[q] = (a[4*q] & 255) | ((a[4*q+1] & 255) << 8) | ((a[4*q+2] & 255) << 16) | ((a[4*q+3] & 255) << 24);
        bv[q] = (b[4*q] & 255) | ((b[4*q+1] & 255) << 8) | ((b[4*q+2] & 255) << 16) | ((b[4*q+3] & 255) << 24);
    }
    i32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
    for (int g = 0; g < 16; ++g) {
        int row = (g & 3) + 8 * (g >> 2) + 4 * h, col = r;      // standard 32x32 C/D map
        C[row * 32 + col] = acc[g];
    }
}

int main() {
    signed char hA[1024], hB[1024]; int hC[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = (signed char)(rand() % 255 - 127); hB[i] = (signed char)(rand() % 255 - 127); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += (int)hA[i*32+kk] * (int)hB[kk*32+j]; ref[i*32+j] = s; }
    signed char *dA, *dB; int *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
        printf("hypothesis %d: %d mismatches of 1024\n", hyp, bad);
    }
    return 0;
}
