// Lane semantics of ds_read_b64_tr_b16 (gfx950): LDS holds a [32 rows][16 cols] bf16 image with value 16 * row + col; every 16-lane group g reads
// the 4 x 16 block of rows 8g .. 8g+3 (lane 4q+p supplies the address of row q, columns 4p .. 4p+3).  Expected: lane i of the group receives
// column i of the four rows, row q in element q.   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/tr_probe tools/probes/tr_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
    __shared__ __attribute__((aligned(16))) __bf16 img[32 * 16];
    for (int i = threadIdx.x; i < 32 * 16; i += 64) img[i] = (__bf16)(float)i;      // 16 * row + col (exact in bf16 up to 256; beyond: rounded)
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const __bf16 *addr = img + (8 * g + q) * 16 + 4 * p;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3))) *)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int g = lane >> 4, i = lane & 15;
            const float want = (float)(__bf16)(float)(16 * (8 * g + e) + i);
            if (h[lane * 4 + e] != want) ++bad;
        }
    printf("mismatches vs (row 8g+e, col lane&15): %d\n", bad);
    for (int lane = 0; lane < 20; lane += 3) printf("lane %2d: %g %g %g %g\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    return 0;
}
