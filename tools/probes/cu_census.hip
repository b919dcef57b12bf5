// Which workgroups share a CU?  512 workgroups of 256 threads with 66 KB of LDS each (two per CU, as k_encoder_b16 runs) record
// HW_REG_HW_ID and HW_REG_XCC_ID; the host checks that the key (xcc, se, sh, cu) takes 256 distinct values, each exactly twice.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/cu_census tools/probes/cu_census.hip     (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void k_census(unsigned *out) {
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
    smem[threadIdx.x] = (unsigned char)hw;
    __syncthreads();
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);                    // stay resident until the whole grid has started
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc + smem[0] * 0;
    }
}

int main() {
    const int n = 512;
    unsigned *d;
    hipMalloc(&d, 2 * n * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 66736);
    hipLaunchKernelGGL(k_census, dim3(n), dim3(256), 66736, 0, d);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> keys, keys_cu_only;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        keys[(xcc << 8) | ((hw >> 8) & 0xff)]++;
        keys_cu_only[(hw >> 8) & 0xff]++;
    }
    int twice = 0;
    for (auto &kv : keys) twice += kv.second == 2;
    printf("workgroups %d, distinct (xcc, hw_id[15:8]) keys %zu, keys seen exactly twice %d, distinct hw_id[15:8] %zu\n", n, keys.size(), twice,
           keys_cu_only.size());
    for (int i = 0; i < 6; ++i) printf("  wg %d: hw_id 0x%08x xcc_id 0x%x\n", i, h[2 * i], h[2 * i + 1]);
    for (int i = 256; i < 260; ++i) printf("  wg %d: hw_id 0x%08x xcc_id 0x%x\n", i, h[2 * i], h[2 * i + 1]);
    return 0;
}
