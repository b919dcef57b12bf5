// Chaos-statistics kernels: the integer/reduction parts of SmokeSimulator.get_chaos_features
// (/root/reference/src/physics/smoke_simulator.py:47-140), batched over n frames.
//   mean            : frame.mean() (fp64 accumulation, rounded once to fp32)
//   box counts      : scales 2,4,8,16,32 of (frame > mean)  (:89-124, the reference's Python double loop)
//   histogram       : torch.histogram(bins=256, range=(0,1)) counts (:134-135): values outside [0,1] dropped, 1.0 -> last bin
//   difference norms: ||frame[i+1] - frame[i]||_2 (:73-79), fp64 accumulation
// One 1024-thread workgroup per frame; wavefront shuffles + LDS for the reductions.
#include "chaos.h"

namespace smk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// block-wide sum of one double per thread (1024 threads = 16 waves); result valid in every thread
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];      // fixed order: deterministic
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(1024) void k_chaos_stats(const float *__restrict__ frames, int64_t stride, int H, int W,
                                                      float *__restrict__ means, int32_t *__restrict__ box_counts,
                                                      int32_t *__restrict__ hist) {
    extern __shared__ unsigned char flags[];        // level-2 box flags [bh2][bw2], reduced in place level by level
    __shared__ double red[16];
    __shared__ int lhist[256];
    __shared__ int lcount[5];
    const float *f = frames + (size_t)blockIdx.x * stride;
    const int tid = threadIdx.x, n = H * W;
    if (tid < 256) lhist[tid] = 0;
    if (tid < 5) lcount[tid] = 0;
    double s = 0.0;
    for (int k = tid; k < n; k += 1024) s += (double)f[k];
    const float mean = (float)(block_sum(s, red) / (double)n);
    if (tid == 0) means[blockIdx.x] = mean;

    // histogram + level-2 boxes
    for (int k = tid; k < n; k += 1024) {
        const float x = f[k];
        if (x >= 0.f && x <= 1.f) {
            int bin = (int)(x * 256.0f);            // exact: power-of-two scale
            bin = bin > 255 ? 255 : bin;
            atomicAdd(&lhist[bin], 1);
        }
    }
    int bh = H / 2, bw = W / 2;
    for (int k = tid; k < bh * bw; k += 1024) {
        const int bi = k / bw, bj = k - bi * bw;
        const float *p = f + (size_t)(2 * bi) * W + 2 * bj;
        const bool any = p[0] > mean || p[1] > mean || p[W] > mean || p[W + 1] > mean;
        flags[k] = any;
        if (any) atomicAdd(&lcount[0], 1);
    }
    __syncthreads();
    // levels 4, 8, 16, 32: a box has a set cell iff one of its four half-size boxes has
    int pw = bw;                                    // row pitch of the current flag level
    for (int lvl = 1; lvl < 5; ++lvl) {
        const int nh = H >> (lvl + 1), nw = W >> (lvl + 1);
        unsigned char vals[16];                     // <= 16 boxes per thread for 512^2 at level 4
        int cnt = 0;
        for (int k = tid, q = 0; k < nh * nw; k += 1024, ++q) {
            const int bi = k / nw, bj = k - bi * nw;
            const unsigned char *p = flags + (size_t)(2 * bi) * pw + 2 * bj;
            const unsigned char any = p[0] | p[1] | p[pw] | p[pw + 1];
            vals[q] = any;
            cnt += any;
        }
        __syncthreads();                            // everyone has read the previous level
        for (int k = tid, q = 0; k < nh * nw; k += 1024, ++q) flags[k] = vals[q];
        if (cnt) atomicAdd(&lcount[lvl], cnt);
        pw = nw;
        __syncthreads();
    }
    if (tid < 256) hist[(size_t)blockIdx.x * 256 + tid] = lhist[tid];
    if (tid < 5) box_counts[(size_t)blockIdx.x * 5 + tid] = lcount[tid];
}

__global__ __launch_bounds__(1024) void k_diff_norms(const float *__restrict__ frames, int64_t stride, int n_cells,
                                                     float *__restrict__ norms) {
    __shared__ double red[16];
    const float *a = frames + (size_t)blockIdx.x * stride, *b = a + stride;
    double s = 0.0;
    for (int k = threadIdx.x; k < n_cells; k += 1024) {
        const double d = (double)b[k] - (double)a[k];
        s += d * d;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) norms[blockIdx.x] = (float)sqrt(t);
}

hipError_t launch_chaos_stats(const float *frames, int64_t stride, int n, int H, int W, float *means, int32_t *box_counts,
                              int32_t *hist, hipStream_t st) {
    const size_t lds = (size_t)(H / 2) * (W / 2);
    hipLaunchKernelGGL(k_chaos_stats, dim3(n), dim3(1024), lds, st, frames, stride, H, W, means, box_counts, hist);
    return hipGetLastError();
}

hipError_t launch_diff_norms(const float *frames, int64_t stride, int n_pairs, int n_cells, float *norms, hipStream_t st) {
    hipLaunchKernelGGL(k_diff_norms, dim3(n_pairs), dim3(1024), 0, st, frames, stride, n_cells, norms);
    return hipGetLastError();
}

}  // namespace smk
