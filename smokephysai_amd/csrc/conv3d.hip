// Conv3d encoder, explicit-GEMM first slice (see conv3d.h; SPEC_3D.md section 8).  Both kernels are pure data movement (HBM-bound):
// coalesced 16-byte accesses along the channel / tap axis.
#include "conv3d.h"

namespace smk {

// C = 1 (conv1: 7 x 7 x 7 = 343 taps of a scalar field): one thread per FOUR consecutive taps of one voxel (one 16-byte store; the index
// arithmetic once per four elements); consecutive threads = consecutive tap quads of one voxel, so a wave writes 1 KiB contiguous and
// the gathered reads of a voxel's window come from the L1 / L2
__global__ __launch_bounds__(256) void k_im2col3d_scalar(const float *__restrict__ src, int D, int H, int W, int ks, int z0, long long total4,
                                                         float *__restrict__ cols, int kpad) {
    const int taps = ks * ks * ks, P = ks / 2, per_vox = kpad / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long vox = i / per_vox;
        const int k0 = (int)(i - vox * per_vox) * 4;
        const int x = (int)(vox % W), y = (int)((vox / W) % H), z = z0 + (int)(vox / ((long long)W * H));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            v[e] = 0.f;
            if (k < taps) {
                const int kx = k % ks, ky = (k / ks) % ks, kz = k / (ks * ks);
                const int xx = x + kx - P, yy = y + ky - P, zz = z + kz - P;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D) v[e] = src[((size_t)zz * H + yy) * W + xx];
            }
        }
        *reinterpret_cast<float4 *>(cols + vox * kpad + k0) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// C % 4 == 0 (conv2: 27 taps x 64 channels): one thread per float4 of a (voxel, tap) piece -- a piece is C contiguous floats on both sides
__global__ __launch_bounds__(256) void k_im2col3d_cl(const float *__restrict__ src, int C, int D, int H, int W, int ks, int z0, long long total4,
                                                     float *__restrict__ cols, int kpad) {
    const int taps = ks * ks * ks, P = ks / 2, c4n = C / 4, per_vox = kpad / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long vox = i / per_vox;
        const int q = (int)(i - vox * per_vox), tap = q / c4n, c4 = q - tap * c4n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tap < taps) {
            const int x = (int)(vox % W), y = (int)((vox / W) % H), z = z0 + (int)(vox / ((long long)W * H));
            const int kx = tap % ks, ky = (tap / ks) % ks, kz = tap / (ks * ks);
            const int xx = x + kx - P, yy = y + ky - P, zz = z + kz - P;
            if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D)
                v = *reinterpret_cast<const float4 *>(src + (((size_t)zz * H + yy) * W + xx) * C + 4 * c4);
        }
        *reinterpret_cast<float4 *>(cols + vox * kpad + 4 * q) = v;
    }
}

hipError_t launch_im2col3d(const float *src, int C, int D, int H, int W, int ksize, int z0, int nz, float *cols, int kpad, hipStream_t st) {
    const long long vox = (long long)nz * H * W;
    if (C == 1) {
        const long long total4 = vox * (kpad / 4);
        const int blocks = (int)((total4 + 255) / 256 < (1 << 20) ? (total4 + 255) / 256 : (1 << 20));
        hipLaunchKernelGGL(k_im2col3d_scalar, dim3(blocks), dim3(256), 0, st, src, D, H, W, ksize, z0, total4, cols, kpad);
    } else {
        const long long total4 = vox * (kpad / 4);
        const int blocks = (int)((total4 + 255) / 256 < (1 << 20) ? (total4 + 255) / 256 : (1 << 20));
        hipLaunchKernelGGL(k_im2col3d_cl, dim3(blocks), dim3(256), 0, st, src, C, D, H, W, ksize, z0, total4, cols, kpad);
    }
    return hipGetLastError();
}

// one workgroup per token (ty, tx) of the 32 x 32 grid; a thread owns four consecutive channels (16-byte loads) of every G-th voxel of the
// token's block, G = 256 / (C / 4) voxel groups per workgroup; the groups' partial sums are added in group order through LDS, and a thread
// walks its voxels in a fixed order (plane, row, column): the result is deterministic.  Launches on one stream accumulate slab after slab.
__global__ __launch_bounds__(256) void k_pool3d_accum4(const float *__restrict__ act, int C, int H, int W, int nz, float *__restrict__ sums) {
    __shared__ float4 part[256];
    const int tok = blockIdx.x, ty = tok / 32, tx = tok % 32, bh = H / 32, bw = W / 32;
    const int c4n = C / 4, G = 256 / c4n, c4 = threadIdx.x % c4n, grp = threadIdx.x / c4n;
    const int nvox = nz * bh * bw;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp < G)
        for (int q = grp; q < nvox; q += G) {
            const int j = q % bw, i = (q / bw) % bh, z = q / (bw * bh);
            const float4 v = *reinterpret_cast<const float4 *>(act + (((size_t)z * H + ty * bh + i) * W + tx * bw + j) * C + 4 * c4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0) {
        for (int k = 1; k < G; ++k) {
            const float4 o = part[k * c4n + c4];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        float4 *dst = reinterpret_cast<float4 *>(sums + (size_t)tok * C + 4 * c4);
        float4 cur = *dst;
        cur.x += s.x; cur.y += s.y; cur.z += s.z; cur.w += s.w;
        *dst = cur;
    }
}

// generic channel counts: one thread per channel
__global__ void k_pool3d_accum(const float *__restrict__ act, int C, int H, int W, int nz, float *__restrict__ sums) {
    const int tok = blockIdx.x, ty = tok / 32, tx = tok % 32, bh = H / 32, bw = W / 32;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z)
            for (int i = 0; i < bh; ++i)
                for (int j = 0; j < bw; ++j)
                    s += act[(((size_t)z * H + ty * bh + i) * W + tx * bw + j) * C + c];
        sums[(size_t)tok * C + c] += s;
    }
}

hipError_t launch_pool3d_accum(const float *act, int C, int H, int W, int nz, float *sums, hipStream_t st) {
    if (C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0)
        hipLaunchKernelGGL(k_pool3d_accum4, dim3(1024), dim3(256), 0, st, act, C, H, W, nz, sums);
    else
        hipLaunchKernelGGL(k_pool3d_accum, dim3(1024), dim3(C < 256 ? C : 256), 0, st, act, C, H, W, nz, sums);
    return hipGetLastError();
}

}  // namespace smk
