// First slice of BASELINE configs[4]'s "MFMA conv3d encoder" (SPEC_3D.md section 8): Conv3d as an explicit GEMM -- the patches of a slab of
// planes are gathered into a [voxels][taps x channels] matrix (k_im2col3d) that the split-bf16 MFMA linear kernel (linear.hip) multiplies
// with the folded weights; k_pool3d_accum reduces the activated slab into the [32 x 32][C] token sums.  No reference counterpart (the
// reference's encoder is Conv2d: smokephys_net.py:24-32); the 2-D rules it generalises are cited in SPEC_3D.md.
#pragma once
#include "common.h"

namespace smk {

// src [D][H][W][C] (channels-last; C = 1: the plain volume), zero padding ksize / 2; rows z0 .. z0+nz-1 of the output:
// cols [(z - z0) H W + y W + x][kpad], column (tap * C + c), tap = (kz * ksize + ky) * ksize + kx; columns >= ksize^3 C are zero.
hipError_t launch_im2col3d(const float *src, int C, int D, int H, int W, int ksize, int z0, int nz, float *cols, int kpad, hipStream_t st);
// act [nz][H][W][C] -> sums [32 * 32][C] += sum over the slab's planes and the (H / 32) x (W / 32) block of each token
hipError_t launch_pool3d_accum(const float *act, int C, int H, int W, int nz, float *sums, hipStream_t st);

}  // namespace smk
