#pragma once
#include "common.h"

namespace smk {

// Training-mode BatchNorm2d (batch statistics) + ReLU + P x P mean pooling over an NCHW fp32 tensor -- the encoder's
// norm / activation / pool blocks under autograd (smokephys_net.py:24-32,87-91; train.py:88-89).
struct BnTrainArgs {
    const float *z;              // [B][C][H][W] convolution output
    const float *gamma, *beta;   // [C]
    int B, C, H, W, pool;        // pool in {1, 4, 8}; pool > 1 needs W == 32 * pool and H % pool == 0
    float eps;
    float *part;                 // workspace: [C][nchunks][2] partial sums (bn_train_workspace_floats)
    // forward
    float *out;                  // [B][C][H/pool][W/pool]
    float *mean, *var, *rstd;    // [C]: batch mean, biased batch variance, 1 / sqrt(var + eps)
    // backward
    const float *dout;           // [B][C][H/pool][W/pool]
    float *dz;                   // [B][C][H][W]
    float *dgamma, *dbeta;       // [C]
    // cross-rank statistics (SyncBatchNorm): element count of the GLOBAL batch per channel for the dz formula; 0 = B * H * W
    float count;
};
long long bn_train_workspace_floats(int B, int C, int H, int W, int pool);
hipError_t launch_bn_relu_pool_forward(const BnTrainArgs &a, hipStream_t st);
hipError_t launch_bn_relu_pool_backward(const BnTrainArgs &a, hipStream_t st);
// The same passes one at a time, for statistics that span several processes (the caller all-reduces between them):
//   stats:  z -> mean / var / rstd of THIS process's batch;   apply: out from given mean / rstd;
//   sums:   dout, z, given mean / rstd -> dgamma / dbeta of this process;   dz: from given TOTAL dgamma / dbeta and a.count.
hipError_t launch_bn_stats(const BnTrainArgs &a, hipStream_t st);
hipError_t launch_bn_relu_pool_apply(const BnTrainArgs &a, hipStream_t st);
hipError_t launch_bn_relu_pool_backward_sums(const BnTrainArgs &a, hipStream_t st);
hipError_t launch_bn_relu_pool_backward_dz(const BnTrainArgs &a, hipStream_t st);

}  // namespace smk
