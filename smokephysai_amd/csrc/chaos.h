#pragma once
#include "common.h"

namespace smk {
hipError_t launch_chaos_stats(const float *frames, int64_t stride, int n, int H, int W, float *means, int32_t *box_counts,
                              int32_t *hist, hipStream_t st);
hipError_t launch_diff_norms(const float *frames, int64_t stride, int n_pairs, int n_cells, float *norms, hipStream_t st);
}  // namespace smk
