/*
 * encoder_fast.c -- TEST / BENCH INFRASTRUCTURE ONLY (part of the CPU oracle; never linked into libsmokehip.so).
 *
 * Timing-grade CPU port of SmokePhysNet.input_encoder + both adaptive pools
 * (/root/reference/src/models/smokephys_net.py:24-32,87-91): Conv2d(1,64,7,p3)+BN+ReLU -> Conv2d(64,128,3,p1)+BN+ReLU ->
 * adaptive_avg_pool2d(D,D) -> adaptive_avg_pool2d(32,32), eval-mode BatchNorm.  It is what bench.py's `cpu_baseline` times beside the
 * GPU path (the reference's torch ops on all host cores, and on one), so it is written the way a CPU conv library works:
 *   - fp32 accumulation with fused multiply-add where the host has it (runtime dispatch: AVX-512 / AVX2+FMA / baseline clones of the
 *     micro-kernel via target_clones -- the .so is built in one container and runs on another machine's host cores);
 *   - a register-blocked micro-kernel: 4 output channels x a 64-pixel strip of one output row stay in vector registers across the
 *     whole (c_in, ki, kj) reduction (16 zmm accumulators), the input is zero-padded once so the taps need no edge branches;
 *   - OpenMP over (output-channel block, row) tasks: 16 x H (conv1) and 32 x H (conv2) tasks -- enough for a 256-thread host, which a
 *     loop over the 64 / 128 output channels alone is not.
 * Same maths as so_encoder_frame (smoke_oracle.c: fp64 accumulation, the parity oracle); tests/test_oracle_golden.py holds this port to it
 * (<= 2e-5 max-norm: summation order and FMA only).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SO_API __attribute__((visibility("default")))
#define STRIP 64
#define CB 4

/* out[c][0..n) = relu((acc + bias) folded with BN) for CB channels of one row strip: the whole reduction in registers */
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
static void conv_strip(const float *inp, int Cin, int Hp, int Wp, int K, const float *wgt /* [Cout][Cin][K][K] */, int co0, int i, int j0,
                       int n /* <= STRIP */, const float *sc, const float *sh, float *out, int H, int W) {
    float acc[CB][STRIP];
    for (int c = 0; c < CB; ++c)
        for (int j = 0; j < STRIP; ++j) acc[c][j] = 0.0f;
    const size_t wstride = (size_t)Cin * K * K;
    for (int ci = 0; ci < Cin; ++ci)
        for (int ki = 0; ki < K; ++ki) {
            const float *src = inp + ((size_t)ci * Hp + i + ki) * Wp + j0;      /* padded input: row i+ki-P, column j0-P */
            const float *w0 = wgt + ((size_t)co0 * Cin + ci) * K * K + (size_t)ki * K;
            for (int kj = 0; kj < K; ++kj) {
                const float wa = w0[kj], wb = w0[wstride + kj], wc = w0[2 * wstride + kj], wd = w0[3 * wstride + kj];
#pragma GCC ivdep
                for (int j = 0; j < STRIP; ++j) {
                    const float x = src[j + kj];
                    acc[0][j] += wa * x;
                    acc[1][j] += wb * x;
                    acc[2][j] += wc * x;
                    acc[3][j] += wd * x;
                }
            }
        }
    for (int c = 0; c < CB; ++c) {
        float *o = out + ((size_t)(co0 + c) * H + i) * W + j0;
        const float s = sc[co0 + c], t = sh[co0 + c];
        for (int j = 0; j < n; ++j) {
            const float y = acc[c][j] * s + t;
            o[j] = y > 0.0f ? y : 0.0f;
        }
    }
}

/* zero-padded copy [C][H+2P][Wp], Wp = W + 2P + STRIP slack so a strip may read past the row's end */
static float *pad_input(const float *in, int C, int H, int W, int P, int *Hp_, int *Wp_) {
    const int Hp = H + 2 * P, Wp = W + 2 * P + STRIP;
    float *p = (float *)calloc((size_t)C * Hp * Wp, sizeof(float));
#pragma omp parallel for schedule(static)
    for (int t = 0; t < C * H; ++t) {
        const int c = t / H, i = t % H;
        memcpy(p + ((size_t)c * Hp + i + P) * Wp + P, in + ((size_t)c * H + i) * W, sizeof(float) * W);
    }
    *Hp_ = Hp; *Wp_ = Wp;
    return p;
}

static void conv_bn_relu_fast(const float *in, int Cin, int H, int W, const float *wgt, const float *bias, int Cout, int K,
                              const float *bn_w, const float *bn_b, const float *bn_mean, const float *bn_var, float *out) {
    int Hp, Wp;
    float *inp = pad_input(in, Cin, H, W, K / 2, &Hp, &Wp);
    float *sc = (float *)malloc(sizeof(float) * 2 * Cout), *sh = sc + Cout;
    for (int co = 0; co < Cout; ++co) {
        const float inv = 1.0f / sqrtf(bn_var[co] + 1e-5f);
        sc[co] = inv * bn_w[co];
        sh[co] = (bias[co] - bn_mean[co]) * sc[co] + bn_b[co];
    }
    const int ntask = (Cout / CB) * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < ntask; ++t) {
        const int co0 = (t / H) * CB, i = t % H;
        for (int j0 = 0; j0 < W; j0 += STRIP)
            conv_strip(inp, Cin, Hp, Wp, K, wgt, co0, i, j0, W - j0 < STRIP ? W - j0 : STRIP, sc, sh, out, H, W);
    }
    free(sc);
    free(inp);
}

/* F.adaptive_avg_pool2d, window [floor(o*I/O), ceil((o+1)*I/O)), fp32 sums, OpenMP over channels */
static void adaptive_pool_fast(const float *in, int C, int H, int W, int OH, int OW, float *out) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (int oi = 0; oi < OH; ++oi) {
            const int i0 = (int)((int64_t)oi * H / OH), i1 = (int)(((int64_t)(oi + 1) * H + OH - 1) / OH);
            for (int oj = 0; oj < OW; ++oj) {
                const int j0 = (int)((int64_t)oj * W / OW), j1 = (int)(((int64_t)(oj + 1) * W + OW - 1) / OW);
                float s = 0.0f;
                for (int i = i0; i < i1; ++i)
                    for (int j = j0; j < j1; ++j) s += in[((size_t)c * H + i) * W + j];
                out[((size_t)c * OH + oi) * OW + oj] = s / (float)((i1 - i0) * (j1 - j0));
            }
        }
}

/* One frame [H][W] -> features [128][32][32]; parameters in state_dict order (smokephys_net.py:24-32). */
SO_API void so_encoder_frame_fast(const float *frame, int H, int W, int input_dim,
                                  const float *c1w, const float *c1b, const float *bn1w, const float *bn1b,
                                  const float *bn1m, const float *bn1v,
                                  const float *c2w, const float *c2b, const float *bn2w, const float *bn2b,
                                  const float *bn2m, const float *bn2v, float *features) {
    float *a1 = (float *)malloc(sizeof(float) * 64 * (size_t)H * W);
    float *a2 = (float *)malloc(sizeof(float) * 128 * (size_t)H * W);
    float *pl = (float *)malloc(sizeof(float) * 128 * (size_t)input_dim * input_dim);
    conv_bn_relu_fast(frame, 1, H, W, c1w, c1b, 64, 7, bn1w, bn1b, bn1m, bn1v, a1);
    conv_bn_relu_fast(a1, 64, H, W, c2w, c2b, 128, 3, bn2w, bn2b, bn2m, bn2v, a2);
    adaptive_pool_fast(a2, 128, H, W, input_dim, input_dim, pl);
    adaptive_pool_fast(pl, 128, input_dim, input_dim, 32, 32, features);
    free(a1); free(a2); free(pl);
}

/* which micro-kernel clone this host runs (reported by bench.py beside the baseline) */
SO_API const char *so_encoder_fast_isa(void) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return "avx512f";
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return "avx2+fma";
    return "sse2";
}
