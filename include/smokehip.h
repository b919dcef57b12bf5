/*
 * smokehip.h -- C ABI of libsmokehip.so: the MI355X (gfx950) implementation of the SmokePhysAI hot path.
 *
 * The reference (MengAiDev/SmokePhysAI) has no FFI: its boundary for this path is the Python class API
 *   src/physics/navier_stokes.py:6-173   NavierStokesSimulator
 *   src/physics/smoke_simulator.py:8-45  SmokeSimulator.{add_incense_source,simulate_step}
 *   src/physics/fractal_generator.py:5-62 FractalGenerator
 *   src/models/smokephys_net.py:24-32,87-91  SmokePhysNet.input_encoder (+ pooling)
 * Each entry point below names the reference interface it replaces.  Callers are the drop-in Python
 * classes in smokephysai_amd/ (ctypes); INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 = smk_status; smk_last_error() gives a thread-local message;
 *  - all device memory for state/frames/weights/features is CALLER-OWNED (torch tensors): the library never
 *    frees or reallocates it; library-owned scratch lives in the handle and dies with smk_sim_destroy;
 *  - all work is enqueued on the caller-supplied hipStream_t (passed as void*), no implicit sync;
 *  - a handle is not thread-safe; distinct handles are independent;
 *  - fields are fp32, batch-major [B][rows][pitch] with explicit row pitches (in floats):
 *        u [B][H+1][pitch_c]  v [B][H][pitch_v]  p,density [B][H][pitch_c]     pitch_c >= W, pitch_v >= W+1
 *    (the reference's shapes are u[H+1,W], v[H,W+1], p/density[H,W]: navier_stokes.py:27-32).
 */
#ifndef SMOKEHIP_H
#define SMOKEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMK_ABI_VERSION 17

typedef enum smk_status {
    SMK_OK = 0,
    SMK_ERR_INVALID = -1,      /* bad argument / shape */
    SMK_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    SMK_ERR_UNSUPPORTED = -3,  /* valid in the reference, not built here (message says what) */
    SMK_ERR_NO_DEVICE = -4,
    SMK_ERR_TIMEOUT = -5       /* a bounded wait inside a persistent launch ran out: results were poisoned (NaN), see smk_sim_status */
} smk_status;

int smk_abi_version(void);
const char *smk_last_error(void);

/* ------------------------------------------------------------------ simulation state */
typedef struct smk_sim smk_sim; /* opaque: one BATCH of independent grids */

typedef struct smk_sim_desc {
    int32_t batch, height, width;   /* B grids of H x W (reference: grid_size=(H,W), B=1) */
    int32_t jacobi_iters;           /* reference hard-codes 20 (navier_stokes.py:139) */
    double dt, viscosity;           /* doubles: the reference multiplies python floats before casting */
    int32_t device_id;
    int32_t pitch_c, pitch_v;       /* row pitches in floats */
    float *u, *v, *p, *density;     /* caller-owned device pointers, layout above */
} smk_sim_desc;

/* NavierStokesSimulator.__init__ (navier_stokes.py:9-22): binds the caller's state tensors, allocates
 * scratch (ping-pong fields, divergence), computes the shape-only fractal constant on the device. Does NOT
 * zero the state: call smk_sim_reset. */
int smk_sim_create(const smk_sim_desc *desc, smk_sim **out);
/* Frees the handle (always).  Returns SMK_ERR_TIMEOUT if a persistent projection of this handle timed out and nobody has been told yet
 * (the frames the caller read last were NaN); SMK_OK otherwise. */
int smk_sim_destroy(smk_sim *sim);

/* Health of the handle -- no reference counterpart (the reference is synchronous: navier_stokes.py:133-149 either returns right results or
 * raises).  The persistent projection (see smk_sim_step) bounds every inter-workgroup wait; when one runs out the kernel turns the band's
 * p / u / v into NaN (so every later frame of the grid is NaN: wrong results cannot pass as data) and sets a host-visible word.  This call
 * reads and acknowledges that word: SMK_ERR_TIMEOUT exactly once per event, SMK_OK otherwise; it does not synchronise -- call it after
 * the stream the steps ran on has been synchronised to learn about THOSE steps (the Python mirror's check() does both).  Every other
 * smk_sim_* entry point performs the same check on entry, so an event is never reported later than the next call on the handle. */
int smk_sim_status(smk_sim *sim);

/* NavierStokesSimulator.setup_grid (navier_stokes.py:24-35): zero u,v,p,density of the grids whose byte in
 * grid_mask (host, B bytes) is non-zero; NULL = all grids.  If a persistent-projection time-out is pending (smk_sim_status), this call
 * returns SMK_ERR_TIMEOUT once like any other entry point -- but it has carried out the reset, so it need not be repeated. */
int smk_sim_reset(smk_sim *sim, const uint8_t *grid_mask, void *stream);

typedef struct smk_source {
    int32_t grid;       /* batch index */
    int32_t x, y;       /* (column, row), as add_smoke_source(x, y, ...) */
    int32_t radius;
    double intensity;
} smk_source;

/* NavierStokesSimulator.add_smoke_source (navier_stokes.py:37-48), n sources in one launch; sources of the
 * same grid are applied in list order (fp32 += is order-sensitive). `sources` is a HOST array. */
int smk_sim_add_sources(smk_sim *sim, const smk_source *sources, int32_t n, void *stream);

/* NavierStokesSimulator.step x n_steps (navier_stokes.py:151-173) + SmokeSimulator.simulate_step's frame emit
 * (smoke_simulator.py:31-39).  After step t (0-based) the emitted frame of grid b is written to
 *     frames + t*frame_stride_t + b*frame_stride_b   as [H][W] contiguous fp32      (frames may be NULL)
 * with frame = density + (fractal_intensity*F)*density when add_fractal != 0 (fractal_generator.py:53-62, F the
 * shape-only constant), else frame = density.  Solver state keeps the unperturbed density.
 * Launches per time step: two where the band plan allows (buoyancy + diffusion + the whole pressure projection as one persistent
 * launch whose workgroups hand halo rows to each other with bounded waits, then the three advections as one launch).  The persistent
 * launch assumes this process has the device to itself (all its workgroups resident at once): if a wait times out (0.5 s) the
 * launch still drains, the affected grids' p / u / v / density / frames become NaN, and smk_sim_status after a stream synchronise -- or
 * at the latest the NEXT call on this handle, smk_sim_destroy included -- returns SMK_ERR_TIMEOUT with a message naming the persistent
 * projection (then: smk_sim_reset; later steps use one launch per chunk of sweeps).  SMK_JACOBI_PERSIST=0 in the environment selects that
 * form from the start; stream capture always records it. */
int smk_sim_step(smk_sim *sim, int32_t n_steps, float *frames, int64_t frame_stride_b, int64_t frame_stride_t,
                 int32_t add_fractal, double fractal_intensity, void *stream);

/* Single stages of step() on the handle's state, for per-stage parity tests.  State after each call is
 * back in the caller's tensors. */
typedef enum smk_stage {
    SMK_STAGE_BUOY_DIFFUSE = 0, /* navier_stokes.py:154-160 */
    SMK_STAGE_PROJECT = 1,      /* navier_stokes.py:133-149 */
    SMK_STAGE_ADVECT_U = 2,     /* navier_stokes.py:166 */
    SMK_STAGE_ADVECT_V = 3,     /* navier_stokes.py:167 */
    SMK_STAGE_ADVECT_D = 4      /* navier_stokes.py:168-171 (advect density + 0.995 decay) */
} smk_stage;
int smk_sim_run_stage(smk_sim *sim, int32_t stage, void *stream);

/* Divergence field of the current state (navier_stokes.py:136) -> out [B][H][W] contiguous. */
int smk_sim_divergence(smk_sim *sim, float *out, void *stream);

/* Back-trace gather indices of advection_step(field; u, v) on the current state (navier_stokes.py:87-92,
 * 115-123): which = 0 (field=u), 1 (field=v), 2 (field=density). x0,y0: int32 [B][R][C] contiguous. */
int smk_sim_backtrace(smk_sim *sim, int32_t which, int32_t *x0, int32_t *y0, void *stream);

/* Device pointer to the fractal constants, [W][H] fp32 each (fractal_generator.py:12-51):
 * kind 0 = perlin, 1 = mandelbrot escape counts/100, 2 = 0.7*perlin+0.3*mandelbrot. Square grids only. */
int smk_sim_fractal(smk_sim *sim, int32_t kind, const float **dev_ptr);

/* Diagnostic: a JSON object (NUL-terminated, written into buf[capacity]) describing how this handle's step is launched -- the
 * projection's kernel, bands per grid, launches and sweeps per launch.  No reference counterpart (the reference runs 20 separate
 * sweeps, navier_stokes.py:139-145); bench.py reports it beside the stencil roofline. */
int smk_sim_describe(smk_sim *sim, char *buf, int64_t capacity);

/* ------------------------------------------------------------------ 3-D simulation state (BASELINE configs[4])
 * No reference counterpart: the reference is 2-D only (navier_stokes.py:10,21).  Semantics = SPEC_3D.md, the rule-by-rule generalisation
 * of NavierStokesSimulator (navier_stokes.py:24-173) to grids [D][H][W]; each entry point names the 2-D method it generalises.
 * Fields, batch-major with explicit row pitches (floats):
 *     u [B][D][H+1][pitch_c]   v [B][D][H][pitch_v]   w [B][D+1][H][pitch_c]   p, density [B][D][H][pitch_c]
 * Conventions as for smk_sim: caller-owned state, library-owned scratch, caller's stream, no implicit sync. */
typedef struct smk_sim3d smk_sim3d;

typedef struct smk_sim3d_desc {
    int32_t batch, depth, height, width;
    int32_t jacobi_iters;           /* SPEC_3D.md section 4: default 20, like navier_stokes.py:139 */
    double dt, viscosity;
    int32_t device_id;
    int32_t pitch_c, pitch_v;
    float *u, *v, *w, *p, *density;
} smk_sim3d_desc;

typedef struct smk_source3d {
    int32_t grid;
    int32_t x, y, z;    /* (column, row, plane): add_smoke_source(x, y, ...) of navier_stokes.py:37 with the depth index appended */
    int32_t radius;
    double intensity;
} smk_source3d;

/* __init__ (navier_stokes.py:9-22) / setup_grid (:24-35) / add_smoke_source (:37-48) in 3-D. */
int smk_sim3d_create(const smk_sim3d_desc *desc, smk_sim3d **out);
int smk_sim3d_destroy(smk_sim3d *sim);
int smk_sim3d_reset(smk_sim3d *sim, const uint8_t *grid_mask, void *stream);
int smk_sim3d_add_sources(smk_sim3d *sim, const smk_source3d *sources, int32_t n, void *stream);

/* step() x n_steps (navier_stokes.py:151-173 -> SPEC_3D.md section 6).  After step t the density of grid b (the returned frame) is
 * written to frames + t*frame_stride_t + b*frame_stride_b as [D][H][W] contiguous fp32 (frames may be NULL). */
int smk_sim3d_step(smk_sim3d *sim, int32_t n_steps, float *frames, int64_t frame_stride_b, int64_t frame_stride_t, void *stream);

/* Single stages of the 3-D step on the handle's state (per-stage parity tests); state is back in the caller's tensors afterwards. */
typedef enum smk_stage3d {
    SMK_STAGE3D_BUOY_DIFFUSE = 0, /* SPEC_3D.md 6.1-6.2 */
    SMK_STAGE3D_PROJECT = 1,      /* section 4 */
    SMK_STAGE3D_ADVECT_U = 2,
    SMK_STAGE3D_ADVECT_V = 3,
    SMK_STAGE3D_ADVECT_W = 4,
    SMK_STAGE3D_ADVECT_D = 5      /* density advect + 0.995 decay */
} smk_stage3d;
int smk_sim3d_run_stage(smk_sim3d *sim, int32_t stage, void *stream);

/* ------------------------------------------------------------------ 3-D encoder pieces (BASELINE configs[4]; SPEC_3D.md section 8)
 * No reference counterpart (the reference's input_encoder is Conv2d: smokephys_net.py:24-32).  Conv3d runs as an explicit GEMM on the
 * split-bf16 MFMA linear kernel (smk_linear_forward): these two entry points are the data movement around it.
 * smk_conv3d_im2col: src [D][H][W][C] fp32 (channels-last; C = 1 is the plain volume; C = 1 or a multiple of 4), zero padding ksize/2
 *   (ksize odd), output rows for the planes z0 .. z0+nz-1: cols [nz*H*W][kpad], column tap*C + c with tap = (kz*ksize + ky)*ksize + kx,
 *   columns >= ksize^3 * C zero (kpad >= ksize^3 * C, a multiple of 4; the linear kernel wants a multiple of 64).
 * smk_pool3d_accumulate: act [nz][H][W][C] -> sums [32*32][C] += the sum over the slab's planes and over each token's (H/32) x (W/32)
 *   block (H, W multiples of 32): the two adaptive average pools of smokephys_net.py:87-91 with the depth axis pooled to 1, as sums --
 *   the caller divides by D * (H/32) * (W/32) once. */
int smk_conv3d_im2col(const float *src, int32_t C, int32_t D, int32_t H, int32_t W, int32_t ksize, int32_t z0, int32_t nz, float *cols,
                      int32_t kpad, void *stream);
int smk_pool3d_accumulate(const float *act, int32_t C, int32_t H, int32_t W, int32_t nz, float *sums, void *stream);

/* Stand-alone stateless ops (pure functions of the reference) -------------------------------------- */
/* NavierStokesSimulator.diffusion_step(field, viscosity) (navier_stokes.py:50-72) on [B][R][pitch]. */
int smk_diffuse(const float *in, float *out, int32_t B, int32_t R, int32_t C, int32_t pitch, double dt,
                double viscosity, void *stream);
/* FractalGenerator.apply_fractal_perturbation(field, intensity) (fractal_generator.py:53-62) on n_fields
 * square [N][N] contiguous fields; computes the constant itself (cached per N per device). */
int smk_apply_fractal(const float *in, float *out, int32_t n_fields, int32_t N, double intensity, void *stream);

/* NavierStokesSimulator.advection_step(field, u, v) (navier_stokes.py:74-95) as a pure function on caller buffers:
 * which = 0/1/2 selects the field's shape (u-like [H+1][pitch_c], v-like [H][pitch_v], cell [H][pitch_c]); no decay. */
int smk_advect(const float *field, float *out, int32_t which, const float *u, const float *v, int32_t B, int32_t H,
               int32_t W, int32_t pitch_c, int32_t pitch_v, double dt, void *stream);
/* NavierStokesSimulator.bilinear_interpolate(field, y, x) (navier_stokes.py:111-131; mode 0),
 * .interpolate_velocity_u(u, y, x) (:97-102; mode 1: x + 0.5 clamped to [0, w-1] first) and .interpolate_velocity_v(v, y, x)
 * (:104-109; mode 2: y + 0.5 clamped to [0, h-1] first) as pure gathers: B fields [h][pitch] (field_stride floats apart), n
 * coordinate pairs per field (coord_stride floats apart; 0 = one list shared by all fields), out [B][n].  Coordinates may be any
 * float (floor -> index clamp -> weights from the clamped indices, so a coordinate exactly on the upper edge yields 0). */
int smk_interpolate(int32_t mode, const float *field, int32_t B, int32_t h, int32_t w, int32_t pitch, int64_t field_stride,
                    const float *y, const float *x, int64_t coord_stride, int64_t n, float *out, void *stream);
/* FractalGenerator.generate_perlin_noise / generate_mandelbrot_field (fractal_generator.py:12-51) for an N x N grid:
 * writes [N][N] fp32 each (any pointer may be NULL): perlin, mandelbrot (counts/100), 0.7*perlin+0.3*mandelbrot. */
int smk_fractal_constants(int32_t N, float *perlin, float *mandel, float *field, void *stream);

/* SmokeSimulator.get_chaos_features' reductions (smoke_simulator.py:47-140) for n frames [H][W] (dense rows, frame
 * stride `frame_stride` floats): means [n] fp32 (frame.mean()), box_counts [n][5] int32 (scales 2,4,8,16,32 of
 * frame > mean), hist [n][256] int32 (torch.histogram(bins=256, range=(0,1)) counts). Needs (H/2)*(W/2) <= 65536. */
int smk_chaos_stats(const float *frames, int64_t frame_stride, int32_t n, int32_t H, int32_t W, float *means,
                    int32_t *box_counts, int32_t *hist, void *stream);
/* ||frame[i+1] - frame[i]||_2 for i < n-1 (smoke_simulator.py:73-79) -> norms [n-1] fp32 (fp64 accumulation). */
int smk_frame_diff_norms(const float *frames, int64_t frame_stride, int32_t n, int32_t H, int32_t W, float *norms,
                         void *stream);

/* ------------------------------------------------------------------ CNN encoder */
/* SMK_F32: fp32 MFMA; SMK_BF16X3: split-bf16 MFMA (fp32-class accuracy); SMK_BF16: single-pass bf16;
 * SMK_I8X3: 16-bit fixed point as two int8 limbs on int8 MFMA (per-tile activation scale, exact i32 accumulation). */
typedef enum smk_dtype { SMK_F32 = 0, SMK_BF16X3 = 1, SMK_BF16 = 2, SMK_I8X3 = 3 } smk_dtype;

/* Eval-mode input_encoder weights (smokephys_net.py:24-32), device pointers, PyTorch layouts:
 * conv1_w [64][1][7][7], conv2_w [128][64][3][3]; BN as (weight,bias,running_mean,running_var), eps 1e-5. */
typedef struct smk_encoder_weights {
    const float *conv1_w, *conv1_b, *bn1_w, *bn1_b, *bn1_mean, *bn1_var;
    const float *conv2_w, *conv2_b, *bn2_w, *bn2_b, *bn2_mean, *bn2_var;
} smk_encoder_weights;

typedef struct smk_encoder smk_encoder; /* opaque: folded/re-laid-out weights on the device */

int smk_encoder_create(const smk_encoder_weights *w, int32_t device_id, void *stream, smk_encoder **out);
int smk_encoder_destroy(smk_encoder *enc);

/* SmokePhysNet.input_encoder + both adaptive pools (smokephys_net.py:87-91):
 * frames [B][H][W] fp32 (row pitch W, frame stride frame_stride floats) -> features [B][128][32][32] fp32.
 * Requires H == W, H % 32 == 0, and input_dim a multiple/divisor of H (then the two pools compose to an
 * (H/32)^2 block mean); anything else returns SMK_ERR_UNSUPPORTED. */
int smk_encoder_forward(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H,
                        int32_t W, int32_t input_dim, float *features, int32_t dtype, void *stream);

/* Same computation, features written token-major: [B][32*32 tokens][128 channels] fp32 -- the layout
 * `encoded.flatten(2).transpose(1, 2)` (smokephys_net.py:95) hands to feature_proj; coalesced stores. bf16 dtypes only. */
int smk_encoder_forward_tokens(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H,
                               int32_t W, int32_t input_dim, float *tokens, int32_t dtype, void *stream);

/* conv1+BN+ReLU activations only (smokephys_net.py:25-27), [B][64][H][W] fp32 -- parity hook. */
int smk_encoder_conv1(smk_encoder *enc, const float *frames, int64_t frame_stride, int32_t B, int32_t H,
                      int32_t W, float *act, void *stream);

/* ------------------------------------------------------------------ transformer-body linear layers */
/* One nn.Linear of SmokePhysNet's token path -- feature_proj (smokephys_net.py:38,97), q/k/v/out projections
 * (chaos_attention.py:25-28,77-79,113), FFN (smokephys_net.py:153-158), output_decoder (smokephys_net.py:50-54) --
 * with its weights re-laid-out on the device for the split-bf16 MFMA kernel (fp32-class accuracy: results within
 * 1e-4 relative of the fp32 GEMM the reference runs; measured ~1e-6).
 * weight [out_features][in_features] fp32 (PyTorch layout), bias [out_features] or NULL; both are read once, on `stream`.
 * Requires in_features % 64 == 0 and out_features % 32 == 0 (else SMK_ERR_UNSUPPORTED). */
/* Activation formats between the body kernels.  SMK_FMT_SPLIT_BF16 keeps an fp32-accurate activation as two bf16 per
 * element (x = hi + lo) in the layout the MFMA kernels consume: per row, per group of 8 consecutive features, 8 hi then
 * 8 lo ([rows][features/8][2][8] bf16 -- the same 4 bytes per element as fp32, rows dense).  A producer that writes it
 * (LayerNorm, a linear layer's epilogue, the attention kernel) saves the consuming linear layer the split arithmetic in
 * its K loop.
 * SMK_FMT_SPLIT4_INPLACE (round 4) is the same pair of bf16 per element laid into the fp32 tensor's own storage: every aligned group
 * of 4 consecutive features holds {hi[0..3], lo[0..3]} in the 16 bytes of its 4 floats (any row pitch, any column range of a row).  The
 * fused q | k | v layer writes its k and v columns that way (smk_linear_forward_ln_split) and the attention kernel stages them without
 * split arithmetic (smk_attention_kv): the split is done once per element instead of once per 128-query block. */
typedef enum smk_format { SMK_FMT_F32 = 0, SMK_FMT_SPLIT_BF16 = 1, SMK_FMT_SPLIT4_INPLACE = 2 } smk_format;

typedef struct smk_linear smk_linear;
typedef enum smk_activation { SMK_ACT_NONE = 0, SMK_ACT_GELU = 1 /* erf form, nn.GELU() */, SMK_ACT_RELU = 2 } smk_activation;

int smk_linear_create(const float *weight, const float *bias, int32_t out_features, int32_t in_features,
                      int32_t device_id, void *stream, smk_linear **out);
int smk_linear_destroy(smk_linear *lin);

/* Re-split new weights into an existing handle (training: the parameters change at every optimizer step, train.py:88-93; no
 * allocation).  transposed = 0: weight is [out_features][in_features] as in smk_linear_create.  transposed = 1: weight is
 * [in_features][out_features] row-major, i.e. the handle computes x W for a PyTorch weight W whose shape is
 * [in_features][out_features] -- the input-gradient GEMM dX = dY W of nn.Linear's backward (autograd's LinearBackward0) on the
 * same kernel.  bias NULL = zeros.  Read once, on `stream`. */
int smk_linear_update(smk_linear *lin, const float *weight, int32_t transposed, const float *bias, void *stream);

/* Weight gradient of nn.Linear, dW [out_features][in_features] = dY^T X (autograd's LinearBackward0; train.py:89 `total.backward()`),
 * on the same split-bf16 kernel: dy [rows][out_features] (row pitch ld_dy floats), x [rows][in_features] (row pitch ldx), dw dense.
 * The reduction runs over the token rows, cut into segments that share one launch; the partial sums are added in segment order
 * (deterministic).  `workspace`: smk_linear_wgrad_workspace(rows, out, in) bytes of device memory, 16-byte aligned, owned by the
 * caller and free for reuse once the enqueued work has run.  Requires in_features % 32 == 0, out_features % 4 == 0,
 * (out_features + 256) * (rows + 4096) < 2^30 (longer inputs: call per row chunk and add).  db (may be NULL): the bias gradient
 * [out_features] = column sums of dy (partials per row segment, added in a fixed order).  With out_features and in_features multiples of 128
 * neither operand is copied: both are staged as they lie and read back transposed from LDS (ds_read_b64_tr_b16).  Enqueued on `stream`. */
int64_t smk_linear_wgrad_workspace(int64_t rows, int32_t out_features, int32_t in_features);
int smk_linear_wgrad(const float *dy, int64_t ld_dy, const float *x, int64_t ldx, int64_t rows, int32_t out_features,
                     int32_t in_features, float *dw, float *db, void *workspace, int64_t workspace_bytes, void *stream);

/* y = act(x W^T + b + addend) + residual over `rows` token rows:
 *   x [rows][in_features], row pitch ldx floats; y [rows][out_features], row pitch ldy (all row starts 16-byte aligned);
 *   residual (or NULL) [rows][out_features], row pitch ldr -- the `x + sublayer(x)` of the pre-LN block
 *   (smokephys_net.py:161-167); may alias y;
 *   periodic_add (or NULL) [rows / rows_per_group][period][out_features]: row i of group g receives
 *   periodic_add[g][(i % rows_per_group) % period] before the activation -- the chaos term folded into Q
 *   (the 5-step Lorenz field tiled along the sequence, chaos_attention.py:61-65,85-100); rows_per_group % 32 == 0.
 *   residual and periodic_add are mutually exclusive (no layer of the path uses both).
 *   x_format / y_format: smk_format of x and y (void pointers: fp32 or split-bf16 storage); a split tensor has dense rows
 *   (ldx == in_features, ldy == out_features); a split y excludes residual.
 * Enqueued on `stream`, no synchronisation. */
int smk_linear_forward(smk_linear *lin, const void *x, int64_t rows, int64_t ldx, void *y, int64_t ldy,
                       const float *residual, int64_t ldr, const float *periodic_add, int32_t rows_per_group,
                       int32_t period, int32_t activation, int32_t x_format, int32_t y_format, void *stream);

/* LayerNorm fused in front of the layer (nn.LayerNorm -> nn.Linear of the pre-LN block, /root/reference/src/models/smokephys_net.py:149-150,
 * 161,165): y = act(LN(x) W^T + b + periodic_add), LN over the in_features of a row with `eps`.  `lin` must have been created from the
 * FOLDED parameters W' = W diag(gamma), b' = b + W beta; wsum [out_features] = row sums of W'.  The kernel reads the raw x, gathers each
 * row's mean / variance while staging it (shifted sums about a per-thread pivot merged pairwise: no cancellation for rows whose mean
 * dwarfs their spread) and applies rstd (x W'^T - mean wsum) + b' in its epilogue: the separate LayerNorm launch and its round trip are
 * gone.  Any row count (round 3 served one tile per workgroup only; smk_linear_ln_max_rows now only reflects the 32-bit offset range).
 * fp32 in / out; no residual. */
int64_t smk_linear_ln_max_rows(smk_linear *lin);
int smk_linear_forward_ln(smk_linear *lin, const float *x, int64_t rows, int64_t ldx, float *y, int64_t ldy, const float *wsum, double eps,
                          const float *periodic_add, int32_t rows_per_group, int32_t period, int32_t activation, void *stream);
/* The same launch with the output columns >= split_from_col (a multiple of 32, below out_features; negative: none) written as
 * SMK_FMT_SPLIT4_INPLACE instead of fp32 -- the k | v columns of the fused q | k | v projection (smokephys_net.py:161,
 * chaos_attention.py:77-79), consumed by smk_attention_kv. */
int smk_linear_forward_ln_split(smk_linear *lin, const float *x, int64_t rows, int64_t ldx, float *y, int64_t ldy, const float *wsum, double eps,
                                const float *periodic_add, int32_t rows_per_group, int32_t period, int32_t activation, int32_t split_from_col,
                                void *stream);

/* Conv3d(64 -> N, kernel 3, padding 1) + bias + activation as an IMPLICIT GEMM on the split-bf16 MFMA layer kernel -- no patch matrix:
 * src [D][H][W][64] fp32 channels-last, `lin` a layer handle with in_features = 27 * 64 whose weight columns are tap * 64 + c
 * (tap = (kz*3 + ky)*3 + kx, i.e. conv.weight.permute(0, 2, 3, 4, 1).reshape(N, 1728)); output rows = the voxels of planes z0 .. z0+nz-1
 * in memory order, y [nz*H*W][ldy].  One call addresses the planes it reads with 32-bit offsets: (nz + 2) * H * W * 256 < 2^32. */
int smk_conv3d_cl_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, int32_t z0, int32_t nz, float *y, int64_t ldy,
                          int32_t activation, void *stream);
/* Conv3d(1 -> N, kernel 7, padding 3) + bias + activation of a SCALAR volume src [D][H][W] as an implicit GEMM on the same kernel (SPEC_3D.md
 * section 8, conv1): `lin` has in_features = 448 = 56 window rows x 8 kx slots, weight column (kz*7 + ky)*8 + kx (slot kx = 7 and window rows
 * >= 49 zero); output rows = the voxels of planes z0 .. z0+nz-1, y [nz*H*W][ldy].  H, W <= 1023; (nz + 6) * H * W * 4 < 2^32. */
int smk_conv3d_s7_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, int32_t z0, int32_t nz, float *y, int64_t ldy,
                          int32_t activation, void *stream);
/* The same Conv3d(64 -> 128, kernel 3, padding 1) + bias + activation (SMK_ACT_NONE or SMK_ACT_RELU) of a whole volume, fused with the DEPTH
 * half of the token pooling (SPEC_3D.md section 8 pools the depth axis to 1): zsum [H][W][128] = sum over z of act(conv(src)[z] + bias), planes
 * added in z order (deterministic).  A workgroup marches an 8 x 16 column of voxels along z with three planes of its halo tile in LDS, so
 * every voxel is staged once per plane instead of once per tap, and the activated convolution output is never written; follow with
 * smk_pool3d_accumulate(zsum, 128, H, W, 1, sums).  `lin` as for smk_conv3d_cl_forward with out_features = 128; H % 8 == 0, W % 16 == 0,
 * H * W * 256 < 2^31. */
int smk_conv3d_cl_zsum_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, float *zsum, int32_t activation, void *stream);
/* Conv3d(1 -> 64, kernel 7, padding 3) + bias + activation (SMK_ACT_NONE or SMK_ACT_RELU) of a whole scalar volume src [D][H][W] into the
 * channels-last activations a1 [D][H][W][64], marched along z: the weights stay in registers, each input plane of a workgroup's 14 x 22 halo
 * tile is expanded once into a table of ready MFMA fragments in LDS (no gather and no split arithmetic in the loop).  `lin`: in_features =
 * 448 = 7 kz x 8 ky slots x 8 kx slots, weight column (kz*8 + ky)*8 + kx (slot 7 of ky and of kx zero), out_features = 64.
 * H % 8 == 0, W % 16 == 0, H * W * 256 < 2^31. */
int smk_conv3d_s7_march_forward(smk_linear *lin, const float *src, int32_t D, int32_t H, int32_t W, float *a1, int32_t activation, void *stream);

/* smk_chaos_addend for up to 8 layers in ONE launch (a model's layers draw their noise up front; six launches of one workgroup per batch
 * element are six launch latencies at batch 1).  Each layer: its own noise [3][B], weights, output and strength; B, D and the Lorenz constants
 * are shared. */
typedef struct smk_chaos_layer {
    const float *noise, *proj_w, *proj_b, *gate_w, *gate_b;
    float *addend;
    int64_t ld_addend;
    double strength;
} smk_chaos_layer;
int smk_chaos_addend_batched(int32_t n_layers, const smk_chaos_layer *layers, int32_t B, int32_t D, double sigma, double rho, double beta,
                             double dt, void *stream);

/* SmokePhysNet's tail (smokephys_net.py:116-118): pooled [B][D] = features.mean(dim=1) of x [B][L][ldx] and out [B][H2] =
 * Linear2(ReLU(Linear1(pooled))) with w1 [H1][D], w2 [H2][H1] (row-major, nn.Linear layout), in three small launches instead of PyTorch-ROCm's
 * seven (reduce, two GEMM calls with their bias copies, ReLU, fill).  workspace: B * (32 * D + H1) floats.  fp32, fixed summation order. */
int smk_pooled_head(const float *x, int32_t B, int32_t L, int32_t D, int64_t ldx, const float *w1, const float *b1, int32_t H1,
                    const float *w2, const float *b2, int32_t H2, float *pooled, float *out, float *workspace, void *stream);

/* ChaosAttention.generate_chaos_field's five explicit-Euler Lorenz states (chaos_attention.py:39-59) for noise [3][B] (the three
 * randn(B,1) draws before the 0.1 scale): states [B][5][3].  The gradient-free part of the chaos term, for the training path. */
int smk_lorenz_states(const float *noise, int32_t B, double sigma, double rho, double beta, double dt, float *states, void *stream);

/* The element-wise chain of ChaosTransformerLayer's FFN under autograd (smokephys_net.py:153-159 Linear -> GELU -> Dropout -> Linear ->
 * Dropout, :165-167 x = x + ffn(norm2(x)); train.py:88-89) as one read + one write per tensor, n contiguous fp32 elements (n % 4 == 0):
 *   SMK_ELT_GELU_DROPOUT_FWD   out = dropout_p(gelu(a))                      a = the first Linear's output h
 *   SMK_ELT_GELU_DROPOUT_BWD   out = b * mask / (1 - p) * gelu'(a)           a = h, b = gradient of the dropout's output
 *   SMK_ELT_DROPOUT_ADD_FWD    out = b + dropout_p(a)                        a = the second Linear's output, b = the residual stream
 *   SMK_ELT_DROPOUT_BWD        out = a * mask / (1 - p)                      a = gradient of the sum (the residual's gradient is a itself)
 * GELU is the exact-erf form (nn.GELU() default).  The keep mask of element i is a pure function of (seed, i): the backward call
 * passes the forward's seed and p and no mask tensor exists (p is applied in units of 2^-16; p = 0 keeps every element). */
enum smk_elt_op { SMK_ELT_GELU_DROPOUT_FWD = 0, SMK_ELT_GELU_DROPOUT_BWD = 1, SMK_ELT_DROPOUT_ADD_FWD = 2, SMK_ELT_DROPOUT_BWD = 3 };
int smk_ffn_elementwise(int32_t op, const float *a, const float *b, float *out, int64_t n, double p, uint64_t seed, void *stream);

/* The owner's pass of the direct gradient exchange (hooks in where the reference steps the optimiser,
 * /root/reference/train.py:88-93 -- the reference itself is single-process and has no exchange): rank r holds `world` shards of n
 * gradients each (its own and the ones the all-to-all brought in, `stride` elements apart) and forms their MEAN: the sum in rank
 * order in fp32 (the same order on every rank, element and run: deterministic), a true divide by `world`, written as fp32 or bf16.
 * dtype: SMK_WIRE_F32 / SMK_WIRE_BF16 for input and output independently (world = 1 makes it the converter of the bf16 wire mode).
 * In place (out == shards) is allowed when the dtypes agree.  fp32 operands 16-byte aligned, bf16 operands 8-byte aligned, stride a
 * multiple of 4.  Enqueued on `stream`. */
enum smk_wire_dtype { SMK_WIRE_F32 = 0, SMK_WIRE_BF16 = 1 };
int smk_reduce_shards(const void *shards, int32_t in_dtype, int32_t world, int64_t n, int64_t stride, void *out, int32_t out_dtype, void *stream);

/* Training-mode BatchNorm2d (batch statistics) + ReLU + pool x pool mean pooling of an NCHW fp32 convolution output -- the
 * norm / activation / pool blocks of SmokePhysNet.input_encoder under autograd (smokephys_net.py:24-32 Conv -> BatchNorm2d -> ReLU,
 * :87-91 the two adaptive average pools as one block mean; train.py:88-89).
 *   forward:  z [B][C][H][W], gamma / beta [C], eps -> out [B][C][H/pool][W/pool] = blockmean(relu(bn(z))), and the batch
 *             statistics mean / var (biased) / rstd [C] (the caller updates running_mean / running_var from them);
 *   backward: dout [B][C][H/pool][W/pool] -> dz [B][C][H][W], dgamma / dbeta [C]; the ReLU mask is recomputed from z.
 * pool in {1, 4, 8}; H * W a multiple of 4,096 (pool 8: 16,384); pool > 1 needs W == 32 * pool.  `workspace`:
 * smk_bn_train_workspace(B, C, H, W, pool) bytes of device memory (partial sums), caller-owned.  Reductions are two-stage
 * in a fixed order (deterministic).  Enqueued on `stream`. */
int64_t smk_bn_train_workspace(int32_t B, int32_t C, int32_t H, int32_t W, int32_t pool);
int smk_bn_relu_pool_forward(const float *z, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma, const float *beta,
                             double eps, int32_t pool, float *out, float *mean, float *var, float *rstd, void *workspace,
                             void *stream);
int smk_bn_relu_pool_backward(const float *z, const float *dout, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma,
                              const float *beta, const float *mean, const float *rstd, int32_t pool, float *dz, float *dgamma,
                              float *dbeta, void *workspace, void *stream);

/* The encoder's first convolution for training: Conv2d(1, 64, 7, padding = 3) under autograd (smokephys_net.py:25; train.py:88-89), plain
 * fp32 on the vector ALUs (its output feeds train-mode BatchNorm + ReLU: as exact as an fp32 convolution has to be).
 *   forward: x [B][H][W] (the single input channel), weight [64][7][7], bias [64] or NULL -> z1 [B][64][H][W]; W % 4 == 0;
 *   wgrad:   dz [B][64][H][W], x -> dW [64][7][7] and db [64] (unless NULL); H % 4 == 0, W % 64 == 0; per-workgroup partial sums added
 *            in a fixed order (deterministic); `workspace`: smk_conv1_train_wgrad_workspace() bytes.
 * (The input frames carry no gradient in train.py; a caller that needs dX keeps PyTorch-ROCm's.) */
int smk_conv1_train_forward(const float *x, const float *weight, const float *bias, int32_t B, int32_t H, int32_t W, float *z1, void *stream);
int64_t smk_conv1_train_wgrad_workspace(void);
int smk_conv1_train_wgrad(const float *dz, const float *x, int32_t B, int32_t H, int32_t W, float *dw, float *db, void *workspace, void *stream);

/* The encoder's second convolution alone, for training: z2 = Conv2d(64, 128, 3, padding = 1)(a1) + bias under autograd
 * (smokephys_net.py:28; train.py:88-89 -- train-mode BatchNorm needs the whole convolution output before it can normalise, so the
 * fused eval encoder does not apply).  a1 [B][64][H][W] and z2 [B][128][H][W] NCHW fp32, weight [128][64][3][3], bias [128] or NULL;
 * H % 8 == 0, W % 16 == 0.  Split-bf16 on the bf16 matrix cores with fp32 accumulation: three bf16 terms per operand, six products
 * (within 2e-6 of an fp64 convolution -- the output feeds BatchNorm + ReLU, whose masks amplify forward error in the gradients).  `workspace`: smk_conv2_train_workspace() bytes of device memory; the split weights are rebuilt from `weight`
 * in the same call, so an optimizer step needs no other notification.  The weight / bias gradients stay with the caller (PyTorch-ROCm's
 * convolution_backward in models/conv.py); the data gradient is smk_conv2_train_dgrad.  Enqueued on `stream`. */
int64_t smk_conv2_train_workspace(void);
int smk_conv2_train_forward(const float *a1, const float *weight, const float *bias, int32_t B, int32_t H, int32_t W, float *z2,
                            void *workspace, void *stream);
/* dX [B][64][H][W] = the data gradient of that convolution from dz [B][128][H][W] (the 3x3 convolution 128 -> 64 with the flipped
 * kernel), same arithmetic and shape rules; `workspace` as above (its own contents: do not share one buffer between a forward and a
 * data-gradient call that may overlap).  */
int smk_conv2_train_dgrad(const float *dz, const float *weight, int32_t B, int32_t H, int32_t W, float *dx, void *workspace, void *stream);
/* dW [128][64][3][3] (and db [128] = sum of dz over batch and pixels, unless NULL) of that convolution from dz [B][128][H][W] and the saved
 * input a1 [B][64][H][W]: per 8 x 16 tile a GEMM with the pixels as the MFMA k dimension, accumulated in registers per workgroup, the
 * workgroups' partial sums added in a fixed order (deterministic, no atomics).  `workspace`: smk_conv2_train_wgrad_workspace() bytes. */
int64_t smk_conv2_train_wgrad_workspace(void);
int smk_conv2_train_wgrad(const float *dz, const float *a1, int32_t B, int32_t H, int32_t W, float *dw, float *db, void *workspace, void *stream);

/* The passes of the two calls above one at a time, for BatchNorm statistics that span several processes: data-parallel training
 * (train.py under DistributedDataParallel) gives every process a shard of the batch, while the reference's BatchNorm2d layers see the
 * whole batch of ONE process (smokephys_net.py:26,29; train.py:59-93).  The host all-reduces the per-channel statistics between the
 * phases (models/sync_bn.py).  Unused pointers may be NULL.
 *   SMK_BN_STATS     z -> mean / var (biased) / rstd [C] of THIS process's batch (workspace as above);
 *   SMK_BN_APPLY     out = blockmean(relu(bn(z))) from the GIVEN mean / rstd;
 *   SMK_BN_BWD_SUMS  dout, z, given mean / rstd -> this process's dgamma = sum dy * zhat and dbeta = sum dy (workspace);
 *   SMK_BN_BWD_DZ    dz from the GIVEN (all-reduced) dgamma / dbeta and count = elements per channel of the GLOBAL batch. */
enum smk_bn_phase { SMK_BN_STATS = 0, SMK_BN_APPLY = 1, SMK_BN_BWD_SUMS = 2, SMK_BN_BWD_DZ = 3 };
int smk_bn_relu_pool_phase(int32_t phase, const float *z, const float *dout, int32_t B, int32_t C, int32_t H, int32_t W, const float *gamma,
                           const float *beta, double eps, float *mean, float *var, float *rstd, int32_t pool, float *out, float *dz,
                           float *dgamma, float *dbeta, double count, void *workspace, void *stream);

/* The chaos term of one ChaosAttention layer folded into Q (chaos_attention.py:39-66 lorenz_system + generate_chaos_field,
 * :85-100 chaos_proj / chaos_gate / chaos_strength): noise [3][B] = the three randn(B,1) draws (before the 0.1 scale),
 * proj_w [D][3], proj_b [D], gate_w [D], gate_b [1] (PyTorch layouts) -> addend [B][5][ld_addend] (columns 0..D-1
 * written; a wider pitch lets it land in the q columns of a fused q|k|v addend), the value row l of the
 * sequence adds to its query: strength * sigmoid(gate(C_t)) * C_t, C_t = chaos_proj(Lorenz state t), t = l mod 5
 * (5 explicit-Euler steps, sigma/rho/beta/dt as given; reference: 10, 28, 8/3, 0.01).  Feed it to smk_linear_forward's
 * periodic_add of q_proj.  Replaces ~90 tiny elementwise launches per layer. */
int smk_chaos_addend(const float *noise, int32_t B, int32_t D, const float *proj_w, const float *proj_b,
                     const float *gate_w, const float *gate_b, double strength, double sigma, double rho, double beta,
                     double dt, float *addend, int64_t ld_addend, void *stream);

/* Softmax attention of ChaosAttention (chaos_attention.py:102-112) once the chaos term is folded into Q
 * (softmax(((Q + addend) K^T) * scale) V, heads merged back): q, k, v [B][L][ld*] fp32 with head h in columns
 * head_dim*h .. of a token row (the layout q_proj / k_proj / v_proj write), out [B][L][ldo] in the same convention
 * (what out_proj reads -- no transpose copy).  Flash style (no L x L tensor), split-bf16 MFMA, fp32-class accuracy.
 * Requires head_dim == 64 and L % 128 == 0 (else SMK_ERR_UNSUPPORTED); scale = 1 / (sqrt(head_dim) * temperature).
 * out_format: smk_format of out (split: dense rows, ldo == H * head_dim). */
int smk_attention(const float *q, const float *k, const float *v, void *out, int32_t B, int32_t L, int32_t H,
                  int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format,
                  void *stream);
/* The same attention with a workspace: when the grid of smk_attention would leave most CUs idle (B * H * L / 128 workgroups: 64 at batch 1 of
 * the model's shape), the keys of each query block are dealt to 2-8 workgroups whose partial (output, max, sum) states a second launch merges
 * in split order (deterministic).  smk_attention_workspace_bytes: the bytes that split needs for this problem on the current device (0: no
 * split would be made -- pass none).  A workspace that is null or too small, or a split-bf16 output, runs the unsplit kernel. */
int64_t smk_attention_workspace_bytes(int32_t B, int32_t L, int32_t H, int32_t head_dim);
int smk_attention_ws(const float *q, const float *k, const float *v, void *out, int32_t B, int32_t L, int32_t H,
                     int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format,
                     void *workspace, int64_t workspace_bytes, void *stream);
/* smk_attention_ws with k and v in kv_format = SMK_FMT_F32 or SMK_FMT_SPLIT4_INPLACE (same pointers, pitches and column convention; the
 * values the kernel multiplies are bit for bit those it would have formed from the fp32 k and v, so the output is identical). */
int smk_attention_kv(const float *q, const void *k, const void *v, void *out, int32_t B, int32_t L, int32_t H,
                     int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale, int32_t out_format, int32_t kv_format,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* Training form of smk_attention (chaos_attention.py:102-112 under autograd, train.py:88-89): the same forward with fp32 output,
 * plus lse [B][L][H] = log2 sum_j 2^(scale * log2(e) * q_i.k_j) per (token, head) -- the softmax normaliser the backward
 * recomputes P from (log2 units). */
int smk_attention_forward_lse(const float *q, const float *k, const float *v, float *out, float *lse, int32_t B, int32_t L,
                              int32_t H, int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, double scale,
                              void *stream);

/* Backward of that attention: dq, dk, dv [B][L][ldd*] (head h = columns 64h .. 64h+63) from q, k, v, the output gradient dout
 * [B][L][ldo], the forward's lse and delta [B][L][H] = sum_d dout * out per (token, head).  Two launches of one kernel body
 * (dk/dv per 128-key block, dq per 128-query block), split-bf16 MFMA like the forward, deterministic (no atomics).
 * Same shape limits as smk_attention. */
int smk_attention_backward(const float *q, const float *k, const float *v, const float *dout, const float *lse,
                           const float *delta, float *dq, float *dk, float *dv, int32_t B, int32_t L, int32_t H,
                           int32_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk,
                           int64_t lddv, double scale, void *stream);
/* delta [rows][H] = sum_d dout[row][64 h + d] * out[row][64 h + d] -- the `delta` input of smk_attention_backward (rowsum(dout * out) per head)
 * as one pass over the two tensors (row pitches ld_dout / ld_out floats). */
int smk_attention_delta(const float *dout, const float *out, int64_t rows, int32_t H, int32_t head_dim, int64_t ld_dout, int64_t ld_out,
                        float *delta, void *stream);

/* nn.LayerNorm over the last dimension (smokephys_net.py:149-150, applied at :161,:165; biased variance, eps as given):
 * x [rows][ldx] -> y [rows][ldy], weight / bias [D].  D % 4 == 0, D <= 2048 (else SMK_ERR_UNSUPPORTED).
 * y_format: smk_format of y (split: D % 8 == 0, dense rows, ldy == D). */
int smk_layernorm(const float *x, int64_t rows, int32_t D, int64_t ldx, const float *weight, const float *bias, double eps,
                  void *y, int64_t ldy, int32_t y_format, void *stream);

/* Backward of smk_layernorm (autograd of nn.LayerNorm, train.py:89): dx [rows][ld_dx] from x, dy and weight (the row statistics
 * are recomputed), dweight / dbias [D] as column sums over the rows added in a fixed order (deterministic).  `workspace`:
 * smk_layernorm_bwd_workspace(D) bytes of device memory, caller-owned.  Same D limits as smk_layernorm. */
int64_t smk_layernorm_bwd_workspace(int32_t D);
int smk_layernorm_backward(const float *x, const float *dy, int64_t rows, int32_t D, int64_t ldx, int64_t ld_dy,
                           const float *weight, double eps, float *dx, int64_t ld_dx, float *dweight, float *dbias,
                           void *workspace, void *stream);

/* ------------------------------------------------------------------ reconstruction head */
/* Eval-mode SmokePhysNet.reconstruction_head (smokephys_net.py:57-66): ConvTranspose2d(64,32,4,2,1) + BN + ReLU ->
 * ConvTranspose2d(32,16,4,2,1) + BN + ReLU -> Conv2d(16,1,3,padding 1) -> Sigmoid.  Device pointers, PyTorch layouts:
 * ct*_w [in][out][4][4], conv_w [1][16][3][3]; BN as (weight, bias, running_mean, running_var), eps 1e-5. */
typedef struct smk_decoder_weights {
    const float *ct1_w, *ct1_b, *bn1_w, *bn1_b, *bn1_mean, *bn1_var;
    const float *ct2_w, *ct2_b, *bn2_w, *bn2_b, *bn2_mean, *bn2_var;
    const float *conv_w, *conv_b;
} smk_decoder_weights;
typedef struct smk_decoder smk_decoder; /* opaque: BN-folded weights on the device */

int smk_decoder_create(const smk_decoder_weights *w, int32_t device_id, void *stream, smk_decoder **out);
int smk_decoder_destroy(smk_decoder *dec);

/* tokens [B][S*S][64] fp32 -- output_decoder's result, read as `transpose(1,2).view(B,64,S,S)` (smokephys_net.py:117) --
 * -> recon [B][1][4S][4S].  tmp1 [B][32][2S][2S] and tmp2 [B][16][4S][4S] are caller-owned scratch.  S % 16 == 0. */
int smk_decoder_forward(smk_decoder *dec, const float *tokens, int32_t B, int32_t S, float *tmp1, float *tmp2,
                        float *recon, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMOKEHIP_H */
