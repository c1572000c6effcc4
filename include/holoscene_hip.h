/*
 * include/holoscene_hip.h -- C ABI of libholoscene_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for HoloScene's Stage-1 hot path.  Every entry
 * point takes plain device pointers and sizes, runs asynchronously on the
 * hipStream_t passed as `stream` (NULL = the null stream), never allocates,
 * never synchronises, never throws.  Return value: 0 on success, <0 = error:
 *
 *   HS_ERR_ARG     (-1)  unsupported D / C / size combination (the reference
 *                        throws std::runtime_error for these: hashencoder.cu:607,622)
 *   HS_ERR_LAUNCH  (-2)  hipGetLastError() reported a failed launch
 *   HS_ERR_NULL    (-3)  a required pointer was NULL
 *
 * Ownership follows the reference (hashencoder/hashgrid.py:34-41,75-76,93-94):
 * the caller allocates every output; buffers documented as "accumulate" must be
 * zeroed (or hold a running sum) before the call.
 *
 * Section 1 mirrors, one to one, the three functions the reference exports
 * through pybind11 (hashencoder/src/bindings.cpp:5-9, prototypes
 * hashencoder/src/hashencoder.h:13-15).  Sections 2+ are the fused entry points
 * the MI355X build adds behind the same Python classes.
 */
#ifndef HOLOSCENE_HIP_H
#define HOLOSCENE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_OK 0
#define HS_ERR_ARG (-1)
#define HS_ERR_LAUNCH (-2)
#define HS_ERR_NULL (-3)

#define HS_MAX_LEVELS 32

/* storage types of activation tensors */
#define HS_F32 0
#define HS_BF16 1

/* ------------------------------------------------------------------ 0. library info */
/* ABI version; bumped whenever a signature below changes. */
int hs_abi_version(void);
/* "gfx950" -- the only architecture this library is built for. */
const char *hs_target_arch(void);

/* ------------------------------------------------------------------ 1. reference-compatible hash encoder
 *
 * All tensors float32, contiguous.  offsets is a DEVICE int32[L+1] array (as in
 * (An EMPTY level -- offsets[l + 1] == offsets[l] -- encodes to zeros and takes no gradient, in every hash entry point below: how a grid of fewer
 * than 16 levels is presented to the fused 16-level kernels, hashencoder/hashgrid.py: HashEncoder.fused_offsets.)
 * the reference, hashencoder.cu:107,120,151).  S = log2(per_level_scale),
 * H = base resolution.  D in {2,3}; C in {1,2,4,8} (second backward: C >= 2,
 * hashencoder.cu:678-684).
 */

/* Replaces hash_encode_forward (hashencoder.cu:728-751 -> kernel_grid :104-254).
 *   inputs [B,D] in [0,1] (points outside produce zeros), embeddings [sum_l n_l, C],
 *   outputs [L,B,C] (written), dy_dx [B, L*D*C] (written iff calc_grad_inputs). */
int hs_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets, float *outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                           int calc_grad_inputs, float *dy_dx, void *stream);

/* Replaces hash_encode_backward (hashencoder.cu:753-783 -> kernel_grid_backward :258-343,
 * kernel_input_backward :347-372).
 *   grad [L,B,C]; grad_embeddings [sum n_l, C] ACCUMULATE; grad_inputs [B,D] written iff calc_grad_inputs. */
int hs_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                            float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            int calc_grad_inputs, const float *dy_dx, float *grad_inputs, void *stream);

/* Replaces hash_encode_second_backward (hashencoder.cu:786-824 -> kernels :376-595).
 *   grad_grad_inputs [B,D]; grad_grad [L,B,C] written; grad2_embeddings ACCUMULATE. */
int hs_hash_encode_second_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                   const float *dy_dx, const float *grad_grad_inputs, float *grad_grad, float *grad2_embeddings,
                                   void *stream);

/* The same three entry points for the reference's other scalar types (AT_DISPATCH_FLOATING_TYPES_AND_HALF: hashencoder.cu:747, 778, 817 -- the
 * tensors' dtype selects the instantiation there; here `dtype` does).  HS_DTYPE_F32 forwards to the functions above; F64 / F16 run plain
 * kernels in the reference's order of operations under C++'s promotions and at::Half's operators (csrc/hash_encode_dt.hip): bit-identical to
 * oracle/hash_oracle_dt.c except for the order of the scatters' atomics.  Layouts as above (outputs [L,B,C], dy_dx [B, L*D*C]), every tensor of `dtype`. */
#define HS_DTYPE_F32 0
#define HS_DTYPE_F64 1
#define HS_DTYPE_F16 2
int hs_hash_encode_forward_dt(int32_t dtype, const void *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void *dy_dx, void *stream);
int hs_hash_encode_backward_dt(int32_t dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets,
                               void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                               const void *dy_dx, void *grad_inputs, void *stream);
int hs_hash_encode_second_backward_dt(int32_t dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx,
                                      const void *grad_grad_inputs, void *grad_grad, void *grad2_embeddings, void *stream);

/* ------------------------------------------------------------------ 2. strided / selective variants
 *
 * Same arithmetic, but (a) feature-like tensors are addressed as
 *   t[level*level_stride + b*point_stride + c]
 * so the kernel can read/write the point-major [B, L*C] layout the MLP consumes
 * without the reference's permute copy (hashgrid.py:44,61), (b) dy_dx is addressed as
 *   dy_dx[level*dydx_level_stride + b*dydx_point_stride + d*C + c]
 * (level-major [L,B,D*C] gives coalesced writes), and (c) NULL output pointers
 * skip the corresponding work (e.g. grad_embeddings == NULL skips the atomic
 * scatter when only d/dx is wanted, as in autograd.grad(sdf, x)).
 * `schedule`: 0 = level-major block order, 1 = XCD-affine (level l pinned to one
 * XCD's L2; needs L % 8 == 0, silently falls back to 0 otherwise).
 */
/* Device-side launch gate: a gated kernel returns at once unless *a > *b (two device floats); a == NULL = always run.
 * The sampler uses it to enqueue round r+1 of Algorithm 1 (gate: max beta of round r > beta0, ray_sampler.py:204) BEFORE
 * the host has read round r's convergence flag, so the device never idles on that read. */
typedef struct hsGate {
    const float *a;
    const float *b;
} hsGate;

/* Reduce-and-step (new: the reference steps torch.optim.Adam over the tables after the backward pass, training/holoscene_train.py:374).
 * When this scatter is the ONLY producer of the table's gradient in the iteration, the workgroup that owns a bin's cells holds their
 * final gradient in LDS -- it applies the Adam update (hs_adam_flat's arithmetic, element for element) to p / m / v right there
 * instead of adding the sums to `grad_embeddings`: the gradient table is neither zero-filled, nor written, nor read back by the
 * optimiser (3 x 48.8 MB per table and iteration at the stock grid).  p, m, v: the table's parameter and moment storage, indexed like
 * `grad_embeddings` (entry offsets[l] + cell, C floats each); state: the DEVICE-resident optimiser state hs_adam_tick has already
 * advanced for this update; group: which of its step sizes applies.
 * Contract on `grad_embeddings`: it must be all zero on entry; contributions that overflow a bin (or belong to a level that is not
 * binned, or arrive without a work space) still go there as atomics, are added by the owner of the cell, and are returned to zero --
 * so the buffer is all zero again afterwards and EVERY entry of the table has been stepped exactly once (entries without a gradient
 * with g = 0: dense Adam semantics, as hs_adam_flat). */
struct hsAdamState;
typedef struct hsTableStep {
    float *p, *m, *v;
    const struct hsAdamState *state;
    float beta1, beta2, eps, grad_scale;
    int32_t group;
    int32_t prior;             /* != 0: an EARLIER producer of this iteration has already added its gradient to `grad_embeddings` the plain way (the
                                * background-patch iteration evaluates the trunk twice): the owner of every cell adds what it finds there and returns
                                * it to zero, as it does for spilled contributions -- the LAST producer steps */
} hsTableStep;

typedef struct hsHashLayout {
    int64_t level_stride;      /* features / grads */
    int64_t point_stride;
    int64_t dydx_level_stride; /* dy_dx */
    int64_t dydx_point_stride;
    int32_t schedule;
    hsGate gate;               /* honoured by hs_hash_fwd only */
    void *scatter_ws;          /* NULL, or work space of hs_hash_scatter_ws_bytes() bytes: hs_hash_bwd / hs_hash_bwd_jac then scatter */
    uint32_t scatter_cap;      /* the hashed levels through per-bin record lists + an LDS reduction instead of global atomics     */
    /* Batched-over-grids launches (ONE launch for the points of N per-object grids, model/network.py:1835-2032 holds one
     * HashEncoder per object): NULL, or grid_id[b] in [0, N) for every point; grid g's table starts g * grid_stride ENTRIES
     * (of C floats) after `embeddings` / `grad_embeddings`, all grids share `offsets`.  Needs scatter_ws == NULL. */
    const int32_t *grid_id;
    int64_t grid_stride;
    int32_t ws_clean;          /* != 0: the caller guarantees that the counters at the head of scatter_ws are zero -- as a fresh zero-filled work
                                * space has them and as every hs_hash_bwd / hs_hash_bwd_jac leaves them -- so the clearing launch is skipped */
    int32_t out_bf16;          /* hs_hash_fwd, C == 2, D == 3, dy_dx == NULL only: `outputs` receives ONE 32-bit word per (point, level) = the two
                                * channels rounded to bf16 (round to nearest even, channel 0 in the low half) -- exactly what the SDF trunk kernel
                                * makes of the fp32 features, at half the bytes both ways (hs_sdf_mlp2_fwd: feat_bf16); the strides count words */
    /* hs_hash_bwd_jac only: NULL, or the dy_dx cotangent of points b < r1_n in RANK-ONE form -- g_dydx[l][b][d * C + c] =
     * (r1_scale * r1_ux[b * L * C + l * C + c]) * r1_g[b * D + d] -- instead of being read from `g_dydx` (those rows of g_dydx are then
     * not touched: the reverse-over-reverse trunk's cotangent is exactly this product, hs_trunk_rr_bwd_grad need not write 24 B x L per
     * sample and this kernel need not read them) */
    const float *r1_ux, *r1_g;
    uint32_t r1_n;
    float r1_scale;
    /* hs_hash_bwd / hs_hash_bwd_jac only: NULL, or the table's optimiser step taken INSIDE the reduction (see hsTableStep) */
    const struct hsTableStep *step;
} hsHashLayout;

/* Work space for the binned scatter (bytes; negative = error code) and the per-bin record capacity to put in the layout. */
#ifndef HS_SCATTER_BINS
#define HS_SCATTER_BINS 128
#endif
int64_t hs_hash_scatter_ws_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t *cap_out);

int hs_hash_fwd(const float *inputs, const float *embeddings, const int32_t *offsets, float *outputs,
                uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                float *dy_dx /* NULL = skip */, const hsHashLayout *layout, void *stream);

int hs_hash_bwd(const float *grad, const float *inputs, const int32_t *offsets,
                float *grad_embeddings /* NULL = skip scatter */, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                const float *dy_dx, float *grad_inputs /* NULL = skip */, const hsHashLayout *layout, void *stream);

int hs_hash_bwd2(const float *grad, const float *inputs, const int32_t *offsets,
                 uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                 const float *dy_dx, const float *grad_grad_inputs, float *grad_grad /* NULL = skip */,
                 float *grad2_embeddings /* NULL = skip */, const hsHashLayout *layout, void *stream);

/* Value+Jacobian backward: ONE scatter pass computing
 *   grad_embeddings += d<features, g_feat>/dE + d<dy_dx, g_dydx>/dE        (ACCUMULATE)
 * g_feat is addressed like features, g_dydx like dy_dx; either may be NULL.  With
 * g_dydx[l,b,d,c] = grad[l,b,c]*ggx[b,d] the second term equals the reference's
 * grad2_embeddings (hashencoder.cu:432-595); like the reference, d/dx of dy_dx is ignored. */
int hs_hash_bwd_jac(const float *g_feat, const float *g_dydx, const float *inputs, const int32_t *offsets, float *grad_embeddings,
                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const hsHashLayout *layout, void *stream);

/* ------------------------------------------------------------------ 3. error-bounded ray sampler (per-ray kernels)
 *
 * Fused replacements for the per-ray arithmetic of ErrorBoundSampler.get_z_vals
 * (model/ray_sampler.py:130-287, get_error_bound :450-458): one wavefront per ray, all
 * per-section arrays in LDS.  R rays; row r of every [R, *] array belongs to ray r.
 */
#define HS_SAMPLER_MAX_M 1024

/* One Algorithm-1 round, part 1 (ray_sampler.py:156-190, 204):
 *   z/sdf [R, ld]: in = sorted sample set of m_old entries per ray, out = merged set of m_old+s_new
 *   (samples/new_sdf [R, s_new] ascending per ray); beta [R]: in = current upper bound, out = the
 *   smallest beta within eps found by `beta_iters` bisection steps from *beta0 (device scalar);
 *   *beta_max = max(*beta_max, beta_r over the rays with beta_r > *beta0) via atomics (zero it before the call): it exceeds *beta0
 *   exactly when some ray is unconverged, which is all Algorithm 1 asks of it. */
int hs_sampler_update(float *z, float *sdf, int32_t ld, int32_t m_old, const float *samples, const float *new_sdf, int32_t s_new,
                      float *beta, const float *beta0, float eps, int32_t beta_iters, float *beta_max, int32_t R, const hsGate *gate /* NULL = none */,
                      const int32_t *m_dev /* NULL, or device count overriding m_old (hsSamplerCtl.m) */, void *stream);

/* Inverse-CDF sampling (ray_sampler.py:206-253): mode 0 = pdf ~ error-bound opacity + add_tiny,
 * mode 1 = pdf ~ rendering weights + 1e-5.  u [R, n_out] explicit, or NULL = linspace(0,1,n_out).
 * out [R, n_out]. */
int hs_sampler_draw(const float *z, const float *sdf, int32_t ld, int32_t m, const float *beta, int32_t mode, float add_tiny, const float *u,
                    int32_t n_out, float *out, int32_t R, const hsGate *gate /* NULL = none */,
                    const int32_t *m_dev /* NULL, or device count overriding m */, void *stream);

/* Device-side control of Algorithm 1's rounds, so that the whole sampler (and with it the whole training iteration) can be
 * captured in one HIP graph: the loop is unrolled max_rounds times, every kernel of round r+1 is gated on `running`
 * (hsGate{&ctl->running, &ctl->half}) and reads the merged sample count from ctl->m.
 *   hs_sampler_step: after a round's hs_sampler_update: m += s_new, rounds += 1, running = 0 when *beta_max <= *beta0
 *   (ray_sampler.py:204) or rounds == max_rounds.  Initialise ctl to {1.0f, 0.5f, 0, 0}.
 *   hs_sampler_pick: the n_extra extra-sample indices (ray_sampler.py:267-271): a partial Fisher-Yates shuffle of [0, m) driven
 *   by u [n_extra] ~ U[0,1) (train), or linspace(0, m-1, n_extra) truncated (u == NULL, eval). */
typedef struct hsSamplerCtl {
    float running, half;
    int32_t m, rounds;
} hsSamplerCtl;
int hs_sampler_step(hsSamplerCtl *ctl, const float *beta_max, const float *beta0, int32_t s_new, int32_t max_rounds, void *stream);
int hs_sampler_pick(const hsSamplerCtl *ctl, const float *u, int32_t n_extra, int64_t *pick, void *stream);

/* hs_sampler_draw + hs_sampler_step (+ hs_ray_points) in one launch, for the device-controlled loop with one hsSamplerCtl slot
 * per round: every workgroup derives the state after the round just updated from *ctl_in and *beta_max (hs_sampler_step's
 * rule), workgroup 0 writes it to *ctl_out (ctl_out != ctl_in), and the draw runs on that state: mode 0 returns at once when
 * the loop has stopped, mode 1 (the final draw) always runs, both with m = ctl_out->m.  z / sdf rows hold ld entries.
 * x != NULL: also the positions of the drawn depths, x / x01 [R*n_out,3] as hs_ray_points writes them. */
int hs_sampler_draw_step(const float *z, const float *sdf, int32_t ld, const float *beta, int32_t mode, float add_tiny, const float *u, int32_t n_out,
                         float *out, int32_t R, const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max, const float *beta0,
                         int32_t s_new, int32_t max_rounds, const float *cam_loc, const float *ray_dirs, float divide_factor,
                         float *x /* or NULL */, float *x01, void *stream);

/* The same with the step rule applied n_steps times, on beta_max[0 .. n_steps): the state after n_steps rounds derived in one go from the
 * initial slot and the per-round maxima.  With hs_sampler_update_draw the rounds themselves need no control slots: round r + 1 is gated
 * on hsGate{beta_max + r, beta0} (a round that did not run leaves its maximum at zero, so the gates chain by themselves) and its
 * merged count is the constant (r + 1) * s_new; only the final draw / pick / final need the realised state. */
int hs_sampler_draw_steps(const float *z, const float *sdf, int32_t ld, const float *beta, int32_t mode, float add_tiny, const float *u, int32_t n_out,
                          float *out, int32_t R, const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max /* [n_steps] */,
                          const float *beta0, int32_t s_new, int32_t max_rounds, int32_t n_steps, const float *cam_loc, const float *ray_dirs,
                          float divide_factor, float *x /* or NULL */, float *x01, void *stream);
/* hs_sampler_draw_steps (mode 1: the final draw of Algorithm 1) + hs_sampler_pick + hs_sampler_final in ONE launch -- the whole tail of
 * ErrorBoundSampler.get_z_vals (model/ray_sampler.py:224-280: final inverse-CDF draw, randperm(m)[:n_extra] / linspace extra samples,
 * cat + sort with near / far, z_samples_eik).  out [R, n_out] receives the drawn depths, z_out [R, n_out + 2 + n_extra] the sorted row,
 * z_eik [R] (optional) its eik_idx / eik_u-selected entry.  pick_in: explicit extra-sample indices (else drawn from u_pick, or the eval-mode
 * linspace when that is NULL too).  Requires n_out + 2 + n_extra <= s_new. */
int hs_sampler_tail(const float *z, const float *sdf, int32_t ld, const float *beta, float add_tiny, const float *u, int32_t n_out, float *out, int32_t R,
                    const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max, const float *beta0, int32_t s_new, int32_t max_rounds,
                    int32_t n_steps, const float *u_pick, const int64_t *pick_in, int32_t n_extra, float near, float far, const float *near_rays,
                    const float *far_rays, const int64_t *eik_idx, const float *eik_u, float *z_out, float *z_eik, void *stream);

/* hs_sampler_update with the NEXT round's draw fused in (mode 0, u = linspace: ray_sampler.py:206-253 on the set just merged, with
 * the beta just found): out [R, n_out] next depths and, with x != NULL, their positions x / x01 [R*n_out,3] as hs_ray_points writes
 * them.  The draw is speculative (whether a next round runs is known only when every ray has reported its beta); nothing but the
 * next round's gated kernels reads its outputs.  m_old is a host constant here (no device count). */
int hs_sampler_update_draw(float *z, float *sdf, int32_t ld, int32_t m_old, const float *samples, const float *new_sdf, int32_t s_new,
                           float *beta, const float *beta0, float eps, int32_t beta_iters, float *beta_max, int32_t R,
                           const hsGate *gate /* NULL = none */, float add_tiny, int32_t n_out, float *out, const float *cam_loc,
                           const float *ray_dirs, float divide_factor, float *x /* or NULL */, float *x01, void *stream);

/* Final sample set (ray_sampler.py:261-280): z_out [R, n_s+2+n_extra] = sort(z_samples ++ near ++ far ++ z[:, pick]);
 * z_eik [R] = z_out[r, eik_idx[r]] (skipped when z_eik is NULL).  pick [n_extra], eik_idx [R]: int64. */
int hs_sampler_final(const float *z_samples, int32_t n_s, const float *z, int32_t ld, const int64_t *pick, int32_t n_extra, float near, float far,
                     const int64_t *eik_idx, float *z_out, float *z_eik, int32_t R,
                     const float *near_rays, const float *far_rays /* [R] per-ray bounds overriding near / far (ray_sampler.py:290-447), or NULL */,
                     const float *eik_u /* NULL, or [R] U[0,1) draws replacing eik_idx: index = min(int(u * n), n - 1) */, void *stream);

/* Camera rays + first (uniform, stratified) depths + Lemma-2 beta in one launch (utils/rend_util.py:56-125 twice incl. the
 * 2x-offset depth-scale rays, model/ray_sampler.py:48-83 and :136-140).  uv [R,2] pixels; ray_offset [R,2] or NULL; pose,
 * intrinsics: DEVICE 4x4 row-major; t_rand [R,S] or NULL (= no stratified jitter, eval mode).
 * Outputs: ray_dirs [R,3], cam_loc [R,3], depth_scale [R], z0 [R,S], beta_init [R]; when x != NULL also the positions of
 * the first SDF sweep, x / x01 [R*S,3] exactly as hs_ray_points(cam_loc, ray_dirs, z0, ..., divide_factor) would write them. */
int hs_ray_setup(const float *uv, const float *ray_offset, const float *pose, const float *intrinsics, const float *t_rand, int32_t S, float near,
                 float far_cap, float bound, float eps, float *ray_dirs, float *cam_loc, float *depth_scale, float *z0, float *beta_init, int32_t R,
                 float divide_factor, float *x /* or NULL */, float *x01,
                 float offset_shift /* added to ray_offset: 0, or -0.5 when ray_offset holds raw U[0,1) draws */,
                 float *rot_out /* NULL, or [3,3]: the world-to-camera rotation pose[:3,:3]^T (network.py:917) */,
                 float *beta_work /* NULL, or [R]: a second copy of beta_init -- the sampler's working state, which its update kernels overwrite */,
                 const float *patch_u /* NULL, or two U[0,1) draws: the rays are the R = patch x patch pixels of a block placed by them (uv unused) --
                                       * the background patch of network.py:919-925: origin = floor(u * (floor(2 c) - patch + 1)) per axis */,
                 int32_t patch, void *stream);

/* Sample positions of a sampler round: x [R*S,3] = cam_loc[r] + z[r,s]*ray_dirs[r] (ray_sampler.py:151-153) and
 * x01 = (x/divide_factor + 1)/2, the hash grid's [0,1] coordinates (network.py:176, hashgrid.py:158), in one launch. */
int hs_ray_points(const float *cam_loc, const float *ray_dirs, const float *z, float *x, float *x01, int64_t R, int32_t S, float divide_factor,
                  const hsGate *gate /* NULL = none */, void *stream);

/* All positions of the render pass in one launch (model/network.py:805-811, 843-854): x [(R*N + 4R), 3] = the R*N rendered
 * samples cam_loc + z_vals*ray_dirs, then (training: z_eik != NULL) the Eikonal set [eik_uniform [R,3] | cam_loc + z_eik*ray_dirs]
 * and its copy jittered by (eik_jitter [2R,3] - 0.5)*0.01;  x01 = (x/divide_factor + 1)/2;  dirs_flat [R*N,3] = ray_dirs per sample.
 * eik_uniform is used as eik_uniform * eik_scale + eik_shift (1, 0: values already in the bounding box; 2b, -b: raw U[0,1) draws). */
int hs_render_points(const float *cam_loc, const float *ray_dirs, const float *z_vals, const float *z_eik, const float *eik_uniform,
                     const float *eik_jitter, int64_t R, int32_t N, float divide_factor, float *x, float *x01, float *dirs_flat, float eik_scale,
                     float eik_shift, void *stream);

/* ------------------------------------------------------------------ 4. value+Jacobian trunk, elementwise stages
 *
 * A, out, G, gA: [B, rows, W], storage type `dtype` = HS_F32 or HS_BF16 (arithmetic is fp32 either way; bias and
 * gbias are always f32); rows = 1 value row + up to 3 tangent rows; W % 4 == 0.
 * forward : v = A[b,0,:]+bias; out[b,0,:] = Softplus(beta=100)(v) (model/network.py:163); out[b,d,:] = sigmoid(100 v)*A[b,d,:]
 * backward: gA = d<out,G>/dA, gbias (ACCUMULATE, may be NULL) += sum_b gA[b,0,:]. */
int hs_softplus_tangent_fwd(const void *A, const float *bias, void *out, int64_t B, int32_t rows, int32_t W, int32_t dtype, void *stream);
int hs_softplus_tangent_bwd(const void *A, const float *bias, const void *G, void *gA, float *gbias, int64_t B, int32_t rows, int32_t W,
                            int32_t dtype, void *stream);
/* Same backward from the layer OUTPUT H (as hs_trunk_mlp_fwd keeps it) instead of the pre-activation: 4 rows per point. */
int hs_softplus_tangent_bwd_h(const void *H, const void *G, void *gA, float *gbias, int64_t B, int32_t W, int32_t dtype, void *stream);

/* ------------------------------------------------------------------ 5. fused dense Adam over a flat parameter buffer
 *
 * Replaces torch.optim.Adam(betas, eps) + ExponentialLR (training/holoscene_train.py:156-169, 374, 428).
 * p/g/m/v: flat f32 buffers of the same length; parameters of group 0 occupy [0, group_end[0]), group 1
 * [group_end[0], group_end[1]), group 2 the rest.  hsAdamState lives in DEVICE memory; initialise step = 0 and
 * lr0[] before the first step.  hs_adam_tick advances it (step += 1, bias corrections, lr_g = lr0_g * gamma^(step-1));
 * hs_adam_flat then updates elements [begin, end) (both multiples of 4) with gradients scaled by grad_scale (1/world_size after a sum
 * all-reduce).  Both are plain launches: no host reads, graph-capturable. */
#define HS_ADAM_MAX_GROUPS 3
typedef struct hsAdamState {
    int64_t step;
    int64_t group_end[2];
    float lr0[HS_ADAM_MAX_GROUPS];
    float lr[HS_ADAM_MAX_GROUPS];        /* current learning rates (written by hs_adam_tick) */
    float step_size[HS_ADAM_MAX_GROUPS]; /* lr / (1 - beta1^step) */
    float bc2_sqrt;                      /* sqrt(1 - beta2^step) */
} hsAdamState;

int hs_adam_tick(hsAdamState *state, float beta1, float beta2, double gamma, void *stream);
int hs_adam_flat(float *p, const float *g, float *m, float *v, int64_t begin, int64_t end, const hsAdamState *state, float beta1, float beta2,
                 float eps, float grad_scale, void *stream);
/* The same update with shard-local gradient / moment buffers (ZeRO-1 data parallelism, SURVEY 8e -- new, the reference is single-GPU):
 * p is the full flat buffer; g[0] holds flat element g_base and m[0], v[0] flat element mv_base (multiples of 4, <= begin), so a
 * rank keeps only its 1/N slice of both moments and steps straight from the reduce-scatter output. */
int hs_adam_flat_shard(float *p, const float *g, float *m, float *v, int64_t begin, int64_t end, int64_t g_base, int64_t mv_base,
                       const hsAdamState *state, float beta1, float beta2, float eps, float grad_scale, void *stream);

/* dst[0..n) = src[0..n) (fp32) for up to HS_COPY_MAX_JOBS tensor pairs in ONE launch: the per-parameter gradient tensors autograd hands
 * over -> their views in the flat gradient buffer (training/flat.py: gather_grads; the reference has no counterpart, its optimiser walks
 * the parameters one by one, training/holoscene_train.py:374). */
#define HS_COPY_MAX_JOBS 64
typedef struct hsCopyJob {
    const float *src;
    float *dst;
    int64_t n;
} hsCopyJob;
int hs_copy_many(const hsCopyJob *jobs, int32_t n_jobs, void *stream);

/* ------------------------------------------------------------------ 6. fused volume-rendering composite
 *
 * One workgroup per ray.  Replaces volume_rendering / occlusion_opacity and the weighted sums of
 * HoloSceneNetwork.forward (model/network.py:815-824, 904-906, 1803-1824) and their backward.
 *   z [R,N]; sdf [R*N] (scene min-SDF); raw [R*N,K] per-object SDFs; rgb [R*N,3]; g [R*N,3] (d sdf/dx);
 *   beta: device scalar; depth_scale [R]; sem_scale = implicit_network.sigmoid.
 * forward writes weights [R,N], transmittance [R,N] (may be NULL), rgb_out [R,3], depth_out [R] (already multiplied by
 * depth_scale), normal_out [R,3] (world frame; with rot != NULL -- a DEVICE row-major 3x3 world-to-camera rotation -- the camera-frame
 * normal map rot . n of network.py:917-918, and backward then expects the cotangent of THAT), sem_out [R,K], opac_out [R,K].
 * N <= 256, N*(8+2K) floats <= 64 KB.
 * backward takes the cotangents of those outputs (any may be NULL = zero) and writes d_sdf [R*N], d_raw [R*N,K],
 * d_rgb [R*N,3] (may be NULL), d_g [R*N,3] (may be NULL), and d_beta [R] (may be NULL): PER-RAY partial derivatives w.r.t. beta,
 * to be summed by the caller (one same-address atomic per ray serialised in the L2). */
int hs_composite_fwd(const float *z, const float *sdf, const float *raw, const float *rgb, const float *g, const float *beta,
                     const float *depth_scale, float sem_scale, int32_t R, int32_t N, int32_t K, float *weights, float *transmittance,
                     float *rgb_out, float *depth_out, float *normal_out, float *sem_out, float *opac_out, const float *rot /* or NULL */,
                     void *stream);
int hs_composite_bwd(const float *z, const float *sdf, const float *raw, const float *rgb, const float *g, const float *beta,
                     const float *depth_scale, float sem_scale, int32_t R, int32_t N, int32_t K, const float *g_weights, const float *g_rgb_out,
                     const float *g_depth, const float *g_normal, const float *g_sem, const float *g_opac, float *d_sdf, float *d_raw,
                     float *d_rgb, float *d_g, float *d_beta, const float *rot /* or NULL */, void *stream);

/* ------------------------------------------------------------------ 7. fused SDF-trunk inference on the matrix cores
 *
 * SDF branch of ObjectImplicitNetworkGrid.forward (model/network.py:169-210) for no-grad queries (the sampler's
 * sweeps, get_sdf_vals / get_object_sdf_vals :305-318), bf16 operands / fp32 accumulation:
 *   x [B,3] f32, feat [B,32] f32 (hash features) -> posenc(6) ++ feat (71, zero-padded to 96) -> 256 -> 256 -> d_out.
 *   W0 [256,96] bf16 (columns >= 71 zero) PRE-MULTIPLIED by 100*log2(e), W1 [256,256] bf16 as is, W2 [2][32*ceil(d_out/32), 256] (TWO planes since ABI 8: the matrix, then matrix - bf16(matrix))
 *   bf16 (rows >= d_out zero) PRE-MULTIPLIED by ln2/100 -- the kernel evaluates Softplus(beta=100) as log2(1 + 2^t) on
 *   t = 100*log2(e)*v (hs_pack_bf16's per-job `scale` does this); the biases are passed unscaled,
 *   b0,b1 [256] f32, b2 [d_out] f32; weights row-major [out][in] like nn.Linear.
 *   out_min [B] = min_k y_k (select < 0) or y_select, or -- select_mask != 0 -- the minimum over the objects whose bit is set
 *   (get_multi_object_sdf_vals, network.py:320-326); out_raw [B,d_out] optional (NULL = skip).  d_out <= 64. */
int hs_sdf_mlp_fwd(const float *x, const float *feat, const void *W0, const float *b0, const void *W1, const float *b1, const void *W2,
                   const float *b2, int32_t d_out, int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B,
                   const hsGate *gate /* NULL = none */,
                   int32_t feat_level_major /* 0: feat [B,32], 1: feat [16,B,2] (level-major, as hs_hash_fwd writes fully coalesced) */, void *stream);

/* The same function in "wave tile" form (csrc/sdf_mlp2.hip; d_out <= 32): every wave owns 32 points end to end, activations stay in
 * registers, W1 / W2 are LDS-resident, no workgroup barrier in the steady state.  Operands are FRAGMENT-ORDER images written by
 * hs_sdf_mlp2_pack from the fp32 effective matrices (row-major [out][in] like nn.Linear; W0 with row pitch ld0 >= 71 columns in the
 * reference's input order x | sin/cos octaves | hash features) -- scaling for the log2-domain softplus included; buffer sizes in
 * bytes from hs_sdf_mlp2_pack_bytes(0..3) = W0f, W1f, W2f, bias; W2f must directly follow W1f in memory (the LDS-resident image is
 * one linear copy).  Arguments of hs_sdf_mlp2_fwd as hs_sdf_mlp_fwd. */
int64_t hs_sdf_mlp2_pack_bytes(int32_t which);
int hs_sdf_mlp2_pack(const float *W0, int32_t ld0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                     void *W0f, void *W1f, void *W2f, float *bias, int32_t log2_domain /* 1: hs_sdf_mlp2_fwd; 0: hs_trunk_mlp2_fwd */, void *stream);

/* The sampler's per-round SDF query as ONE launch (ErrorBoundSampler.get_z_vals' `model.implicit_network.get_sdf_vals(points)`, ray_sampler.py:151-157
 * = HashEncoder.forward hashgrid.py:154 + the SDF branch of ObjectImplicitNetworkGrid.forward network.py:169-210): hs_hash_fwd(out_bf16) and
 * hs_sdf_mlp2_fwd / _wide in one kernel -- every lane gathers the eight levels of its own point that it feeds to the first layer.  Bit-identical to the
 * two launches.  x [B,3] world positions, x01 [B,3] grid coordinates, embeddings [offsets[16], 2] fp32, offsets int32 [17] (empty levels allowed behind
 * the grid's own), S / H as hs_hash_fwd; the packs as hs_sdf_mlp2_fwd; W2f_b / bias_b: second output tile's pack when 32 < d_out <= 64, else NULL. */
int hs_sdf_sweep_fwd(const float *x, const float *x01, const float *embeddings, const int32_t *offsets, float S, uint32_t H, const void *W0f,
                     const void *W1f, const void *W2f, const float *bias, const void *W2f_b, const float *bias_b, int32_t d_out, int32_t select,
                     uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate /* NULL = none */, void *stream);

/* Training form of the wave-tile kernel (csrc/trunk_mlp2.hip; d_out <= 32, L*C = 32, 6 encoding octaves): the value+Jacobian trunk pass of
 * hs_trunk_mlp_fwd below, same rows (4 per point: value, d/dx, d/dy, d/dz) and the same saved tensors for hs_trunk_mlp_bwd -- H0, H1
 * [M,256] bf16 row-major layer OUTPUTS, Y [M,d_out] f32 -- built straight from x [M/4,3], feat [M/4,32] (point-major) and dydx [L,M/4,3C]
 * (level-major, as hs_hash_fwd writes them), with the input rows it assembled stored as Xp [M,80] bf16 in the kernel's own column order
 * (hs_trunk_mlp2_input_column(c) = position of reference input column c, for un-permuting the first layer's weight gradient).
 * Operands: hs_sdf_mlp2_pack(..., log2_domain = 0). */
/* split != NULL (then Y may be NULL): the kernel writes what hs_trunk_split_fwd would derive from Y -- per-object SDFs, their minimum, its
 * index (lowest among equals) and the gradient of the minimum for the rendered points [0, n_main), the Eikonal outputs for the rest -- from
 * its output registers instead of storing Y [M, d_out] for a second kernel to read back (d_out = K objects). */
typedef struct hsTrunkSplit {
    int64_t n_main;        /* points [0, n_main) are rendered samples, [n_main, M/4) the Eikonal set */
    float *sdf_raw;        /* [n_main, K] */
    float *sdf;            /* [n_main] */
    int64_t *idx;          /* [M/4] */
    float *grad;           /* [n_main, 3] */
    float *y_eik;          /* [Be, K] */
    float *min_eik;        /* [Be] */
    float *grad_theta;     /* [(K+1) Be, 3] */
} hsTrunkSplit;
int32_t hs_trunk_mlp2_input_column(int32_t reference_column);
/* (hs_trunk_mlp2_fwd: ld = points per level of dydx; 0 = M / 4.  w2_planes: 2 = y = (W2hi + W2lo) h1, the default of every caller whose values
 * are used; 1 = the high plane only -- for a call that evaluates nothing but the Eikonal regulariser's points (ABI 8)) */
int hs_trunk_mlp2_fwd(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                      int32_t d_out, void *H0, void *H1, float *Y, void *Xp, int64_t M, float jac_scale, const hsTrunkSplit *split, int64_t ld,
                      int32_t w2_planes, void *stream);
/* 33 <= d_out <= 64 with the split outputs (Y is never stored): the last layer's second 32-row tile after the first, as hs_sdf_mlp2_fwd_wide --
 * W2f_b / bias_b are the W2f and bias buffers of a second hs_sdf_mlp2_pack(log2_domain = 0) whose W2 / b2 arguments are rows 32.. */
int hs_trunk_mlp2_fwd_wide(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                           const void *W2f_b, const float *bias_b, int32_t d_out, void *H0, void *H1, void *Xp, int64_t M, float jac_scale,
                           const hsTrunkSplit *split, int64_t ld, int32_t w2_planes, void *stream);
/* (hs_sdf_mlp2_fwd: feat_level_major 0 = fp32 [B, 32]; 1 = fp32 [16, B, 2]; 2 = `feat` points at uint32 [16, B], each word a level's two
 * channels as bf16 -- what hs_hash_fwd writes with hsHashLayout::out_bf16; results identical to the fp32 forms, which round the same way) */
int hs_sdf_mlp2_fwd(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, int32_t d_out,
                    int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate, int32_t feat_level_major,
                    void *stream);
/* 33 <= d_out <= 64 objects (the reference sizes the last layer by the scene's label count, training/holoscene_train.py:119-122): the last layer's
 * second 32-row tile after the first.  W0f / W1f / W2f / bias: a pack of (W0, W1, W2[0:32], b2[0:32]) with d_out = 32; W2f_b / bias_b: the W2f and
 * bias buffers of a second pack whose W2 / b2 arguments are rows 32.. (d_out - 32 of them).  select / select_mask index all d_out outputs. */
int hs_sdf_mlp2_fwd_wide(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, const void *W2f_b,
                         const float *bias_b, int32_t d_out, int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B,
                         const hsGate *gate, int32_t feat_level_major, void *stream);

/* The same function on FP32 operands (csrc/sdf_mlp32.hip: v_mfma_f32_32x32x2_f32, fp32 activations in registers, torch.nn.Softplus(beta = 100)
 * by expf / log1pf) -- the reference's own arithmetic (training/holoscene_train.py:45: no autocast) for the sampler sweeps of the fp32
 * configuration.  Images: hs_sdf_mlp32_pack_bytes(0..2) bytes for W0i, W1i (W1's tiles, each followed by its slice of W2), bias; features fp32, point-major [B, 32]
 * (feat_level_major = 0) or level-major [16, B, 2] (1); every other argument as hs_sdf_mlp2_fwd. */
int64_t hs_sdf_mlp32_pack_bytes(int32_t which);
int hs_sdf_mlp32_pack(const float *W0, int32_t ld0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                      float *W0i, float *W1i, float *bias, void *stream);
int hs_sdf_mlp32_fwd(const float *x, const float *feat, const float *W0i, const float *W1i, const float *bias, int32_t d_out,
                     int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate, int32_t feat_level_major,
                     void *stream);

/* Training form of the same trunk over value+Jacobian rows (4 rows per point; replaces the three nn.Linear + Softplus
 * applications of model/network.py:203-206 AND the autograd.grad re-traversals of :213-236, see DESIGN V1).
 *   X  [M, 96] bf16: hs_trunk_input_fwd output with pitch 96 (M = 4 * points, M % 4 == 0)
 *   W0 [256, 96], W1 [256, 256], W2 [2][32*ceil(d_out/32), 256] bf16 (zero-padded; two planes since ABI 8: the matrix, then matrix - bf16(matrix)), biases fp32
 *   H0, H1 [M, 256] bf16: layer outputs kept for the backward pass (value rows softplus100(v), tangent rows
 *   sigmoid(100 v) * pre-activation);   Y [M, d_out] fp32 (b2 added on value rows only).
 *   X == NULL: the input rows are built inside the kernel from x [M/4,3], feat [M/4, L*C] and dydx [L, M/4, 3*C] (exactly
 *   hs_trunk_input_fwd with nfreq = 6, L*C = 32) and written to Xout [M, 96] for the weight gradient. */
int hs_trunk_mlp_fwd(const void *X, const void *W0, const float *b0, const void *W1, const float *b1, const void *W2, const float *b2,
                     int32_t d_out, void *H0, void *H1, float *Y, int64_t M, const float *x, const float *feat, const float *dydx, void *Xout, int32_t L,
                     int32_t C, float jac_scale, void *stream);

/* Backward data path of hs_trunk_mlp_fwd in one kernel.
 *   g [M, g_pitch] bf16: cotangent of Y, zero-padded to g_pitch = 32 or 64 columns
 *   W2t [256, g_pitch] = W2^T (zero-padded), W1t [256, 256] = W1^T, both bf16
 *   gA1, gA0 [M, 256] bf16: cotangents of the two hidden pre-activations (feed the weight-gradient GEMMs and, for gA0,
 *   the input-gradient GEMM);  gb1, gb0 [256] fp32 (+=): bias gradients (may be NULL).
 *   W0t [256, 256] bf16 = W0^T (rows = the 96 padded input columns, zero beyond): if given, the cotangent gA0 . W0 of
 *   hs_trunk_input_fwd's output is formed in the same pass and its hash-feature part is written in the form the table
 *   scatter reads: g_feat [L, M/4, C] fp32 (LEVEL-major) from the value rows, g_dydx [L, M/4, 3*C]
 *   fp32 = jac_scale * the tangent rows.  Needs L*C == 32. */
int hs_trunk_mlp_bwd(const void *g, int32_t g_pitch, const void *H1, const void *H0, const void *W2t, const void *W1t, void *gA1, void *gA0,
                     float *gb1, float *gb0, const void *W0t /* NULL = skip */, float *g_feat, float *g_dydx, int32_t L, int32_t C, float jac_scale,
                     int64_t M, float *gb2 /* [g_pitch] fp32 (+=): column sums of g's value rows = last layer's bias gradient, or NULL */,
                     float *dW2_part /* NULL, or [hs_trunk_bwd_parts(M), g_pitch, 256] fp32: per-workgroup slices of the last layer's weight
                                        gradient g^T . H1 (sum them with hs_sum_slices, src_f32 = 1; rows >= d_out are zero) */,
                     int64_t ld /* points per level of the g_feat / g_dydx buffers; 0 = M / 4 (> M / 4: the buffers also hold other points) */,
                     void *stream);
int32_t hs_trunk_bwd_parts(int64_t M);   /* number of slices hs_trunk_mlp_bwd writes into dW2_part */

/* Consumers of hs_trunk_mlp_fwd's Y [4*B, K] and producers of its cotangent (K <= 64).  idx [B] = argmin_k of the value row
 * (lowest index among equals).  Points b < n_main are rendered samples: sdf_raw [n_main,K] = value rows, sdf [n_main] = min_k,
 * grad [n_main,3] = the Jacobian row of the minimum (model/network.py:289-299).  Points b >= n_main are the Eikonal set
 * (Be = B - n_main points): y_eik [Be,K], min_eik [Be], grad_theta [(K+1)*Be, 3] = the stacked gradient rows of
 * ObjectImplicitNetworkGrid.gradient (network.py:212-254: row k*Be+e = d sdf_k/dx, row K*Be+e = d min sdf/dx).
 * hs_trunk_split_bwd assembles the bf16 [4*B, KP] cotangent image hs_trunk_mlp_bwd reads from the cotangents of those outputs
 * (any may be NULL = zero). */
int hs_trunk_split_fwd(const float *Y, int64_t B, int64_t n_main, int32_t K, float *sdf_raw, float *sdf, int64_t *idx, float *grad, float *y_eik,
                       float *min_eik, float *grad_theta, void *stream);
int hs_trunk_split_bwd(const float *g_sdf_raw, const float *g_sdf, const int64_t *idx, const float *g_grad, const float *g_y_eik,
                       const float *g_min_eik, const float *g_grad_theta, int64_t B, int64_t n_main, int32_t K, int32_t KP, void *g, void *stream);

/* ------------------------------------------------------------------ 7b. fused colour branch of a rendered sample
 * Replaces, per rendered point: color_grid_feature_map_mlp (model/network.py:99-101, 186-188), the posenc + concat of
 * RenderingNetwork.forward (:586-596) and its three weight-normalised Linear layers + ReLU + sigmoid (:598-612), and
 * their autograd backward.  bf16 operands, fp32 accumulation.  B points.
 *   featc [16,B,2] fp32 colour hash features, LEVEL-major (hs_hash_fwd with level_stride = 2B, point_stride = 2: its coalesced
 *   store order); points, dirs, normals [B,3] fp32
 *   Wc0 [256,32], Wc1 [256,256]: colour MLP;  W_R0 [256,337] split into Wr0p [256,96] (columns 0..80 = encoded point |
 *   view dir | normal, zero-padded) and Wr0f [256,256] (columns 81..336 = feature vector);  Wr1 [256,256];  Wr2 [32,256]
 *   (rows 0..2 used).  All bf16 row-major (hs_pack_bf16 builds them); biases fp32 (br2: 3 values).
 *   Kept for the backward pass / weight gradients, all bf16: xin [B,128] = [featc | encoded inputs], hc, fv, r0, r1 [B,256]
 *   (layer outputs).  rgb [B,3] fp32 = sigmoid(...).
 *   relu_masks (may be NULL): hs_appearance_mask_words(B) 8-byte words receiving the signs of the three ReLU layers' outputs as wave
 *   ballots; given to hs_appearance_bwd they replace its reads of hc, r0, r1 (3 x B x 256 bf16 -> 3 x B x 256 bits). */
int64_t hs_appearance_mask_words(int64_t B);
int hs_appearance_fwd(const float *featc, const float *points, const float *dirs, const float *normals, const void *Wc0, const void *Wc1,
                      const void *Wr0f, const void *Wr0p, const void *Wr1, const void *Wr2, const float *bc0, const float *bc1, const float *br0,
                      const float *br1, const float *br2, void *xin, void *hc, void *fv, void *r0, void *r1, float *rgb, int64_t B,
                      uint64_t *relu_masks, void *stream);

/* Backward data path.  Transposed bf16 weights: Wr2t [256,32], Wr1t [256,256], Wr0ft [256,256] (= Wr0f^T), Wr0nt [32,256]
 * (rows j < 27 = column 54+j of W_R0: the encoded-normal inputs), Wc1t [256,256], Wc0t [32,256].
 * Outputs: gy [B,32] bf16 (cotangent of the pre-sigmoid outputs, columns 0..2), gA_r1, gA_r0, g_fv, gA_hc [B,256] bf16
 * (pre-activation cotangents; g_fv = cotangent of the feature vector), d_normals [B,3], g_featc [16,B,2] fp32 (level-major),
 * gbias [5,256] fp32 (+=; rows: br1, br0, bc1, bc0, br2 (3 values); may be NULL). */
int hs_appearance_bwd(const float *g_rgb, const float *rgb, const float *normals, const void *r1, const void *r0, const void *hc, const void *Wr2t,
                      const void *Wr1t, const void *Wr0ft, const void *Wr0nt, const void *Wc1t, const void *Wc0t, void *gy, void *gA_r1, void *gA_r0,
                      void *g_fv, void *gA_hc, float *d_normals, float *g_featc, float *gbias, int64_t B,
                      const uint64_t *relu_masks /* NULL: the masks are taken from r1, r0, hc; else those three may be NULL */, void *stream);

/* ------------------------------------------------------------------ 7c. the colour branch in wave-tile form (csrc/appearance2.hip)
 * Same function as hs_appearance_fwd / _bwd; a wave owns 32 samples through all five layers, the workgroup shares a pipeline of weight
 * chunks through LDS.  Weights as FRAGMENT images built by hs_appearance2_pack from the fp32 effective matrices (Wc0 [256,32], Wc1
 * [256,256], Wr0 [256,ldr0 >= 337], Wr1 [256,256], Wr2 [>=3,256]): buffers of hs_appearance2_pack_bytes(0: streamed image, 1: R2f,
 * 2: bias block) bytes.  featc [16,n,2] fp32 level-major; points, dirs, normals [n,3].  Kept for the backward pass and the weight
 * gradients, all TILE-PACKED (section 11): XAt [tiles][8 k-steps][64] x 16 B = [colour features | encodings] in the kernel's slot order
 * (hs_appearance2_enc_column), HCt, FVt, R0t, R1t [tiles][16][64] x 16 B; masks [tiles][3][64][4] uint32 = the ReLU signs of hc, r0, r1.
 * rgb [n,3]. */
int64_t hs_appearance2_pack_bytes(int32_t which);
int hs_appearance2_enc_column(int32_t half, int32_t slot);     /* encoded-input column (0..80) of slot 0..47 of lane half 0/1, -1 = padding */
int hs_appearance2_pack(const float *Wc0, const float *Wc1, const float *Wr0, int32_t ldr0, const float *Wr1, const float *Wr2, const float *bc0,
                        const float *bc1, const float *br0, const float *br1, const float *br2, void *stream_image, void *R2f, float *bias,
                        void *streamT_image /* NULL, or hs_appearance2_pack_t_bytes() bytes: the transposed image of the backward kernel */, void *stream);
/* (hs_appearance2_fwd: featc_words != 0: `featc` points at uint32 [16, n], a level's two channels as bf16 -- hsHashLayout::out_bf16) */
int hs_appearance2_fwd(const float *featc, const float *points, const float *dirs, const float *normals, const void *stream_image, const void *R2f,
                       const float *bias, void *XAt, void *HCt, void *FVt, void *R0t, void *R1t, uint32_t *masks, float *rgb, int64_t n, int32_t featc_words, void *stream);
/* Backward data path of the same: the transposed fragment image (hs_appearance2_pack_t_bytes() bytes, built by hs_appearance2_pack), the forward
 * pass's masks and rgb.  Outputs: gy [n,32] bf16 row-major (cotangent of the pre-sigmoid outputs, columns 0..2),
 * GR1t, GR0t, GFVt, GHCt tile-packed (pre-activation cotangents of r1, r0, the feature vector, hc), d_normals [n,3], g_featc [16,n,2] fp32
 * (level-major), gb2 [ceil(n / 32), 4] = per-tile partial sums of the last layer's bias gradient (columns 0..2; may be NULL).  The other bias gradients are column sums of the tile-packed
 * cotangents: hs_wgrad_pairs produces them beside the weight gradients (hsWgradPairJob::colsum).  normals_add != 0: d_normals already holds the
 * normals' cotangent from their other consumer (the compositing backward's normal map) and this kernel ADDS its own -- autograd's `grad += grad`
 * launch for a tensor with two consumers, folded into the store. */
int64_t hs_appearance2_pack_t_bytes(void);
int hs_appearance2_bwd(const float *g_rgb, const float *rgb, const float *normals, const uint32_t *masks, const void *streamT_image, void *gy, void *GR1t,
                       void *GR0t, void *GFVt, void *GHCt, float *d_normals, float *g_featc, float *gb2, int64_t n, int32_t normals_add, void *stream);

/* fp32 master matrices -> bf16 operand images in ONE launch: dst [dst_rows, dst_cols] (row-major bf16) receives the
 * [rows, cols] block of src (leading dimension ld) starting at (row0, col0) -- or, with transpose != 0, its transpose
 * (dst[r][c] = src[row0+c][col0+r]) -- zero-padded to the destination shape. */
#define HS_PACK_MAX_JOBS 16
typedef struct hsPackJob {
    const float *src;
    void *dst;
    int32_t ld, row0, col0;
    int32_t rows, cols;         /* valid extent, in destination orientation */
    int32_t dst_rows, dst_cols;
    int32_t transpose;
    float scale;                /* every element is multiplied by this before the bf16 rounding */
} hsPackJob;
int hs_pack_bf16(const hsPackJob *jobs, int32_t n_jobs, void *stream);

/* Weight gradients as split-M streaming reductions (csrc/wgrad.hip): for each job, part[s] = A[rows of slice s]^T . B[rows of slice s] as a
 * bf16 [slices, NA, MB] stack (sum it with hs_sum_slices), A [M, NA] and B [M, MB] row-major bf16, M % slices == 0.  All jobs of a call
 * run in ONE launch (workgroup = (job, slice)), so two 256 x 256 products fill the chip where the library's batched GEMM leaves half
 * of it idle.  Shapes: (NA, MB) in {(256, 256), (256, 128), (32, 256)}.  Replaces autograd's grad_output.t() @ input of nn.Linear. */
#define HS_WGRAD_MAX_JOBS 8
typedef struct hsWgradJob {
    const void *A;      /* [M, NA] bf16: cotangent of the layer's pre-activation (rows of the result) */
    const void *B;      /* [M, MB] bf16: the layer's input (columns of the result) */
    void *part;         /* [slices, NA, MB] bf16 */
    int64_t M;
    int32_t NA, MB;
} hsWgradJob;
int hs_wgrad_rows(const hsWgradJob *jobs, int32_t n_jobs, int32_t slices, void *stream);

/* dst[i] = sum_s src[s*n + i] (src bf16 or fp32 [slices, n], dst fp32 [n], n % 4 == 0) for up to HS_PACK_MAX_JOBS matrices in one launch:
 * the final reduction of the split-M weight-gradient GEMMs. */
typedef struct hsSumJob {
    const void *src;
    float *dst;
    int64_t n;
    int32_t slices;
    int32_t src_f32;   /* 0: src is bf16 (the GEMM partials), 1: src is fp32 (hs_trunk_mlp_bwd's dW2_part) */
} hsSumJob;
int hs_sum_slices(const hsSumJob *jobs, int32_t n_jobs, void *stream);

/* Small fp32 matrices assembled in ONE launch (csrc/small_ops.hip):
 *   dst[r * dst_ld + c] = sum over terms t of  sum_{k < red_t} src_t[k * red_stride_t + r * ld_t + (col_map_t ? col_map_t[c] : col0_t + c)]
 * for r < rows, c < cols -- bias gradient = accumulator + a column of a partial-sum matrix, weight gradient = a column selection of the
 * padded result, a column sum of per-workgroup partials (red > 1) ... what autograd would spend one ~5 us launch per operator on. */
#define HS_ASM_MAX_JOBS 12
#define HS_ASM_MAX_TERMS 3
typedef struct hsAsmTerm {
    const float *src;          /* fp32, or bf16 when src_bf16 (then ld / red_stride / col0 count bf16 elements) */
    const int32_t *col_map;    /* NULL: columns col0 .. col0 + cols - 1 */
    int64_t ld;                /* row stride of src (0: every row reads the same source row) */
    int64_t red_stride;
    int32_t col0;
    int32_t red;               /* >= 1: number of source blocks summed (e.g. the slices of a split-M partial stack [red, rows, ld]) */
    int32_t src_bf16;          /* 1: src holds bf16 (the weight-gradient kernels' partial stacks): the slice sum of hs_sum_slices and the column
                                * selection / accumulation of its result in the same launch */
    int32_t reserved;
} hsAsmTerm;
typedef struct hsAsmJob {
    float *dst;
    int64_t dst_ld;
    int32_t rows, cols, n_terms, reserved;
    hsAsmTerm term[HS_ASM_MAX_TERMS];
} hsAsmJob;
int hs_assemble(const hsAsmJob *jobs, int32_t n_jobs, void *stream);

/* LaplaceDensity.get_beta (model/density.py:28-30): y = |x| + shift[0] (y != NULL) and / or its backward gx = gy * sgn(x) (gx != NULL). */
int hs_abs_shift(const float *x, const float *shift, float *y, const float *gy, float *gx, int32_t n, void *stream);

/* Weight normalisation of several layers in one launch (nn.utils.weight_norm with dim = 0, model/network.py:158-159):
 * forward  W[r,:] = g[r] * v[r,:] / ||v[r,:]||;   backward (gW given)  gg[r] = <gW[r,:], v[r,:]> / ||v[r,:]||,
 * gv[r,:] = g[r]/||v[r,:]|| * (gW[r,:] - v[r,:] * <gW[r,:], v[r,:]> / ||v[r,:]||^2).  All fp32, row-major [rows, cols]. */
typedef struct hsWnJob {
    const float *v, *g, *gW;
    float *W, *gv, *gg;
    int32_t rows, cols;
} hsWnJob;
int hs_weight_norm(const hsWnJob *jobs, int32_t n_jobs, int32_t backward, void *stream);

/* Every weight image of a Stage-1 iteration in ONE launch: hs_sdf_mlp2_pack(log2_domain = 1) -> s* (the sampler sweeps), hs_trunk_pack_all ->
 * W0f .. w0t (the training trunk), hs_appearance2_pack -> stream_image .. streamT_image (the colour branch); arguments as in those three.  A NULL
 * first output of a group (sW0f, W0f, stream_image) leaves the group out; w1t / streamT_image NULL: no row-major transposes / no backward image. */
int hs_pack_iteration(const float *W0, int32_t ld0, int32_t f_in, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2,
                      int32_t d_out, void *sW0f, void *sW1f, void *sW2f, float *sbias, void *W0f, void *W1f, void *W2f, float *bias, void *W1Tf, void *W0Tf,
                      void *W2Tf, float *W2tab, void *w1t, void *w2t, void *w0t, const float *Wc0, const float *Wc1, const float *Wr0, int32_t ldr0,
                      const float *Wr1, const float *Wr2, const float *bc0, const float *bc1, const float *br0, const float *br1, const float *br2,
                      void *stream_image, void *R2f, float *abias, void *streamT_image, void *stream);

/* Head and tail of a training iteration, one launch each (csrc/iter_ops.hip).
 * hs_iter_prologue: (a) the weight-norm FORWARD of `jobs` (as hs_weight_norm); (b) rng_pool[0..n_rng) <- U[0, 1) draws, Philox-4x32-10 keyed by
 * rng_state[0] (seed) at counter rng_state[1]; the launch advances the counter (rng_state: three device uint64, [2] = scratch, zero between
 * launches), so every replay of a captured graph draws a new pool -- replaces torch.rand's generator kernel + its two Philox-state fills
 * (the reference draws where it needs them: network.py:785, 847-853, ray_sampler.py:79, 238, 269, 279); (c) beta_out = |beta| + beta_min[0]
 * (model/density.py:28-30); (d) adam != NULL: hs_adam_tick's update of the optimiser state; (e) zero[0..n_zero) <- 0 (16-byte aligned: the iteration's
 * gradient memset, optimizer.zero_grad of holoscene_train.py:354 for everything the scatters do not own).  Any part may be empty.
 * hs_iter_epilogue: the weight-norm BACKWARD of `jobs` + g_beta_out[i] = sgn(beta[i]) * (sum over up to 4 arrays [part_len[q], n_beta] of partial
 * cotangents -- the per-ray partials the compositing backward leaves --), written where the caller points (the flat gradient buffer's views). */
int hs_iter_prologue(const hsWnJob *jobs, int32_t n_jobs, float *rng_pool, int64_t n_rng, uint64_t *rng_state, const float *beta,
                     const float *beta_min, float *beta_out, int32_t n_beta, hsAdamState *adam, float beta1, float beta2, double gamma, float *zero, int64_t n_zero, void *stream);
int hs_iter_epilogue(const hsWnJob *jobs, int32_t n_jobs, const float *beta, const float *const *g_beta_parts, const int32_t *part_len,
                     int32_t n_parts, float *g_beta_out, int32_t n_beta, void *stream);

/* Batch assembly from device-resident arrays: dst[i, :] = src[idx[i], :], rows of row_bytes (a multiple of 4) bytes, for up to
 * HS_GATHER_MAX_JOBS arrays in one launch -- the sampled-pixel gather of a training batch (datasets/scene_dataset.py:143-167:
 * uv / rgb / depth / normal / label rows at sampling_idx) written straight into the training graph's static input block. */
#define HS_GATHER_MAX_JOBS 12
typedef struct hsGatherJob {
    const void *src;
    void *dst;
    const int64_t *idx;   /* [n] row indices into src */
    int64_t n;
    int32_t row_bytes;
} hsGatherJob;
int hs_gather_rows(const hsGatherJob *jobs, int32_t n_jobs, void *stream);

/* hs_trunk_rr_fwd_value + hs_trunk_rr_fwd_grad of the same samples in one launch (arguments as in those two; H0t / H1t are outputs only).
 * The activations stay in registers between the two chains; the five weight images cycle through LDS in chunks shared by the workgroup.
 * Replaces the same reference lines as the pair: model/network.py:ObjectImplicitNetworkGrid.forward + gradient (autograd.grad of the
 * minimum SDF), instant-sdf trunk. */
int hs_trunk_rr_fwd(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                    const float *W2tab, const void *W1Tf, const void *W0Tf, int32_t d_out, void *H0t, void *H1t, void *Xp, float *sdf_raw, float *sdf,
                    int64_t *idx, void *onehot, void *U0t, void *V1t, void *V0t, float *grad, float *uxh, float jac_scale, int64_t n, int64_t ld,
                    void *stream);

/* ------------------------------------------------------------------ reverse-over-reverse trunk of the rendered samples (csrc/trunk_rr.hip)
 *
 * ObjectImplicitNetworkGrid.get_outputs (model/network.py:273-301) for the rendered samples -- K per-object SDFs, their minimum, its
 * index and d min / dx -- and its double backward (the reference: autograd.grad(create_graph=True) :293-299, then loss.backward()), in
 * closed form on rows that are SAMPLES: see the header of csrc/trunk_rr.hip for the equations.  bf16 operands, fp32 accumulation.
 * "TP" = tile-packed activation tensor: [ceil(n / 32)][16][64] x 8 bf16 (lane (sample, half)'s four packed words of every k-step);
 * rows past n are written as zeros.  Images: W0f / W1f / W2f / bias from hs_sdf_mlp2_pack(log2_domain = 0); W1Tf / W0Tf / W2Tf / W2tab
 * from hs_trunk_rr_pack (sizes: hs_trunk_rr_pack_bytes(0..3)).  W2f is TWO bf16 planes since ABI 8 (hs_sdf_mlp2_pack_bytes(2) = 32 KB): W2's
 * fragments, then those of W2 - bf16(W2); every forward kernel forms y = (W2hi + W2lo) h1 -- the last layer's rows are a large common value plus
 * small learned structure, which one bf16 plane loses.
 *   fwd_value: x [n,3], feat [n,32] (point-major)  ->  H0t, H1t (TP), Xp [n,80] bf16 (hs_trunk_mlp2_input_column order), sdf_raw [n,K],
 *              sdf [n], idx [n] (arg-min, lowest among equals), onehot [n,32] bf16 (1 at idx)
 *   fwd_grad:  dydx [L=16, n, 6] as hs_hash_fwd wrote it  ->  U0t, V1t, V0t (TP), grad [n,3] = d min / dx, uxh [n,32] (hash columns of ux)
 *   bwd_grad:  g_grad [n,3]  ->  U0bt, A0pt, A1pt, U1bt (TP), UXb [n,80] bf16, g_dydx [16, n, 6] (cotangent of dy_dx, for hs_hash_bwd_jac)
 *   bwd_value: gy [n,32] bf16 (cotangent of the K outputs, the minimum's folded in at idx); A0pt / A1pt NULL when bwd_grad did not run
 *              ->  A0t, A1t (TP), g_feat [16, n, 2] (cotangent of the hash features, level-major) */
/* gy [n,32] bf16 = g_raw [n,K] (NULL = zeros) with g_sdf [n] (NULL = none) added at column idx[n]; gb2_part [HS_RR_GY_BLOCKS, 32] fp32 (NULL =
 * skip): per-block column sums of gy (their sum = the last layer's bias gradient) */
#define HS_RR_GY_BLOCKS 512
int hs_trunk_rr_gy(const float *g_raw, const float *g_sdf, const int64_t *idx, int32_t K, void *gy, float *gb2_part, int64_t n, void *stream);
/* hs_trunk_rr_gy and hs_trunk_split_bwd(NULL, NULL, idx_e, NULL, g_y_eik, g_min_eik, g_grad_theta, Be, 0, K, KPe, g_img) -- the cotangent image of the
 * Be Eikonal points' value+Jacobian rows -- in one launch (the two read nothing of each other). */
int hs_trunk_rr_gy_split(const float *g_raw, const float *g_sdf, const int64_t *idx, int32_t K, void *gy, float *gb2_part, int64_t n, const int64_t *idx_e,
                         const float *g_y_eik, const float *g_min_eik, const float *g_grad_theta, int64_t Be, int32_t KPe, void *g_img, void *stream);
int64_t hs_trunk_rr_pack_bytes(int32_t which);
int hs_trunk_rr_pack(const float *W0, int32_t ld0, const float *W1, const float *W2, int32_t d_out, void *W1Tf, void *W0Tf, void *W2Tf, float *W2tab,
                     void *stream);
/* hs_sdf_mlp2_pack (plain-domain images: log2_domain = 0) + hs_trunk_rr_pack + the three row-major bf16 transposes the value+Jacobian backward of
 * the Eikonal points reads (w1t [256,256] = W1^T, w2t [256,32] = W2^T zero-padded, w0t [256,256] = W0^T with rows >= f_in zero; all three NULL
 * = skip) in ONE launch: what a training pass of the trunk needs from its fp32 effective matrices. */
int hs_trunk_pack_all(const float *W0, int32_t ld0, int32_t f_in, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                      void *W0f, void *W1f, void *W2f, float *bias, void *W1Tf, void *W0Tf, void *W2Tf, float *W2tab, void *w1t, void *w2t, void *w0t,
                      void *stream);
int hs_trunk_rr_fwd_value(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, int32_t d_out,
                          void *H0t, void *H1t, void *Xp, float *sdf_raw, float *sdf, int64_t *idx, void *onehot, int64_t n, void *stream);
int hs_trunk_rr_fwd_grad(const float *x, const float *dydx, const int64_t *idx, const float *W2tab, const void *W1Tf, const void *W0Tf, const void *H0t,
                         const void *H1t, void *U0t, void *V1t, void *V0t, float *grad, float *uxh, float jac_scale, int64_t n, int64_t ld, void *stream);
/* (hs_trunk_rr_bwd_grad: g_dydx may be NULL -- its content is the rank-one product jac_scale * uxh[b, level * 2 + c] * g_grad[b, d], which
 * hs_hash_bwd_jac can form itself: hsHashLayout::r1_ux) */
int hs_trunk_rr_bwd_grad(const float *x, const float *dydx, const float *g_grad, const float *uxh, const int64_t *idx, const float *W2tab, const void *W0f,
                         const void *W1f, const void *H0t, const void *H1t, const void *U0t, void *U0bt, void *A0pt, void *A1pt, void *U1bt, void *UXb,
                         float *g_dydx, float jac_scale, int64_t n, int64_t ld, void *stream);
int hs_trunk_rr_bwd_value(const void *gy, const void *W2Tf, const void *W1Tf, const void *W0Tf, const void *H0t, const void *H1t, const void *A0pt,
                          const void *A1pt, void *A0t, void *A1t, float *g_feat, int64_t n, int64_t ld, void *stream);

/* The same two kernels for 33..64 objects (confs/custom/siebelgame: d_out = 64; training/holoscene_train.py:119-122: d_out = len(label_mapping)): the last
 * layer as TWO 32-row tiles.  W0f .. bias / W2Tf: the images of rows 0..31 (hs_trunk_pack_all), W2f_b / bias_b / W2Tf_b: W2f / bias / W2Tf of the packs of rows
 * 32..63 (hs_sdf_mlp2_pack(log2_domain = 0), hs_trunk_rr_pack), W2tab fp32 [64, 256] (both halves' tables, rows >= d_out unused).  sdf_raw [n, d_out];
 * onehot and gy (hs_trunk_rr_gy with K > 32; gb2_part then [blocks, 64]) are two planes [2][n][32]: objects 0..31 | 32..63 -- each plane is what the
 * 32-column weight-gradient jobs of hs_wgrad_pairs take.  hs_trunk_rr_bwd_grad needs no second form: it reads W2tab by the arg-min index. */
int hs_trunk_rr_fwd_wide(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                         const void *W2f_b, const float *bias_b, const float *W2tab, const void *W1Tf, const void *W0Tf, int32_t d_out, void *H0t, void *H1t,
                         void *Xp, float *sdf_raw, float *sdf, int64_t *idx, void *onehot, void *U0t, void *V1t, void *V0t, float *grad, float *uxh,
                         float jac_scale, int64_t n, int64_t ld, void *stream);
int hs_trunk_rr_bwd_value_wide(const void *gy, const void *W2Tf, const void *W2Tf_b, const void *W1Tf, const void *W0Tf, const void *H0t, const void *H1t,
                               const void *A0pt, const void *A1pt, void *A0t, void *A1t, float *g_feat, int64_t n, int64_t ld, void *stream);

/* Weight gradients of that formulation: part[slice] = sum over the job's one or two operand pairs of A^T B over the slice's rows
 * (csrc/wgrad_pairs.hip).  kind: HS_WGP_256x256 (A, B tile-packed), HS_WGP_256x80 (A tile-packed, B row-major [rows, 80]; result
 * [256, 128], columns >= 80 zero), HS_WGP_32x256 (A row-major [rows, 32], B tile-packed).  M = 32 * number of tiles; a slice is
 * ceil(tiles / slices) whole tiles, the last slices may be short or empty (their partial is then zero); rows <= M the valid rows of
 * the row-major operands; part: bf16 [slices, NA, MB]. */
#define HS_WGP_256x256 0
#define HS_WGP_256x80 1
#define HS_WGP_32x256 2
#define HS_WGP_256x256_RM 3     /* as HS_WGP_256x256 / _256x80 with BOTH operands row-major ([rows, 256] / [rows, 80]) */
#define HS_WGP_256x80_RM 4
#define HS_WGP_256x128_RM 5     /* A [rows, 256], B [rows, 128], both row-major */
#define HS_WGP_32x256_RM 6      /* A [rows, 32],  B [rows, 256], both row-major */
#define HS_WGP_256x128_TP 7     /* A tile-packed (16 k-steps), B tile-packed with 8 k-steps (hs_appearance2_fwd's assembled inputs) */
typedef struct hsWgradPairJob {
    const void *A0, *B0, *A1, *B1;      /* second pair optional (both NULL) */
    void *part;
    int64_t M, rows;
    int32_t kind, slices;
    int32_t ones;       /* HS_WGP_256x80 / _RM: column 80 of the result = column sums of A0 (a bias gradient); with B0 == NULL nothing else */
    int32_t reserved;   /* bit 0: keep this job on the register-staged form; bit 1: consecutive tiles per slice in the LDS-DMA form (see below) */
    float *colsum;      /* NULL, or fp32 [slices, NA]: per-slice column sums of A0 (kinds with a 256-row result: one more bias gradient per job) */
} hsWgradPairJob;
/* Two forms of the row stream: kinds 0, 1, 2, 7 (at least one tile-packed operand) bring their rows in by LDS-DMA, four 32-row stages in
 * LDS, three in flight; there a ROW-MAJOR operand's rows in [rows, M) are read as copies of row rows - 1 -- its partner in the pair is
 * tile-packed, zero in those rows, so they contribute nothing; slice s sums tiles s, s + slices, s + 2 slices, ... (with reserved bit 1:
 * ceil(tiles / slices) consecutive tiles, as the register form does).  The row-major-only kinds, jobs with `ones`, jobs with reserved bit 0 and
 * every job under HOLOSCENE_WGRAD_DMA=0 stage 64-row chunks through registers (rows >= `rows` read as zero). */
int hs_wgrad_pairs(const hsWgradPairJob *jobs, int32_t n_jobs, void *stream);

/* fp32 matrix products on the bf16 matrix cores (csrc/gemm_split.hip): each fp32 operand is split into `planes` bf16 planes (3: 24 mantissa
 * bits, the accuracy of an fp32 FMA chain, six plane products; 2: 16 bits, relative error ~1e-5, three) and the plane products are
 * accumulated in fp32.  The reference trains in fp32 (training/holoscene_train.py:45; nn.Linear = torch.addmm on fp32 operands): these two
 * replace the library GEMMs behind model/network.py's `_linear_rows` in the fp32 configuration.
 *   hs_gemm_split_nt:  C [M, N] (ldc) = A [M, K] (lda) . B [N, K]^T (ldb) + bias [N] (NULL = none)          -- y = x W^T + b;  g W via B = W^T
 *   hs_gemm_split_tn:  C_parts [slices, N, K] = per-slice A [m, N]^T . B [m, K] over ceil(M / slices) rows   -- dW = g^T x as split-M partials
 * All matrices row-major fp32; any M, N, K >= 0 (edges are masked; 16-byte-aligned rows take vector loads). */
int hs_gemm_split_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias, int64_t M, int32_t N, int32_t K,
                     int32_t planes, void *stream);
int hs_gemm_split_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C_parts, int64_t M, int32_t N, int32_t K, int32_t slices, int32_t planes,
                     void *stream);

/* The pixel draw of one training batch (datasets/ns_dataset.py:409-430: NSDataset.__getitem__'s class-balanced rule, there a dozen
 * torch.randperm calls per batch in DataLoader worker processes) as ONE launch.  The frame's pixels grouped by instance class in CSR
 * form: class_ptr [n_cls + 1] offsets into class_pix (pixel indices, class 0 = background first).  Class c contributes
 * min(size_c, quota_c) pixels, quota_0 = n_bg, quota_c = per_class -- a uniformly random subset when the class is larger than its
 * quota, all of it otherwise --, followed by n_uniform distinct pixels drawn uniformly from [0, total_pixels).  out_off [n_cls + 2]:
 * where each class's share (and, last but one, the uniform share) starts in `out` (int64 [sum of the shares]); the caller computes it
 * from the class sizes, which it knows.  (seed, counter) name the batch: the same pair gives the same batch on every run.
 * n_out = out_off[n_cls + 1] = the batch size (one thread per output position; any quota).
 * hs_draw_gather: the same draw plus the batch's row gather (hs_gather_rows, section 8) in the same launch -- a job whose `idx` IS `out`
 * gathers by the pixels just drawn (its n must equal n_out), any other job by its own index array. */
int hs_draw_pixels(const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                   int32_t n_uniform, int32_t total_pixels, int32_t n_out, uint64_t seed, uint64_t counter, int64_t *out, void *stream);
int hs_draw_gather(const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                   int32_t n_uniform, int32_t total_pixels, int32_t n_out, uint64_t seed, uint64_t counter, int64_t *out,
                   const struct hsGatherJob *jobs, int32_t n_jobs, void *stream);

/* The same launch driven from DEVICE memory, so that it can be a node of the training iteration's graph (an eager launch between two graph
 * replays costs the chip ~14 us of idle around it): batch number b = cursor[0]; frame f = sched[b % n_sched]; the frame's class lists and
 * per-frame sources come from frames[f]; the draw's counter is counter_base + b.  The last workgroup to finish stores cursor[0] = b + 1
 * (cursor[1] is its ticket: zero before the first launch, left zero).  jobs as hs_draw_gather's, with two extensions: jobs[j].src == NULL takes
 * frames[f].src[j]; jobs[j].idx == NULL gathers row f for every i < n (the frame's pose / intrinsics row).  Every frame must yield n_out rays. */
typedef struct hsFrameDesc {
    const int32_t *class_ptr, *class_pix, *out_off;
    int32_t n_cls, per_class, n_bg, reserved;
    const void *src[HS_GATHER_MAX_JOBS];
} hsFrameDesc;
typedef struct hsDrawSched {
    const hsFrameDesc *frames;      /* [n_frames] device array */
    const int32_t *sched;           /* [n_sched] frame of batch b at b % n_sched; the host refills it in stream order */
    uint64_t *cursor;               /* [2]: next batch number | ticket */
    uint64_t seed, counter_base;
    int32_t n_sched, n_frames;
} hsDrawSched;
int hs_draw_gather_sched(const hsDrawSched *sched, int32_t n_uniform, int32_t total_pixels, int32_t n_out, int64_t *out,
                         const struct hsGatherJob *jobs, int32_t n_jobs, void *stream);
/* hs_iter_prologue with hs_draw_gather_sched(draw, n_uniform, total_pixels, n_out, draw_out, gather, n_gather) riding in the same launch, its workgroups
 * first (draw == NULL: exactly hs_iter_prologue).  Nothing in the prologue reads what the draw writes; as a launch of its own the draw is 22 us of
 * dependent round trips on four workgroups. */
/* hs_hash_bwd with hs_draw_gather_sched(draw, ...) riding in front of the scatter's workgroups (draw == NULL: exactly hs_hash_bwd): the NEXT iteration's
 * batch, drawn where nothing reads the static batch block any more (after the loss) and where a 22-us latency chain costs nothing (under a 39-us scatter
 * on 6 000 workgroups) instead of being the next iteration's first link.  Needs a scatter launch (B > 0, grad_embeddings != NULL). */
int hs_hash_bwd_draw(const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                     float S, uint32_t H, const float *dy_dx, float *grad_inputs, const hsHashLayout *layout, const hsDrawSched *draw, int32_t n_uniform,
                     int32_t total_pixels, int32_t n_out, int64_t *draw_out, const struct hsGatherJob *gather, int32_t n_gather,
                     const struct hsAsmJob *sums /* NULL, or hs_assemble(sums, n_sums) riding along as well: slice sums of a backward stage whose inputs are
                                                  * complete before this launch and whose only reader comes after it */,
                     int32_t n_sums, void *stream);
/* hs_hash_bwd_jac with hs_assemble(sums, n_sums) riding in front of the scatter's workgroups (n_sums == 0: exactly hs_hash_bwd_jac). */
int hs_hash_bwd_jac_sums(const float *g_feat, const float *g_dydx, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                         uint32_t C, uint32_t L, float S, uint32_t H, const hsHashLayout *layout, const struct hsAsmJob *sums, int32_t n_sums, void *stream);
int hs_iter_prologue_draw(const struct hsWnJob *jobs, int32_t n_jobs, float *rng_pool, int64_t n_rng, uint64_t *rng_state, const float *beta,
                          const float *beta_min, float *beta_out, int32_t n_beta, struct hsAdamState *adam, float beta1, float beta2, double gamma, float *zero,
                          int64_t n_zero, const hsDrawSched *draw, int32_t n_uniform, int32_t total_pixels, int32_t n_out, int64_t *draw_out,
                          const struct hsGatherJob *gather, int32_t n_gather, void *stream);

/* ------------------------------------------------------------------ 8. fused network-input builders
 *
 * Positional encoding (model/embedder.py:11-36, order [v, sin 2^0 v, cos 2^0 v, sin 2^1 v, ...]) and concatenation
 * (model/network.py:181-185, 586-596) in one pass.  `dtype` (HS_F32/HS_BF16) is the storage type of out/G/feature_vectors.
 *
 * trunk input:  out [B,4,pitch], pitch >= F = 3+6*nfreq+L*C (columns >= F are zero: row padding for aligned matrix-core
 *   operands).  Row 0 = [posenc(x), feat]; row 1+d = its derivative w.r.t. x_d, the hash
 *   part being jac_scale * dydx[l,b,d*C+c] (dydx level-major [L,B,3*C] as written by hs_hash_fwd).
 *   backward (x is a constant): g_feat [B,L*C] = G[b,0,P:], g_dydx [L,B,3*C] = jac_scale * G[b,1+d,P+l*C+c]  -- both f32, in
 *   the layouts hs_hash_bwd_jac consumes.
 * render input: out [B, 3*(3+6*nfreq)+Fv] = [posenc(points), posenc(view_dirs), posenc(normals), feature_vectors];
 *   backward: d_normals [B,3] f32, d_feature_vectors [B,Fv] (dtype). */
int hs_trunk_input_fwd(const float *x, const float *feat, const float *dydx, void *out, int64_t B, int32_t nfreq, int32_t L, int32_t C,
                       float jac_scale, int32_t pitch, int32_t dtype, void *stream);
int hs_trunk_input_bwd(const void *G, float *g_feat, float *g_dydx, int64_t B, int32_t nfreq, int32_t L, int32_t C, float jac_scale, int32_t pitch,
                       int32_t dtype, void *stream);
int hs_render_input_fwd(const float *points, const float *view_dirs, const float *normals, const void *feature_vectors, void *out, int64_t B,
                        int32_t nfreq, int32_t Fv, int32_t dtype, void *stream);
int hs_render_input_bwd(const void *G, const float *normals, float *d_normals, void *d_feature_vectors, int64_t B, int32_t nfreq, int32_t Fv,
                        int32_t dtype, void *stream);

/* ------------------------------------------------------------------ 9. fused Stage-1 objective (value + analytic gradient)
 *
 * Terms of MonoSDFLoss / HoloSceneLoss (model/loss.py:211-346, 487-492) that depend on per-ray outputs, and the two terms
 * over the stacked SDF gradients.  Every g_* / d_* output is dLoss/dinput already multiplied by the term's weight.
 * hs_loss_rays (a wave-per-ray pass for the foreground flags and the opacity term, then one workgroup for the global
 *   reductions and the depth least-squares solve): rgb [R,3], depth [R], normal_map [R,3] (camera frame), opacity [R,K], sdf [R,N] (only its
 *   signs are used: foreground mask), priors rgb_gt/depth_gt/normal_gt/gt_mask [R,*], segs [R] int64.
 *   out5 = {rgb L1, scale-shift-invariant clipped depth MSE, normal L1, normal 1-cos, opacity BCE} (unweighted means).
 * hs_loss_eikonal: g1, g2 [H,3] = first / second half of the stacked gradient rows; acc2 (zeroed by the caller) receives
 *   {sum (|g1|-1)^2, sum |n1-n2|}; divide by H for the means. */
int hs_loss_rays(const float *rgb, const float *rgb_gt, const float *depth, const float *depth_gt, const float *normal_map, const float *normal_gt,
                 const float *gt_mask, const float *sdf, const float *opacity, const int64_t *segs, int32_t R, int32_t N, int32_t K, float w_rgb,
                 float w_depth, float w_l1, float w_cos, float w_opac, float *out5, float *g_rgb, float *g_depth, float *g_normal_map,
                 float *g_opacity, float *scratch /* [2R] work space */, void *stream);
int hs_loss_eikonal(const float *g1, const float *g2, int64_t H, float w_eik, float w_smooth, float *acc2, float *d_g1, float *d_g2, void *stream);
/* Both of the above plus the weighted total (loss.py:325-334, 655-657) as two launches with no host-side glue:
 * weights7 (HOST array) = weights of {rgb, depth, normal_l1, normal_cos, opacity, eikonal, smooth};
 * out8 = the seven unweighted terms in that order, then sum_i weights7[i] * term_i.  scratch [2R + 2 * HS_LOSS_EIK_BLOCKS] (needs no initialisation). */
#define HS_LOSS_EIK_BLOCKS 256
int hs_loss_stage1(const float *rgb, const float *rgb_gt, const float *depth, const float *depth_gt, const float *normal_map, const float *normal_gt,
                   const float *gt_mask, const float *sdf, const float *opacity, const int64_t *segs, int32_t R, int32_t N, int32_t K, const float *g1,
                   const float *g2, int64_t H, const float *weights7, float *out8, float *g_rgb, float *g_depth, float *g_normal_map, float *g_opacity,
                   float *d_g1, float *d_g2, float *scratch, void *stream);

/* Background-surface smoothness of the side x side (side <= 32) background patch (model/loss.py:519-557, 652-657):
 * compute_grad_error(depth) + compute_grad_error(normal), 4 scales of masked absolute first differences, mask = labels != 0.
 * depth [P], normal [P,3] (pixel-major), labels int64 [P], P = side^2.  out[0] = the loss, g_depth / g_normal = its gradient. */
int hs_bg_smooth_loss(const float *depth, const float *normal, const int64_t *labels, int32_t side, float *out, float *g_depth, float *g_normal,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HOLOSCENE_HIP_H */
