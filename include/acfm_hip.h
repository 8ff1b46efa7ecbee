/*
 * acfm_hip.h -- C ABI of libacfm_hip.so, the MI355X (gfx950) implementation of ACFM's
 * differentiable-render hot path.
 *
 * The reference (fkokkinos/acfm_video_3d_reconstruction) has no C ABI for this path: its
 * boundary is Python (multiframe/nnutils/nmr.py, geom_utils.py, loss_utils.py) on top of
 * the third-party PyTorch3D 0.3.0 extension `pytorch3d._C`.  Each entry point below names
 * the reference interface it replaces; the Python operator surface that mirrors the
 * reference's own signatures lives in acfm_video_3d_reconstruction_amd/nnutils/ and binds
 * these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *     tensors are dense, row-major, fp32 / int64 / int32 as declared;
 *   - the caller owns every buffer, outputs and workspace are pre-allocated
 *     (no allocation, no host synchronisation inside: all entry points are stream-ordered
 *     and hipGraph-capturable);
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - return value: 0 = ok, ACFM_E_* otherwise (the Python layer raises RuntimeError);
 *   - workspace sizes come from the matching *_workspace_bytes() query.
 *
 * Geometry conventions: cams [N,7] = (scale, tx, ty, qw, qx, qy, qz);
 * rendered image row/column grow with +y_p/+x_p of orthographic_proj_withz; packed face id
 * = n*F + f, -1 = empty; top-K ordered by ascending depth, ties -> smaller face id.
 */
#ifndef ACFM_HIP_H_
#define ACFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACFM_OK 0
#define ACFM_E_BADARG 1   /* shape / parameter outside what the kernels support */
#define ACFM_E_LAUNCH 2   /* hipLaunch or a preceding asynchronous error */
#define ACFM_E_WORKSPACE 3 /* workspace too small */

#define ACFM_MAX_K 32      /* faces_per_pixel upper bound */
#define ACFM_MAX_FACES 65535 /* faces per mesh (16-bit local ids in the per-pixel lists) */
#define ACFM_DETERMINISTIC 1 /* AcfmRasterTuning.flags */
#define ACFM_STORE_F16 2
#define ACFM_RECORD_COVER 4

/* library / device info ------------------------------------------------------------- */
int acfm_version(void);              /* 1000*major + minor */
const char* acfm_arch(void);         /* "gfx950" */

/* 0 when `stream` is not being captured into a hipGraph, else the (non-zero) id of the capture
 * (hipStreamGetCaptureInfo).  The Python layer uses it so that a cached face setup is never shared
 * across a graph boundary (ops._SETUP); no reference counterpart. */
int acfm_stream_capture_id(void* stream, unsigned long long* id_host);

/* ---- per-kernel timing (measurement aid, off by default) -------------------------------
 * When enabled, every kernel launch inside the entry points is bracketed by hipEvents on
 * the launch stream (ring of ACFM_PROF_RING pairs).  acfm_prof_collect synchronises those
 * events and returns, per kernel id, the summed duration in ms and the launch count since
 * the last collect/enable.  Not for use under hipGraph capture.  The ring is one per process
 * (all devices and threads share it, guarded by a mutex): a diagnostic, not a product feature. */
#define ACFM_PROF_SETUP 0
#define ACFM_PROF_SIL_FWD 1
#define ACFM_PROF_SIL_BWD 2
#define ACFM_PROF_PROJ_BWD 3
#define ACFM_PROF_TEX_FWD 4
#define ACFM_PROF_HARD_FWD 5
#define ACFM_PROF_TEX_BWD 6
#define ACFM_PROF_MASK_LOSS 7
#define ACFM_PROF_MASK_LOSS_BWD 8
#define ACFM_PROF_VISIBLE 9
#define ACFM_PROF_BDS 10
#define ACFM_PROF_BDS_BWD 11
#define ACFM_PROF_PROJECT 12
#define ACFM_PROF_TEX_MSE 13
#define ACFM_PROF_TEX_MSE_BWD 14
#define ACFM_PROF_DEFORM 15
#define ACFM_PROF_DEFORM_BWD 16
#define ACFM_PROF_SOLVE 17
#define ACFM_PROF_SOLVE_BWD 18
#define ACFM_PROF_NKERNELS 24
#define ACFM_PROF_RING 8192
int acfm_prof_enable(int on);
int acfm_prof_collect(float* ms_host, int* count_host, int n);
const char* acfm_prof_name(int id);

/* ---- projection ------------------------------------------------------------------
 * replaces geom_utils.orthographic_proj_withz / orthographic_proj / quat_rotate
 * (multiframe/nnutils/geom_utils.py:48-79, 134-152) and NeuralRenderer.project_points
 * (multiframe/nnutils/nmr.py:127-129).
 * verts [N,V,3], cams [N,7] -> proj [N,V,3] = (s*rot(q,X).xy + t, s*rot(q,X).z + offset_z) */
int acfm_project(const float* verts, const float* cams, int N, int V, float offset_z,
                 float* proj, void* stream);
/* grad_proj [N,V,3] -> grad_verts [N,V,3] (may be NULL), grad_cams [N,7] (may be NULL) */
int acfm_project_backward(const float* verts, const float* cams, const float* grad_proj, int N,
                          int V, float* grad_verts, float* grad_cams, void* stream);
/* The (x, y) part only, [N,V,2]: geom_utils.orthographic_proj (geom_utils.py:48-59) and
 * NeuralRenderer.project_points (nmr.py:127-129) without the [:, :, :2] slice and, backward, without the
 * zero-padded [N,V,3] gradient the slice's autograd builds. */
int acfm_project_xy(const float* verts, const float* cams, int N, int V, float offset_z, float* proj_xy,
                    void* stream);
int acfm_project_xy_backward(const float* verts, const float* cams, const float* grad_proj_xy, int N, int V,
                             float* grad_verts, float* grad_cams, void* stream);

/* ---- template deformation --------------------------------------------------------------
 * replaces the per-frame Cholesky solve of multiframe/main.py:586-609 (== predictor.py:260-276,
 * 313-315, monocular/main.py:204-218) once P = (L^T L + A^T A)^-1 A^T [V,K_h] is known
 * (factorised once per optimiser step, see deform.py):  verts[n] = mean_v + P delta[n].
 *   mean_v [V,3], P [V,K_h], delta [N,K_h,3] -> verts [N,V,3]   (f32 MFMA 16x16x4)
 * backward: grad_verts [N,V,3] -> grad_delta [N,K_h,3] = P^T g_n, grad_mean [V,3] = sum_n g_n,
 *   grad_P [V,K_h] = sum_n g_n delta_n^T (each may be NULL). */
int acfm_deform_apply(const float* mean_v, const float* P, const float* delta, int N, int V, int Kh,
                      float* verts, void* stream);
int acfm_deform_apply_backward(const float* P, const float* delta, const float* grad_verts, int N, int V,
                               int Kh, float* grad_delta, float* grad_mean, float* grad_P, void* stream);
/* The pre-solve sums of a frame-sharded step in double (the buffer [G = sum_n g_n delta_n^T | sum_n g_n | ..]
 * that the ranks all-reduce): G64 [V,K_h] and, optionally, mean64 [V,3], accumulated in double in a fixed order;
 * grad_P (optional) = G64 rounded to float, the same figure acfm_deform_apply_backward's grad_P approximates in
 * float arithmetic.  d lbs = acfm_deform_solve_backward(G) amplifies the rounding of G by the conditioning of the
 * system: with doubles through the exchange and ONE rounding behind it, the split of the frames over the ranks no
 * longer shows in the handle-weight gradient. */
int acfm_deform_presolve_sums_f64(const float* delta, const float* grad_verts, int N, int V, int Kh, double* G64,
                                  double* mean64, float* grad_P, void* stream);

/* ---- correlation cost volume (forward only) ----------------------------------------------
 * replaces correlation_cuda.forward as MaskFlownet calls it (multiframe/data/optical_flow/model/
 * correlation_package/correlation_cuda_kernel.cu:73-147, correlation.py:20-37; MaskFlownet.py:116,
 * 416: pad_size = max_displacement = md, kernel_size = 1, stride1 = stride2 = 1, corr_multiply = 1):
 *   f1, f2 [N,C,H,W] f32 -> out [N,(2md+1)^2,H,W],
 *   out[n,(tj+md)(2md+1)+(ti+md),y,x] = mean_c f1[n,c,y,x] f2[n,c,y+tj,x+ti]  (zero outside), md in 1..4.
 * The flow network is frozen in ACFM (flows are precomputed by the optical_flow scripts): no backward. */
int acfm_correlation_forward(const float* f1, const float* f2, int N, int C, int H, int W, int md, float* out,
                             void* stream);

/* ---- optical-flow loss -----------------------------------------------------------------
 * replaces the tail of loss_utils.optical_flow_loss (multiframe/nnutils/loss_utils.py:445-474):
 *   proj [B*T,V,3] projected vertices (proj_fn output, x/y in [-1,1]), flows [B*T,H,W,2] GT flow
 *   images, vis [B*T,V] visible-vertex bitmap of the hard raster  ->  loss [B,T-1]:
 *   sum over the kept vertices of |gt - (pix[k-1] - pix[k])|_1 / H / (kept + 1), gt = nearest pixel
 *   (grid_sample nearest, align_corners False, zeros), kept = visible in frame k and gt != 0;
 *   count [B,T-1] is saved for the backward pass (grad_loss [B,T-1] -> grad_proj [B*T,V,3]). */
int acfm_of_loss(const float* proj, const float* flows, const uint8_t* vis, int B, int T, int V, int H, int W,
                 float* loss, float* count, void* stream);
int acfm_of_loss_backward(const float* proj, const float* flows, const uint8_t* vis, const float* count,
                          const float* grad_loss, int B, int T, int V, int H, int W, float* grad_proj,
                          void* stream);
/* the same with the GT flows as the data loader holds them (multiframe/main.py:676-686 flips them in time, masks
 * them and repeats them for the G hypotheses every step; the loss reads V pixels per frame): flows [clips,T,H,W,2],
 * clip c of the B = G*clips rendered ones reads clip c % clips, frame k reads frame T-1-k if flip_t, and the value is
 * multiplied by masks [clips*T,H,W] at the same pixel of frame k if masks is not NULL. */
int acfm_of_loss_shared(const float* proj, const float* flows, const float* masks, const uint8_t* vis, int B, int T,
                        int V, int H, int W, int clips, int flip_t, float* loss, float* count, void* stream);
int acfm_of_loss_shared_backward(const float* proj, const float* flows, const float* masks, const uint8_t* vis,
                                 const float* count, const float* grad_loss, int B, int T, int V, int H, int W, int clips,
                                 int flip_t, float* grad_proj, void* stream);

/* ---- camera hypothesis pipeline --------------------------------------------------------
 * replaces the camera decode + mirror_cameras + transform_cameras chain of ShapeTrainer.forward /
 * warmup (multiframe/main.py:551-584, 452-466; the functions at :97-138): per camera row
 *   (s, t, q) = (relu(decay*e0 + 1) + 1e-12, e1..2, normalize(e3..6));
 *   blended by the frame's mirror flag with (s, -tx, ty, standardize(q_y(pi) * standardize(q)));
 *   then (s*a, tx*a + dx, ty*a + dy, q) blended by the frame's transform flag.
 *   emb [R,7] f32 (R = G*N rows, row r belongs to frame r % N), mirror_flag [N] i64,
 *   transforms [N,4] f32 (a, dx, dy, flag) -> cams [R,7];  backward: grad_cams -> grad_emb. */
int acfm_camera_pipeline(const float* emb, const int64_t* mirror_flag, const float* transforms, int R, int N,
                         float scale_lr_decay, float* cams, void* stream);
int acfm_camera_pipeline_backward(const float* emb, const int64_t* mirror_flag, const float* transforms,
                                  const float* grad_cams, int R, int N, float scale_lr_decay, float* grad_emb,
                                  void* stream);
/* pose of the horizontally flipped image of decoded cameras [R,7] (multiframe/main.py:97-125 with the flag set; the
 * texture branch renders every frame once more under it, main.py:627-636): (s, -tx, ty, standardize(q_y(pi) *
 * standardize(q))).  Forward only: the texture render sends no gradient to its cameras. */
int acfm_camera_mirror(const float* cams, int R, float* out, void* stream);
/* the same straight from the per-hypothesis embedding tables (multiframe/nnutils/mesh_net.py:436-444: one
 * nn.Embedding(frames, 7) per hypothesis; main.py:551-570 looks every one of them up and stacks / gathers the rows):
 * row r = g*N + n reads tables[selected ? selected[r] : g][frames_idx[n]] (tables: HOST array of n_tables <= 32 device
 * pointers to [n_frames,7] f32; selected [R] i64 or NULL: main.py:541-548's top-k choice).  backward: grad_tables[t]
 * ([n_frames,7], HOST array of device pointers, NULL entries skipped) are WRITTEN -- zero except the rows the batch
 * looked up, like nn.Embedding's dense gradient.
 * Out-of-range ids (frames_idx outside [0, n_frames), selected outside [0, n_tables): nn.Embedding raises there) are never
 * dereferenced: the forward writes a NaN camera for such a row, the backward skips it. */
int acfm_camera_pipeline_tables(const void* const* tables, int n_tables, int n_frames, const int64_t* frames_idx,
                                const int64_t* selected, const int64_t* mirror_flag, const float* transforms, int R,
                                int N, float scale_lr_decay, float* cams, void* stream);
int acfm_camera_pipeline_tables_backward(const void* const* tables, int n_tables, int n_frames, const int64_t* frames_idx,
                                         const int64_t* selected, const int64_t* mirror_flag, const float* transforms,
                                         const float* grad_cams, int R, int N, float scale_lr_decay,
                                         void* const* grad_tables, void* stream);

/* cam = (s, tx, ty, q / max(|q|, 1e-12)) of a raw [N,7] camera parameter: the per-iteration
 * torch.cat([scale, trans, F.normalize(quat)]) of the refinement loop (nnutils/predictor.py:301-308). */
int acfm_camera_normalize(const float* cam_raw, int N, float* cams, void* stream);
int acfm_camera_normalize_backward(const float* cam_raw, const float* grad_cams, int N, float* grad_raw,
                                   void* stream);

/* ---- template deformation solve ------------------------------------------------------
 * replaces the per-frame torch.cholesky / torch.cholesky_solve of multiframe/main.py:586-609
 * (also optimization/main.py:474-496, nnutils/predictor.py:260-276): with A = softmax(lbs, dim 0)^T
 * [K_h,V] (mesh_net.py:597-599) and the dense cotangent Laplacian L [V,V] of the mean shape,
 *   P = (L^T L + A^T A)^-1 A^T   [V,K_h]
 * by a blocked fp64 Cholesky on the matrix cores; once per optimiser step for all frames
 * (pred_v_n = mean_v + P delta_n, acfm_deform_apply).  K_h <= 32.
 * The workspace keeps the factor: acfm_deform_solve_backward turns grad_P [V,K_h] into
 * grad_lbs [V,K_h] (L carries no gradient, geom_utils.py:245).  acfm_deform_solve_info copies
 * the factorisation status to the host and synchronises the stream: 0 = ok; bit ACFM_SOLVE_INFO_HANDOFF set = a
 * hand-off wait of the single-launch factorisation expired (a starved wave: P holds NaNs; re-run the solve);
 * otherwise the low bits are 1 + first row of the 32-row tile with a non-positive pivot (the reference's
 * torch.cholesky raises there).  acfm_deform_solve_info_offset: byte offset of that int32 status word inside the
 * workspace, for callers that copy it without blocking (ops.py polls it that way between steps). */
#define ACFM_SOLVE_INFO_HANDOFF 0x40000000
size_t acfm_deform_solve_workspace_bytes(int V, int Kh);
int acfm_deform_solve(const float* L, const float* lbs, int V, int Kh, float* P, void* ws, size_t ws_bytes,
                      void* stream);
int acfm_deform_solve_backward(const float* grad_P, int V, int Kh, void* ws, size_t ws_bytes, float* grad_lbs,
                               void* stream);
int acfm_deform_solve_info(const void* ws, size_t ws_bytes, int V, int* info_host, void* stream);
size_t acfm_deform_solve_info_offset(int V);

/* ---- rasterisation workspace -------------------------------------------------------
 * Scratch of one render call of N meshes with V verts and F faces each at H x H pixels
 * (face records, NDC verts, per-mesh boxes, tile schedule, gradient scratch). */
size_t acfm_raster_workspace_bytes(int N, int V, int F, int H);

/* Per-call launch tuning of the raster entry points (pure speed: results never depend on it, `flags` apart).
 * NULL = the defaults.  There is no process-global tuning state in the library.
 *   split_mode: the heaviest 8x8 blocks of a launch are rendered by four workgroups each (a launch lasts at least as
 *               long as its longest block: all of a small launch's heavy blocks, the top few dozen of a large one);
 *               < 0 automatic: decided on the device from the cost histogram, a block is split when its cost exceeds
 *               (-split_mode / 4) x the mean work per wave slot (default -5: 1.25 x), 0 never, 1 always;
 *   grid_div:   workgroups per XCD group = entries / div for [0] the K-nearest forward, [1] the
 *               nearest-face (K = 1) forward, [2] the silhouette backward; 0 = default (4, 2, 4).
 *   flags:      bit 0 = deterministic silhouette backward: vertex gradients are accumulated in 64-bit fixed point
 *               (2^-36 units) with integer atomics, so the sums do not depend on the order in which blocks are
 *               served -- two runs are bit-identical, within 1e-6 of the default floating-point atomics (which are
 *               reproducible only to ~1e-6 relative);
 *               bit 1 = ACFM_STORE_F16 (BASELINE config 5, "fp16 render with fp32 loss accumulate"): the buffers
 *               declared `void*` [real] below hold IEEE half instead of float (rendered masks, images, silhouettes,
 *               atlases and the reference masks / distance transforms / images of the fused operators), and
 *               pix_to_face is an int32 [N,H,H] nearest-face plane (k_out must be 1).  Only storage changes: every
 *               accept / reject decision, depth, blend and loss sum stays fp32, so face ids are identical to the
 *               fp32 build and losses differ by the half rounding of what is stored (IoU drift < 1e-4).
 *               Gradients (grad_mask, grad_losses, grad_atlas, grad_verts, grad_cams) are always float.
 *               These two bits are the only fields that are not pure speed.
 *               bit 2 = ACFM_RECORD_COVER (pure speed): acfm_sil_forward / acfm_sil_loss_forward also leave, in the
 *               workspace, the nearest face that COVERS each pixel -- the answer of the hard K = 1 render of the same
 *               geometry (nmr.py:173-200: blur 0, clipped barycentrics), found during the walk the K-nearest render
 *               does anyway.  A texture render that takes the workspace over with ws_ready = 2 shades from that plane
 *               instead of binning and walking the faces again: identical outputs, bit for bit.
 * A backward call must pass the tuning of the forward whose workspace it takes over. */
typedef struct AcfmRasterTuning {
  int split_mode;
  int grid_div[3];
  int flags;
} AcfmRasterTuning;

/* ---- soft silhouette ---------------------------------------------------------------
 * replaces NeuralRenderer.forward, mask branch (multiframe/nnutils/nmr.py:143-172):
 * proj_fn -> y flip -> view (R=diag(-1,1,1), T=(0,0,2.732)) -> PyTorch3D
 * rasterize_meshes(K, blur_radius, bin_size=None) -> sigmoid_alpha_blend(sigma).
 *   verts_world [N,V,3] f32, faces [N,F,3] i64, cams [N,7] f32
 *   -> mask [N,H,H] f32, pix_to_face [N,H,H,k_out] i64 (packed ids, ascending depth, -1 empty)
 *      k_out = K: every kept face, as PyTorch3D returns them; k_out = 1: only the nearest-face
 *      plane (the K faces are still found and blended; it is the one slot the reference's
 *      callers read, loss_utils.py:214,431, and saves 8*(K-1) bytes per pixel of HBM writes)
 *   -> kth [N,H,H] u64 (optional, NULL to skip): state for acfm_sil_backward -- the
 *      (depth bits << 32 | face) key of the K-th kept face where K faces were kept, else ~0; defined where
 *      mask != 0 (all the backward reads) -- the 8x8 blocks no face comes near are not written
 *   -> vis [N,V] u8 (optional): 1 for every vertex of a face that is nearest in some pixel,
 *      i.e. the visible-vertex set of loss_utils.bds_loss (:214-224), fused into the raster
 * K in {2,4,8,10,20,32}. */
int acfm_sil_forward(const float* verts_world, const int64_t* faces, const float* cams, int N,
                     int V, int F, int H, int K, int k_out, float blur_radius, float sigma,
                     float offset_z, void* mask /* [real] */, void* pix_to_face /* int64, or the int32 plane */,
                     uint64_t* kth, uint8_t* vis, void* ws, size_t ws_bytes, const AcfmRasterTuning* tuning,
                     void* stream);

/* Extras of the silhouette render for what the reference's callers do NEXT TO it with the same vertices and cameras
 * (every field optional, NULL = not wanted; the `_ex` entry points are their plain namesakes + this structure):
 *   proj_xy         forward out [N,V,2]: NeuralRenderer.project_points(vertices, cams) (nmr.py:127-129), which the
 *                   trainer calls on the prediction it has just rendered (main.py:715, predictor.py:319): the face
 *                   setup projects the vertices anyway, so it hands (x, y) out instead of a second projection kernel;
 *   grad_proj_xy    backward in [N,V,2]: the upstream gradient of that proj_xy (the boundary loss's), added inside the
 *                   ONE projection backward of the silhouette render -- no second projection backward, no sum of two
 *                   [N,V,3] / [N,7] gradients afterwards;
 *   tex_*           forward: the pair renderer(...) / tex_renderer(...) on one prediction (main.py:620-626): besides its
 *                   own outputs the silhouette kernel stores the CONSTANT outputs of that texture render -- imgs 0, sil 0,
 *                   pix_to_face -1, texel_idx -1 -- on the 8x8 blocks no face comes near (~80 % of a frame) into the
 *                   caller's buffers for them (all four or none; needs ACFM_RECORD_COVER and float storage);
 *                   acfm_tex_forward(ws_ready = 3) on the same workspace then writes only the blocks with work. */
typedef struct AcfmSilExtras {
  float* proj_xy;
  const float* grad_proj_xy;
  float* tex_imgs;            /* [N,3,H,H] */
  float* tex_sil;             /* [N,H,H] */
  int64_t* tex_pix_to_face;   /* [N,H,H,1] */
  int32_t* tex_texel_idx;     /* [N,H,H] */
} AcfmSilExtras;
int acfm_sil_forward_ex(const float* verts_world, const int64_t* faces, const float* cams, int N,
                        int V, int F, int H, int K, int k_out, float blur_radius, float sigma,
                        float offset_z, void* mask, void* pix_to_face, uint64_t* kth, uint8_t* vis, void* ws,
                        size_t ws_bytes, const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream);
int acfm_sil_backward_ex(const float* verts_world, const int64_t* faces, const float* cams,
                         const void* mask, const uint64_t* kth, const float* grad_mask, int N, int V,
                         int F, int H, float blur_radius, float sigma, float offset_z,
                         float* grad_verts, float* grad_cams, void* ws, size_t ws_bytes,
                         int ws_from_forward, const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream);

/* replaces autograd through SoftSilhouetteShader + pytorch3d._C.rasterize_meshes_backward
 * (dists path) + the projection chain.  mask / kth are the forward's outputs;
 * grad_mask [N,H,H] -> grad_verts [N,V,3], grad_cams [N,7] (either may be NULL).
 * ws_from_forward != 0: `ws` is the untouched workspace of the matching acfm_sil_forward call
 * (same verts/faces/cams/H/blur) and the face setup is not repeated; 0: it is rebuilt here. */
int acfm_sil_backward(const float* verts_world, const int64_t* faces, const float* cams,
                      const void* mask /* [real] */, const uint64_t* kth, const float* grad_mask, int N, int V,
                      int F, int H, float blur_radius, float sigma, float offset_z,
                      float* grad_verts, float* grad_cams, void* ws, size_t ws_bytes,
                      int ws_from_forward, const AcfmRasterTuning* tuning, void* stream);

/* ---- fused soft silhouette + silhouette losses (opt-in) --------------------------------
 * acfm_sil_forward and acfm_mask_losses as ONE operator: the reference consumes the rendered mask at once
 * (multiframe/main.py:644-645, 715-716: l1_loss / edt_loss of mask_pred; predictor.py:317-320), so the loss terms
 * are summed in the raster kernel's epilogue (per-block partial sums, then summed per mesh in fixed order together
 * with sum(gt): deterministic, no atomics) and the backward forms d loss / d mask on the fly from gt / edt and
 * the per-mesh gradients -- the separate passes over the mask (acfm_mask_losses, acfm_mask_losses_backward)
 * and the [N,H,H] grad_mask buffer disappear.  Same outputs as acfm_sil_forward plus
 *   losses [N,4] = (mean|m - gt|, sum m gt, sum(m + gt - m gt), mean edt m)   (acfm_mask_losses' vector);
 * gt / edt [ref_batch,H,H] (either may be NULL = zeros), prediction n <-> reference n % ref_batch.
 * backward: grad_losses [N,4] -> grad_verts / grad_cams. */
int acfm_sil_loss_forward(const float* verts_world, const int64_t* faces, const float* cams, const void* gt /* [real] */,
                          const void* edt /* [real] */, int ref_batch, int N, int V, int F, int H, int K, int k_out,
                          float blur_radius, float sigma, float offset_z, void* mask /* [real] */, void* pix_to_face,
                          uint64_t* kth, uint8_t* vis, float* losses, void* ws, size_t ws_bytes,
                          const AcfmRasterTuning* tuning, void* stream);
int acfm_sil_loss_backward(const float* verts_world, const int64_t* faces, const float* cams, const void* mask,
                           const uint64_t* kth, const void* gt, const void* edt, int ref_batch,
                           const float* grad_losses, int N, int V, int F, int H, float blur_radius, float sigma,
                           float offset_z, float* grad_verts, float* grad_cams, void* ws, size_t ws_bytes,
                           int ws_from_forward, const AcfmRasterTuning* tuning, void* stream);
int acfm_sil_loss_forward_ex(const float* verts_world, const int64_t* faces, const float* cams, const void* gt /* [real] */,
                             const void* edt /* [real] */, int ref_batch, int N, int V, int F, int H, int K, int k_out,
                             float blur_radius, float sigma, float offset_z, void* mask, void* pix_to_face,
                             uint64_t* kth, uint8_t* vis, float* losses, void* ws, size_t ws_bytes,
                             const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream);
int acfm_sil_loss_backward_ex(const float* verts_world, const int64_t* faces, const float* cams, const void* mask,
                              const uint64_t* kth, const void* gt, const void* edt, int ref_batch,
                              const float* grad_losses, int N, int V, int F, int H, float blur_radius, float sigma,
                              float offset_z, float* grad_verts, float* grad_cams, void* ws, size_t ws_bytes,
                              int ws_from_forward, const AcfmRasterTuning* tuning, const AcfmSilExtras* extras,
                              void* stream);

/* ---- hard rasteriser (K = 1, blur 0) -------------------------------------------------
 * replaces OF_NeuralRenderer.forward (multiframe/nnutils/nmr.py:224-238): verts are
 * ALREADY projected by proj_fn; no y flip; view R=diag(-1,1,1), T=(0,0,2.732).
 * -> pix_to_face [N,H,H,1] i64 */
int acfm_hard_raster(const float* verts_proj, const int64_t* faces, int N, int V, int F, int H,
                     int64_t* pix_to_face, uint8_t* vis /* optional [N,V], as above */, void* ws,
                     size_t ws_bytes, const AcfmRasterTuning* tuning, void* stream);

/* ---- atlas-textured render -----------------------------------------------------------
 * replaces NeuralRenderer.forward, texture branch with atlas=True
 * (multiframe/nnutils/nmr.py:173-200): hard raster K=1, clip_barycentric_coords=True,
 * TexturesAtlas.sample_textures, ambient-only SoftPhongShader, softmax_rgb_blend.
 *   atlas [N,F,R,R,3] f32 -> imgs [N,3,H,H], sil [N,H,H], pix_to_face [N,H,H,1] i64,
 *   texel_idx [N,H,H] i32 (linear texel index n*F*R*R + f*R*R + y*R + x, -1 empty;
 *   saved for the backward pass).
 * ws_ready != 0: `ws` already holds the face setup of THE SAME verts / faces / cams / H /
 * offset_z, left there by acfm_sil_forward with blur_radius = ws_blur (the reference renders the
 * silhouette and the texture of one prediction back to back, main.py:616-636): projection and
 * face setup are skipped and the blur-expanded boxes tightened by sqrt(ws_blur).
 * ws_ready == 2: that acfm_sil_forward ran with ACFM_RECORD_COVER (and the same tuning is passed here): the nearest
 * covering face of every pixel is read from the workspace, nothing is binned or walked.
 * ws_ready == 3: as 2, and that render was acfm_sil_forward_ex (AcfmSilExtras.tex_*) with THESE imgs / sil / pix_to_face / texel_idx
 * buffers: the blocks no face comes near hold their constants already and are not written again.
 * atlas_batch: number of distinct atlases, atlas [atlas_batch,F,R,R,3]; mesh n samples atlas
 * n % atlas_batch (the trainer renders G camera hypotheses of every frame with the frame's one
 * texture, textures.repeat(G, ...) at main.py:627-636: atlas_batch = N / G spares the copies, and
 * the backward accumulates the G renders straight into the one gradient). */
int acfm_tex_forward(const float* verts_world, const int64_t* faces, const float* cams,
                     const void* atlas /* [real] */, int N, int V, int F, int H, int R, float sigma,
                     float gamma, float offset_z, void* imgs /* [real] */, void* sil /* [real] */, void* pix_to_face,
                     int32_t* texel_idx, void* ws, size_t ws_bytes, int ws_ready, float ws_blur,
                     int atlas_batch, const AcfmRasterTuning* tuning, void* stream);
/* NeuralRenderer.forward with atlas=False (multiframe/nnutils/nmr.py:177-179, used by
 * utils/bird_vis.py for visualisation): Textures(verts_rgb) = barycentric interpolation of
 * per-vertex colours verts_rgb [N,V,3]; forward only.  Workspace: raster workspace + 4*N*H*H. */
int acfm_vertex_color_forward(const float* verts_world, const int64_t* faces, const float* cams,
                              const float* verts_rgb, int N, int V, int F, int H, float sigma,
                              float gamma, float offset_z, float* imgs, float* sil,
                              int64_t* pix_to_face, void* ws, size_t ws_bytes, int ws_ready, float ws_blur,
                              const AcfmRasterTuning* tuning, void* stream);
/* grad_imgs [N,3,H,H] -> grad_atlas [N,F,R,R,3] (zeroed here, then scatter-added).
 * Integer texel indexing sends no gradient to geometry. */
int acfm_tex_backward(const float* grad_imgs, const int32_t* texel_idx, int N, int F, int H, int R, int atlas_batch,
                      float* grad_atlas, void* stream);
/* Same gradient in gather form (R <= 8): one wave per (atlas, face) sums the pixels of the face's box
 * whose texel belongs to it and stores all its texels -- no global atomics, no zero fill.  ws = the
 * workspace acfm_tex_forward ran on (its face boxes), ws_blur = the blur it was set up with (0 unless
 * taken over from a silhouette render, ws_ready).  Replaces the TexturesAtlas index_put backward. */
int acfm_tex_backward_faces(const float* grad_imgs, const int32_t* texel_idx, const void* ws, size_t ws_bytes,
                            float ws_blur, int N, int V, int F, int H, int R, int atlas_batch, float* grad_atlas,
                            void* stream);

/* ---- fused atlas-textured render + masked texture MSE (opt-in) ---------------------------
 * acfm_tex_forward and acfm_tex_mse as ONE operator (multiframe/main.py:627-636 renders texture_pred and :655-662
 * takes F.mse_loss(texture_pred * mask, imgs * mask).mean((1,2,3)) of it at once): the squared differences are summed
 * in the raster kernel's epilogue (per-block partials, fixed-order sums: deterministic), and the atlas gradient is
 * gathered per face straight from (imgs, ref_img, ref_mask, grad_loss) -- no [N,3,H,H] gradient image, no separate
 * passes (acfm_tex_mse, acfm_tex_mse_backward).  Outputs of acfm_tex_forward plus loss [N];
 * ref_img [ref_batch,3,H,H], ref_mask [ref_batch,H,H], prediction n <-> reference n % ref_batch.  R <= 8. */
int acfm_tex_mse_forward(const float* verts_world, const int64_t* faces, const float* cams, const void* atlas,
                         const void* ref_img, const void* ref_mask, int ref_batch, int N, int V, int F, int H, int R,
                         float sigma, float gamma, float offset_z, void* imgs, void* sil, void* pix_to_face,
                         int32_t* texel_idx, float* loss, void* ws, size_t ws_bytes, int ws_ready, float ws_blur,
                         int atlas_batch, const AcfmRasterTuning* tuning, void* stream);
int acfm_tex_mse_backward_faces(const void* imgs, const void* ref_img, const void* ref_mask, int ref_batch,
                                const float* grad_loss, const int32_t* texel_idx, const void* ws, size_t ws_bytes,
                                float ws_blur, int N, int V, int F, int H, int R, int atlas_batch, float* grad_atlas,
                                const AcfmRasterTuning* tuning, void* stream);

/* ---- loss combination ------------------------------------------------------------------
 * replaces the elementwise tail of the trainer's total loss (multiframe/main.py:716-746, 749-765:
 * weight * term + ... then .mean() over the batch): total = (1/N) sum_n sum_t sum_c w[t][c] T_t[n,c]
 * for up to 4 terms T_t [N, cols[t]] (cols <= 4, dense row-major), one launch each way.
 * terms / grads / cols / weights are HOST arrays (device pointers inside terms and grads; weights
 * flattened term by term); a NULL grads[t] skips that term. */
int acfm_combine_losses(const void* const* terms, const int* cols, const float* weights, int nterms, int N,
                        float* total, void* stream);
int acfm_combine_losses_backward(const float* grad_total, void* const* grads, const int* cols,
                                 const float* weights, int nterms, int N, void* stream);

/* ---- total per hypothesis + hypothesis weighting --------------------------------------------------
 * replaces the per-hypothesis total of ShapeTrainer.forward and its softmax weighting (multiframe/main.py:716-746:
 * weight * term + ..., probs = softmax(-total, dim 0).detach(), (total * probs).sum(0).mean()) by one launch each way:
 *   total[g,n] = sum_t w_t T_t[g,n]  for nterms <= 8 terms T_t [G*N] f32 (row g*N + n);
 *   probs[g,n] = softmax over g of -total[.,n];   out[0] = (1/N) sum_n sum_g probs total (the weighted loss),
 *   out[1] = mean total (what the reference logs as camera_loss), out[2..3] = means of the two auxiliary sums
 *   aux_k[g,n] = sum_{t: aux_group[t] == k} aux_weights[t] T_t[g,n] (logged terms such as sil_cons; aux0 / aux1 and
 *   aux_group / aux_weights may be NULL), out[4 + t] = mean of term t.  out: 12 floats.
 * backward: grads[t][g,n] = grad_weighted * w_t * probs[g,n] / N (NULL grads[t] skipped); probs carry no gradient.
 * terms / grads / weights / aux_* are HOST arrays (device pointers inside terms and grads). */
int acfm_hypothesis_total(const void* const* terms, const float* weights, const int* aux_group, const float* aux_weights,
                          int nterms, int G, int N, float* total, float* probs, float* aux0, float* aux1, float* out,
                          void* stream);
int acfm_hypothesis_total_backward(const float* grad_weighted, const float* probs, const float* weights, int nterms,
                                   int G, int N, void* const* grads, void* stream);

/* ---- texture temporal-consistency term ------------------------------------------------------------
 * replaces multiframe/main.py:705-711 as written there: the per-frame atlases textures [B*T,F,R,R,3] regrouped as
 * [B,F,R,R,T,3], that buffer reshaped to rows [-1,R,R], loss = mean over (row block m, i < R-1) of the L2 norm over
 * j of X[m,i,j] - X[m,i+1,j] (zero norms contribute no gradient, as torch.norm's backward).
 * scratch: acfm_texture_cycle_scratch_floats(B,T,F,R) floats (the norms, kept for the backward, + partial sums);
 * any B, T, F; R >= 2.  Sums in a fixed order: reproducible. */
size_t acfm_texture_cycle_scratch_floats(int B, int T, int F, int R);
int acfm_texture_cycle(const float* textures, int B, int T, int F, int R, float* scratch, float* loss, void* stream);
int acfm_texture_cycle_backward(const float* textures, const float* scratch, const float* grad_loss, int B, int T, int F,
                                int R, float* grad_textures, void* stream);

/* ---- fused silhouette losses ---------------------------------------------------------
 * replaces loss_utils.l1_loss / iou / iou_loss / edt_loss with reduce=False
 * (multiframe/nnutils/loss_utils.py:18-32, 72-77, 245-253) in one pass over the mask:
 *   out [N,4] = (mean|m-gt|, sum m*gt, sum (m+gt-m*gt), mean edt*m); gt / edt may be NULL.
 * ref_batch (here and in the losses below): the number of distinct references; prediction n is
 * compared with reference n % ref_batch (gt, edt [ref_batch,HW]).  The trainer scores G camera
 * hypotheses per frame against the frame's one ground truth (masks.repeat(G, 1, 1) at
 * main.py:472-479, 644-662): ref_batch = N / G spares those copies; ref_batch = N is the plain case. */
int acfm_mask_losses(const float* mask, const float* gt, const float* edt, int N, int HW, int ref_batch,
                     float* out, void* stream);
/* grad_mask [N,HW] = w_l1[n]*sign(m-gt)/HW + w_edt[n]*edt/HW + IoU term
 * (w_inter[n]*gt + w_union[n]*(1-gt)); weights are per-mesh upstream gradients [N,4]. */
int acfm_mask_losses_backward(const float* mask, const float* gt, const float* edt,
                              const float* grad_out, int N, int HW, int ref_batch, float* grad_mask,
                              void* stream);

/* ---- masked texture MSE ---------------------------------------------------------------
 * replaces the inline texture term of multiframe/main.py:655-662,
 * F.mse_loss(texture_pred * mask, imgs * mask, reduction='none').mean((1,2,3)):
 *   tex [N,3,HW], img [ref_batch,3,HW], mask [ref_batch,HW] f32 -> out [N]; backward -> grad_tex [N,3,HW]. */
int acfm_tex_mse(const float* tex, const float* img, const float* mask, int N, int HW, int ref_batch,
                 float* out, void* stream);
int acfm_tex_mse_backward(const float* tex, const float* img, const float* mask,
                          const float* grad_out, int N, int HW, int ref_batch, float* grad_tex, void* stream);

/* ---- visibility + boundary loss ------------------------------------------------------
 * visible-vertex bitmap shared by loss_utils.bds_loss (:214-224) and optical_flow_loss
 * (:432-443): vis [N,V] u8 = 1 for every vertex of a face that appears in
 * pix_to_face[..., 0].  pix_to_face has `K` ids per pixel (only slot 0 is read). */
int acfm_visible_vertices(const int64_t* pix_to_face, const int64_t* faces, int N, int V, int F,
                          int HW, int K, uint8_t* vis, void* stream);
/* loss_utils.bds_loss (:204-237) given vis: for each boundary point the squared distance to
 * the nearest visible projected vertex (1000 where none), times the point's valid flag,
 * summed per mesh.  verts_xy [N,V,2], bds [ref_batch,P,3] -> loss [N], argmin [N,P] i32 (saved). */
int acfm_bds_loss(const float* verts_xy, const float* bds, const uint8_t* vis, int N, int V, int P,
                  int ref_batch, float* loss, int32_t* argmin, void* stream);
int acfm_bds_loss_backward(const float* verts_xy, const float* bds, const int32_t* argmin,
                           const float* grad_loss, int N, int V, int P, int ref_batch, float* grad_verts_xy,
                           void* stream);

/* ---- mesh priors -------------------------------------------------------------------------
 * Packed meshes: verts [P,3] f32, faces [F,3] / edges [E,2] i64 with packed vertex ids.
 * acfm_cot_laplacian: geom_utils.mesh_laplacian(mesh, 'cot') (multiframe/nnutils/geom_utils.py:
 *   158-324) for ONE mesh: dense L [V,V] = W - diag(rowsum W), W_ij = (cot a_ij + cot b_ij)/4. */
int acfm_cot_laplacian(const float* verts, const int64_t* faces, int V, int F, float* L, void* stream);
/* acfm_laplacian_smoothing: pytorch3d.loss.mesh_laplacian_smoothing (multiframe/main.py:699-704):
 *   method 0 'cot' (conn = faces [F,3]): loss = sum_v vweight[v] * |(W v)_v / rowsum_v - v_v|;
 *   method 1 'uniform' (conn = unique edges [F,2]): W_ij = 1 on edges.
 *   vweight [P] = 1 / (verts of the vertex's mesh); the caller divides the sum by the mesh count.
 *   verts_per_mesh / faces_per_mesh > 0 (method 0): the packed arrays are equal-sized meshes one after the
 *   other (mesh m = vertices [m vpm, (m+1) vpm), faces [m fpm, (m+1) fpm)): one workgroup per mesh accumulates
 *   in LDS, one launch each way; 0 = unknown layout (global atomics).
 *   loss: 1 float (device); state: acfm_laplacian_smoothing_state_floats(P, F) floats kept for
 *   the backward, which takes the upstream gradient as a DEVICE scalar. */
size_t acfm_laplacian_smoothing_state_floats(int P, int F);
int acfm_laplacian_smoothing(const float* verts, const int64_t* conn, const float* vweight, int P, int F,
                             int method, int verts_per_mesh, int faces_per_mesh, float* loss, float* state,
                             void* stream);
int acfm_laplacian_smoothing_backward(const int64_t* conn, const float* state, const float* grad_loss, int P,
                                      int F, int method, int verts_per_mesh, int faces_per_mesh,
                                      float* grad_verts, void* stream);
/* acfm_edge_rigidity: loss_utils.locally_rigid_fn (multiframe/nnutils/loss_utils.py:150-164):
 *   loss = sum_e (|v[e0]-v[e1]| - |vt[et0]-vt[et1]|)^2 (the caller divides by the mesh count). */
int acfm_edge_rigidity(const float* verts, const int64_t* edges, const float* verts_t, const int64_t* edges_t,
                       int E, float* loss, void* stream);
int acfm_edge_rigidity_backward(const float* verts, const int64_t* edges, const float* verts_t,
                                const int64_t* edges_t, int E, int P, int Pt, int verts_per_mesh,
                                const float* grad_loss, float* grad_verts, float* grad_verts_t, void* stream);
/* verts_per_mesh > 0: equal-sized meshes and `edges` sorted by its first vertex (Meshes.edges_packed()):
 * one workgroup per mesh accumulates in LDS (used when only grad_verts is asked for); 0: unknown layout. */

/* ---- on-device input preparation ----------------------------------------
 * replaces the per-batch CPU work of ShapeTrainer.set_input (multiframe/main.py:365-377) and
 * its device->host->device round trip of the masks.
 * acfm_edt: scipy.ndimage.distance_transform_edt(1 - mask) / divisor for every mask of the batch
 *   (multiframe/utils/image.py:94-102; divisor = 1 for norm=False, max(H,W) for norm=True):
 *   Euclidean distance of every pixel to the nearest pixel with mask == 1 (0 on those pixels).
 *   Exact integer squared distances, sqrt in fp64, one rounding to fp32.
 *   mask [N,H,W] f32 (0/1) -> out [N,H,W] f32. */
size_t acfm_edt_workspace_bytes(int N, int H, int W);
int acfm_edt(const float* mask, int N, int H, int W, int divisor, float* out, void* ws, size_t ws_bytes,
             void* stream);
/* acfm_boundaries: skimage.segmentation.find_boundaries(mask) (mode='thick', connectivity=1) +
 *   the point list of compute_boundaries (multiframe/utils/image.py:122-146): for every mask
 *   the boundary pixels in row-major order as (x, y, valid) with x = (col/W - 0.5)*2,
 *   y = (row/H - 0.5)*2; rows beyond the count are (-1, -1, 0) like the reference's padding.
 *   mask [N,H,W] f32 -> out [N,cap,3] f32 (first min(count, cap) points), counts [N] i32. */
int acfm_boundaries(const float* mask, int N, int H, int W, int cap, float* out, int* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACFM_HIP_H_ */
