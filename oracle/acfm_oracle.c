/*
 * acfm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic on ACFM's render hot path, used only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker for the HIP kernels.  Nothing in acfm_video_3d_reconstruction_amd/
 * may import, link or call this file.
 *
 * What it restates (reference = /root/reference, fkokkinos/acfm_video_3d_reconstruction):
 *   - weak-perspective projection     multiframe/nnutils/geom_utils.py:62-79, 107-152
 *   - NeuralRenderer camera chain     multiframe/nnutils/nmr.py:143-149 (y flip, R[0,0]=-1, T=(0,0,2.732))
 *   - the rasterizer / blending / atlas arithmetic that the reference delegates to the
 *     third-party dependency PyTorch3D == 0.3.0 (README.md:12; call sites nmr.py:152-200,
 *     224-238).  PyTorch3D's source is NOT in /root/reference; the algorithm restated here
 *     is its published naive CPU rasterizer (rasterize_meshes_cpu.cpp: RasterizeMeshesNaiveCpu,
 *     RasterizeMeshesBackwardCpu), geometry_utils.h (EdgeFunctionForward,
 *     BarycentricCoordinatesForward, BarycentricClipForward, PointLineDistanceForward/Backward,
 *     PointTriangleDistanceForward/Backward), blending.py (sigmoid_alpha_blend,
 *     softmax_rgb_blend) and TexturesAtlas.sample_textures, as summarised in SURVEY.md App-A.
 *
 * PARITY STATUS: the projection / losses / solve parts are pinned by golden vectors
 * generated from the importable reference modules (tests/golden/make_golden.py).
 * The RASTERIZER part is "parity unpinned": the reference ships no tests or fixtures for
 * it and PyTorch3D cannot be imported here, so it is anchored on analytic known-answer
 * tests (tests/test_oracle_raster.py) only.
 *
 * Numerics: fp32 throughout, compiled with -ffp-contract=off so that every multiply and
 * add rounds separately, exactly like the chain of separate torch kernels in the
 * reference.  The HIP kernels are compiled the same way; face indices must match bit for bit.
 *
 * Tie-break for equal depth (implementation-defined in PyTorch3D): smaller face id wins.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define K_EPS 1e-8f
#define MAXK 64

/* number of host threads for the parallel loops (used by bench.py's cpu_baseline leg) */
int oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* ------------------------------------------------------------------ projection */
/* geom_utils.hamilton_product (geom_utils.py:107-131): same operand order, same
 * left-to-right evaluation as the Python expression. */
static void hamilton(const float a[4], const float b[4], float o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

/* geom_utils.orthographic_proj_withz (geom_utils.py:62-79) + quat_rotate (:134-152).
 * verts [N,V,3], cams [N,7] = (s, tx, ty, q0..q3) -> out [N,V,3]. */
void oracle_project(const float* verts, const float* cams, int N, int V, float offset_z,
                    float* out) {
  for (int n = 0; n < N; ++n) {
    const float* c = cams + 7 * n;
    const float q[4] = {c[3], c[4], c[5], c[6]};
    const float qc[4] = {c[3], -1.0f * c[4], -1.0f * c[5], -1.0f * c[6]};
    for (int v = 0; v < V; ++v) {
      const float* x = verts + ((size_t)n * V + v) * 3;
      const float X[4] = {x[0] * 0.0f, x[0], x[1], x[2]};
      float t[4], r[4];
      hamilton(X, qc, t);
      hamilton(q, t, r);
      float* o = out + ((size_t)n * V + v) * 3;
      o[0] = c[0] * r[1] + c[1];
      o[1] = c[0] * r[2] + c[2];
      o[2] = c[0] * r[3] + offset_z;
    }
  }
}

/* ------------------------------------------------------------------ geometry helpers
 * (PyTorch3D geometry_utils.h semantics) */
static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

static inline float point_line_dist(float px, float py, float ax, float ay, float bx, float by) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= K_EPS) {
    const float dx = px - bx, dy = py - by;
    return dx * dx + dy * dy;
  }
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float qx = ax + t * bax, qy = ay + t * bay;
  const float dx = qx - px, dy = qy - py;
  return dx * dx + dy * dy;
}

static inline float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

typedef struct {
  float z;
  int64_t f;
  float d;
  float b0, b1, b2;
} cand_t;

/* (z, f) lexicographic "a before b" */
static inline int cand_lt(const cand_t* a, const cand_t* b) {
  return (a->z < b->z) || (a->z == b->z && a->f < b->f);
}

/* ------------------------------------------------------------------ naive rasterizer
 * face_verts [N*F,3,3] (NDC x, NDC y, view z), packed ids n*F+f.
 * Outputs: pix_to_face int64 [N,H,W,K], zbuf/dists f32 [N,H,W,K], bary f32 [N,H,W,K,3];
 * unused slots are -1.  (RasterizeMeshesNaiveCpu) */
void oracle_rasterize(const float* face_verts, int N, int F, int H, int W, int K,
                      float blur_radius, int clip_bary, int64_t* pix_to_face, float* zbuf,
                      float* bary, float* dists) {
  const float margin = sqrtf(blur_radius);
  /* per-face area and blur-expanded bounding box, computed once (ComputeFaceAreas /
   * ComputeFaceBoundingBoxes in PyTorch3D's CPU path) */
  float* pre = (float*)malloc(sizeof(float) * (size_t)N * F * 5);
#pragma omp parallel for
  for (size_t i = 0; i < (size_t)N * F; ++i) {
    const float* fv = face_verts + i * 9;
    pre[i * 5 + 0] = edge_fn(fv[6], fv[7], fv[0], fv[1], fv[3], fv[4]);
    pre[i * 5 + 1] = min3f(fv[0], fv[3], fv[6]) - margin;
    pre[i * 5 + 2] = max3f(fv[0], fv[3], fv[6]) + margin;
    pre[i * 5 + 3] = min3f(fv[1], fv[4], fv[7]) - margin;
    pre[i * 5 + 4] = max3f(fv[1], fv[4], fv[7]) + margin;
  }
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int n = 0; n < N; ++n) {
    for (int yi = 0; yi < H; ++yi) {
      const float yf = pix_to_ndc(H - 1 - yi, H);
      for (int xi = 0; xi < W; ++xi) {
        const float xf = pix_to_ndc(W - 1 - xi, W);
        cand_t best[MAXK];
        int cnt = 0;
        for (int f = 0; f < F; ++f) {
          const float* pf = pre + ((size_t)n * F + f) * 5;
          const float area = pf[0];
          if (area <= K_EPS && area >= -1.0f * K_EPS) continue;
          if (xf > pf[2] || xf < pf[1] || yf > pf[4] || yf < pf[3]) continue;
          const float* fv = face_verts + ((size_t)n * F + f) * 9;
          const float x0 = fv[0], y0 = fv[1], z0 = fv[2];
          const float x1 = fv[3], y1 = fv[4], z1 = fv[5];
          const float x2 = fv[6], y2 = fv[7], z2 = fv[8];
          const float denom = area + K_EPS;
          const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / denom;
          const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / denom;
          const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / denom;
          float c0 = w0, c1 = w1, c2 = w2;
          if (clip_bary) {
            c0 = fmaxf(fminf(w0, 1.0f), 0.0f);
            c1 = fmaxf(fminf(w1, 1.0f), 0.0f);
            c2 = fmaxf(fminf(w2, 1.0f), 0.0f);
            const float s = fmaxf(c0 + c1 + c2, 1e-5f);
            c0 = c0 / s;
            c1 = c1 / s;
            c2 = c2 / s;
          }
          const float pz = c0 * z0 + c1 * z1 + c2 * z2;
          if (pz < 0.0f) continue;
          const float d01 = point_line_dist(xf, yf, x0, y0, x1, y1);
          const float d02 = point_line_dist(xf, yf, x0, y0, x2, y2);
          const float d12 = point_line_dist(xf, yf, x1, y1, x2, y2);
          const float d = fminf(fminf(d01, d02), d12);
          const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
          if (!inside && d >= blur_radius) continue;
          cand_t c;
          c.z = pz;
          c.f = (int64_t)n * F + f;
          c.d = inside ? -d : d;
          c.b0 = c0;
          c.b1 = c1;
          c.b2 = c2;
          /* sorted insert, keep the K smallest (z, f) */
          int pos = cnt;
          if (cnt == K) {
            if (!cand_lt(&c, &best[K - 1])) continue;
            pos = K - 1;
          } else {
            cnt++;
          }
          while (pos > 0 && cand_lt(&c, &best[pos - 1])) {
            best[pos] = best[pos - 1];
            pos--;
          }
          best[pos] = c;
        }
        const size_t base = (((size_t)n * H + yi) * W + xi) * K;
        for (int k = 0; k < K; ++k) {
          if (k < cnt) {
            pix_to_face[base + k] = best[k].f;
            zbuf[base + k] = best[k].z;
            dists[base + k] = best[k].d;
            bary[(base + k) * 3 + 0] = best[k].b0;
            bary[(base + k) * 3 + 1] = best[k].b1;
            bary[(base + k) * 3 + 2] = best[k].b2;
          } else {
            pix_to_face[base + k] = -1;
            zbuf[base + k] = -1.0f;
            dists[base + k] = -1.0f;
            bary[(base + k) * 3 + 0] = -1.0f;
            bary[(base + k) * 3 + 1] = -1.0f;
            bary[(base + k) * 3 + 2] = -1.0f;
          }
        }
      }
    }
  }
  free(pre);
}

/* PointLineDistanceBackward: clamped t treated as a constant. */
static inline void point_line_dist_bwd(float px, float py, float ax, float ay, float bx, float by,
                                       float g, float* ga, float* gb) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= K_EPS) {
    /* d = |p - b|^2 : grad_b = -2 (p - b) g */
    gb[0] += -2.0f * (px - bx) * g;
    gb[1] += -2.0f * (py - by) * g;
    return;
  }
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float qx = (1.0f - t) * ax + t * bx, qy = (1.0f - t) * ay + t * by;
  const float ex = 2.0f * (qx - px), ey = 2.0f * (qy - py);
  ga[0] += g * (1.0f - t) * ex;
  ga[1] += g * (1.0f - t) * ey;
  gb[0] += g * t * ex;
  gb[1] += g * t * ey;
}

/* RasterizeMeshesBackwardCpu restricted to the dists path (the silhouette shader sends
 * gradient only through dists).  grad_dists [N,H,W,K] ->
 * grad_face_verts [N*F,3,3] (only the xy entries are ever non-zero).  Serial on purpose:
 * the summation order (n, y, x, k) is the oracle's definition. */
void oracle_rasterize_backward_dists(const float* face_verts, const int64_t* pix_to_face,
                                     const float* grad_dists, int N, int F, int H, int W, int K,
                                     float* grad_face_verts) {
  memset(grad_face_verts, 0, sizeof(float) * (size_t)N * F * 9);
  for (int n = 0; n < N; ++n)
    for (int yi = 0; yi < H; ++yi) {
      const float yf = pix_to_ndc(H - 1 - yi, H);
      for (int xi = 0; xi < W; ++xi) {
        const float xf = pix_to_ndc(W - 1 - xi, W);
        const size_t base = (((size_t)n * H + yi) * W + xi) * K;
        for (int k = 0; k < K; ++k) {
          const int64_t f = pix_to_face[base + k];
          if (f < 0) break;
          const float gup = grad_dists[base + k];
          const float* fv = face_verts + (size_t)f * 9;
          float* gf = grad_face_verts + (size_t)f * 9;
          const float x0 = fv[0], y0 = fv[1], x1 = fv[3], y1 = fv[4], x2 = fv[6], y2 = fv[7];
          const float area = edge_fn(x2, y2, x0, y0, x1, y1);
          const float denom = area + K_EPS;
          const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / denom;
          const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / denom;
          const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / denom;
          const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
          const float g = inside ? -gup : gup;
          const float d01 = point_line_dist(xf, yf, x0, y0, x1, y1);
          const float d02 = point_line_dist(xf, yf, x0, y0, x2, y2);
          const float d12 = point_line_dist(xf, yf, x1, y1, x2, y2);
          if (d01 <= d02 && d01 <= d12)
            point_line_dist_bwd(xf, yf, x0, y0, x1, y1, g, gf + 0, gf + 3);
          else if (d02 <= d01 && d02 <= d12)
            point_line_dist_bwd(xf, yf, x0, y0, x2, y2, g, gf + 0, gf + 6);
          else
            point_line_dist_bwd(xf, yf, x1, y1, x2, y2, g, gf + 3, gf + 6);
        }
      }
    }
}

/* ------------------------------------------------------------------ blending */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* sigmoid_alpha_blend (SoftSilhouetteShader), SURVEY App-A.5:
 * mask = 1 - prod_k (1 - sigmoid(-dists_k / sigma) * [f_k >= 0]). */
void oracle_sigmoid_alpha_blend(const int64_t* pix_to_face, const float* dists, size_t P, int K,
                                float sigma, float* mask) {
#pragma omp parallel for
  for (size_t p = 0; p < P; ++p) {
    float alpha = 1.0f;
    for (int k = 0; k < K; ++k) {
      const float prob = (pix_to_face[p * K + k] >= 0) ? sigmoidf_(-dists[p * K + k] / sigma) : 0.0f;
      alpha = alpha * (1.0f - prob);
    }
    mask[p] = 1.0f - alpha;
  }
}

/* d mask / d dists_k = -prod_{j != k}(1 - p_j) * p_k (1 - p_k) / sigma  (exact product form,
 * evaluated in double so the oracle gradient is the reference value the fp32 kernels are
 * compared with). */
void oracle_sigmoid_alpha_blend_backward(const int64_t* pix_to_face, const float* dists,
                                         const float* grad_mask, size_t P, int K, float sigma,
                                         float* grad_dists) {
#pragma omp parallel for
  for (size_t p = 0; p < P; ++p) {
    double pr[MAXK];
    for (int k = 0; k < K; ++k)
      pr[k] = (pix_to_face[p * K + k] >= 0)
                  ? 1.0 / (1.0 + exp((double)(dists[p * K + k] / sigma)))
                  : 0.0;
    for (int k = 0; k < K; ++k) {
      double others = 1.0;
      for (int j = 0; j < K; ++j)
        if (j != k) others *= (1.0 - pr[j]);
      grad_dists[p * K + k] =
          (pix_to_face[p * K + k] >= 0)
              ? (float)(-(double)grad_mask[p] * others * pr[k] * (1.0 - pr[k]) / (double)sigma)
              : 0.0f;
    }
  }
}

/* ------------------------------------------------------------------ texture branch (K = 1)
 * TexturesAtlas.sample_textures + ambient-only Phong (colour = texel) + softmax_rgb_blend
 * with K = 1.  atlas [N*F,R,R,3]; outputs rgb [N,H,W,3], sil [N,H,W],
 * texel_idx int32 [N,H,W] = linear index into atlas/3 of the sampled texel (-1 = empty). */
void oracle_atlas_shade(const int64_t* pix_to_face, const float* zbuf, const float* bary,
                        const float* dists, const float* atlas, size_t P, int R, float sigma,
                        float gamma, float* rgb, float* sil, int32_t* texel_idx) {
  const float znear = 1.0f, zfar = 100.0f, eps = 1e-10f;
#pragma omp parallel for
  for (size_t p = 0; p < P; ++p) {
    const int64_t f = pix_to_face[p];
    if (f < 0) {
      /* prob = 0, z_inv_max = eps, delta = exp(0) = 1 -> rgb = bg = 0 */
      rgb[p * 3 + 0] = rgb[p * 3 + 1] = rgb[p * 3 + 2] = 0.0f;
      sil[p] = 0.0f;
      texel_idx[p] = -1;
      continue;
    }
    const float w0 = bary[p * 3 + 0], w1 = bary[p * 3 + 1];
    int ix = (int)(w0 * (float)R), iy = (int)(w1 * (float)R);
    const int below = ((w0 + w1) * (float)R - ((float)ix + (float)iy)) <= 1.0f;
    if (!below) {
      ix = R - 1 - ix;
      iy = R - 1 - iy;
    }
    /* PyTorch3D would index out of range for a barycentric exactly 1.0; clamp instead. */
    ix = ix < 0 ? 0 : (ix > R - 1 ? R - 1 : ix);
    iy = iy < 0 ? 0 : (iy > R - 1 ? R - 1 : iy);
    const size_t ti = (((size_t)f * R + iy) * R + ix);
    const float prob = sigmoidf_(-dists[p] / sigma);
    const float z_inv = (zfar - zbuf[p]) / (zfar - znear);
    const float z_inv_max = fmaxf(z_inv, eps);
    const float wnum = prob * expf((z_inv - z_inv_max) / gamma);
    const float delta = fmaxf(expf((eps - z_inv_max) / gamma), eps);
    const float den = wnum + delta;
    for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = (wnum * atlas[ti * 3 + c] + delta * 0.0f) / den;
    sil[p] = 1.0f - (1.0f - prob);
    texel_idx[p] = (int32_t)ti;
  }
}
