/* oracle/orp_oracle2.c -- CPU ORACLE part 2 (TEST INFRASTRUCTURE ONLY): the small per-element ops.
 *   pointsJf          mmdet/ops/point_justify/src/points_justify_kernel.cu:25-102
 *   ChamferDistance2D mmdet/ops/chamfer_2d/src/chamfer_2d.cu:12-124 (forward), :145-158 (backward)
 *   sigmoid focal     mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:23-97
 * Pinned against oracle/_ref (the reference kernels run under a 1-thread launch emulation) in
 * tests/test_oracle_vs_ref.py.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* ray casting with the reference's early-`break` semantics (points_justify_kernel.cu:51-99) */
static float point_in_quad(float px, float py, const float* q) {
  int ncross = 0;
  int i, j;
  for (i = 0, j = 3; i < 4; j = i, i++) {
    float sx = q[2 * i], sy = q[2 * i + 1], tx = q[2 * j], ty = q[2 * j + 1];
    if (py < (sy < ty ? sy : ty)) continue;
    if (py > (sy > ty ? sy : ty)) continue;
    if ((sx == px && sy == py) || (tx == px && ty == py)) break;
    if ((sy < py && ty >= py) || (sy >= py && ty < py)) {
      float x = sx + (py - sy) * (tx - sx) / (ty - sy);
      if (x == px) break;
      if (x > px) ncross++;
    }
  }
  return (ncross % 2 == 1) ? 1.0f : 0.0f;
}
void orc_points_justify(const float* points, int m, const float* polys, int k, float* out) {
  for (int r = 0; r < m; r++)
    for (int c = 0; c < k; c++) out[(size_t)r * k + c] = point_in_quad(points[2 * r], points[2 * r + 1], polys + 8 * c);
}
/* aligned form = the diagonal the SpatialBorderLoss reads (spatial_border_loss.py:24-67) */
void orc_points_in_quad_aligned(const float* pts18, const float* quads, int m, float* out9) {
  for (int r = 0; r < m; r++)
    for (int t = 0; t < 9; t++)
      out9[(size_t)r * 9 + t] = point_in_quad(pts18[(size_t)r * 18 + 2 * t], pts18[(size_t)r * 18 + 2 * t + 1], quads + 8 * (size_t)r);
}

/* nearest neighbour, first minimum wins (strict <); chamfer_2d.cu:21-121 for the sizes the head uses.
 * NB the reference's `end_ka = end_k - (end_k & 2)` reads one stale LDS slot when m % 4 == 3 -- undefined there,
 * the mathematically defined nearest neighbour here. */
void orc_chamfer_nn(const float* xyz, const float* xyz2, int b, int n, int m, float* result, int32_t* result_i) {
  for (int bi = 0; bi < b; bi++)
    for (int j = 0; j < n; j++) {
      float x1 = xyz[((size_t)bi * n + j) * 2], y1 = xyz[((size_t)bi * n + j) * 2 + 1];
      float best = 0; int best_i = 0;
      for (int kk = 0; kk < m; kk++) {
        float x2 = xyz2[((size_t)bi * m + kk) * 2] - x1, y2 = xyz2[((size_t)bi * m + kk) * 2 + 1] - y1;
        float d = x2 * x2 + y2 * y2;
        if (kk == 0 || d < best) { best = d; best_i = kk; }
      }
      result[(size_t)bi * n + j] = best; result_i[(size_t)bi * n + j] = best_i;
    }
}
/* chamfer_2d.cu:145-158; accumulates into grad_xyz1 / grad_xyz2 (caller zeroes them) */
void orc_chamfer_grad(const float* xyz1, const float* xyz2, int b, int n, int m, const float* grad_dist1,
                      const int32_t* idx1, float* grad_xyz1, float* grad_xyz2) {
  for (int bi = 0; bi < b; bi++)
    for (int j = 0; j < n; j++) {
      size_t o1 = ((size_t)bi * n + j) * 2;
      int j2 = idx1[(size_t)bi * n + j];
      size_t o2 = ((size_t)bi * m + j2) * 2;
      float g = grad_dist1[(size_t)bi * n + j] * 2;
      grad_xyz1[o1] += g * (xyz1[o1] - xyz2[o2]);
      grad_xyz1[o1 + 1] += g * (xyz1[o1 + 1] - xyz2[o2 + 1]);
      grad_xyz2[o2] += -(g * (xyz1[o1] - xyz2[o2]));
      grad_xyz2[o2 + 1] += -(g * (xyz1[o1 + 1] - xyz2[o2 + 1]));
    }
}

/* sigmoid focal loss, expression structure (float/double mix) as written in the reference */
void orc_focal_forward(const float* logits, const int64_t* targets, int num, int classes, float gamma, float alpha,
                       float* losses) {
  for (long i = 0; i < (long)num * classes; i++) {
    int n = (int)(i / classes), d = (int)(i % classes);
    int t = (int)targets[n];
    float c1 = (t == (d + 1));
    float c2 = (t >= 0 & t != (d + 1));
    float zn = (float)(1.0 - alpha), zp = alpha;
    float x = logits[i];
    float p = (float)(1. / (1. + expf(-x)));
    float term1 = powf((float)(1. - p), gamma) * logf(p > FLT_MIN ? p : FLT_MIN);
    float term2 = (float)(powf(p, gamma) * (-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))));
    float l = 0.0f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}
void orc_focal_backward(const float* logits, const int64_t* targets, const float* d_losses, int num, int classes,
                        float gamma, float alpha, float* d_logits) {
  for (long i = 0; i < (long)num * classes; i++) {
    int n = (int)(i / classes), d = (int)(i % classes);
    int t = (int)targets[n];
    float c1 = (t == (d + 1));
    float c2 = (t >= 0 & t != (d + 1));
    float zn = (float)(1.0 - alpha), zp = alpha;
    float x = logits[i];
    float p = (float)(1. / (1. + expf(-x)));
    float term1 = (float)(powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf(p > FLT_MIN ? p : FLT_MIN))));
    float term2 = (float)(powf(p, gamma) *
                          ((-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))) * (1. - p) * gamma - p));
    float g = 0.0f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}

/* ------------------------------------------------------------------------------------------------------- */
/* a2: deformable convolution forward                                                                        */
/*   bilinear sampling = deformable_im2col_bilinear (deform_conv_cuda_kernel.cu:84-115) in float, zero unless  */
/*   -1 < h < H and -1 < w < W (:229); contraction out = W . im2col (deform_conv_cuda.cpp:222-237) accumulated  */
/*   here in DOUBLE so that the oracle is the mathematically tight value for any GEMM summation order.         */
/*   mask != NULL -> DCNv2 modulation (:620-640), bias != NULL -> + bias (deform_conv_cuda.cpp:560-566).        */
/* ------------------------------------------------------------------------------------------------------- */
static float dcn_bilinear(const float* bottom, int data_width, int height, int width, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = bottom[h_low * data_width + w_low];
  if (h_low >= 0 && w_high <= width - 1) v2 = bottom[h_low * data_width + w_high];
  if (h_high <= height - 1 && w_low >= 0) v3 = bottom[h_high * data_width + w_low];
  if (h_high <= height - 1 && w_high <= width - 1) v4 = bottom[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* col [C*kh*kw][B][Ho][Wo]  (deformable_im2col_gpu_kernel :190-243) */
void orc_dcn_im2col(const float* im, const float* offset, int B, int C, int H, int W, int kh, int kw, int pad_h,
                    int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* col) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int cpdg = C / dg;
  for (int c = 0; c < C; c++)
    for (int b = 0; b < B; b++)
      for (int ho = 0; ho < Ho; ho++)
        for (int wo = 0; wo < Wo; wo++) {
          const float* imp = im + ((size_t)b * C + c) * H * W;
          const float* op = offset + ((size_t)b * dg + c / cpdg) * 2 * kh * kw * Ho * Wo;
          for (int i = 0; i < kh; i++)
            for (int j = 0; j < kw; j++) {
              float oh = op[((size_t)(2 * (i * kw + j)) * Ho + ho) * Wo + wo];
              float ow = op[((size_t)(2 * (i * kw + j) + 1) * Ho + ho) * Wo + wo];
              float h_im = (ho * stride_h - pad_h) + i * dil_h + oh;
              float w_im = (wo * stride_w - pad_w) + j * dil_w + ow;
              float val = 0;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = dcn_bilinear(imp, W, H, W, h_im, w_im);
              col[((((size_t)c * kh * kw + i * kw + j) * B + b) * Ho + ho) * Wo + wo] = val;
            }
        }
}

void orc_dcn_forward(const float* x, const float* offset, const float* mask, const float* weight, const float* bias,
                     float* out, int B, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w,
                     int pad_h, int pad_w, int dil_h, int dil_w, int groups, int dg) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int taps = kh * kw, cpg = C / groups, opg = Cout / groups, cpdg = C / dg;
  /* sampled values once per (b, c, tap, ho, wo) */
  float* samp = (float*)malloc(sizeof(float) * (size_t)C * taps * Ho * Wo);
  for (int b = 0; b < B; b++) {
    for (int c = 0; c < C; c++) {
      const float* imp = x + ((size_t)b * C + c) * H * W;
      const float* op = offset + ((size_t)b * dg + c / cpdg) * 2 * taps * Ho * Wo;
      const float* mp = mask ? mask + ((size_t)b * dg + c / cpdg) * taps * Ho * Wo : 0;
      for (int t = 0; t < taps; t++)
        for (int ho = 0; ho < Ho; ho++)
          for (int wo = 0; wo < Wo; wo++) {
            int i = t / kw, j = t % kw;
            float oh = op[((size_t)(2 * t) * Ho + ho) * Wo + wo], ow = op[((size_t)(2 * t + 1) * Ho + ho) * Wo + wo];
            float h_im = (ho * stride_h - pad_h) + i * dil_h + oh;
            float w_im = (wo * stride_w - pad_w) + j * dil_w + ow;
            float val = 0;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = dcn_bilinear(imp, W, H, W, h_im, w_im);
            if (mp) val = val * mp[((size_t)t * Ho + ho) * Wo + wo];
            samp[(((size_t)c * taps + t) * Ho + ho) * Wo + wo] = val;
          }
    }
    for (int o = 0; o < Cout; o++) {
      int g = o / opg;
      for (int ho = 0; ho < Ho; ho++)
        for (int wo = 0; wo < Wo; wo++) {
          double acc = 0;
          for (int cc = 0; cc < cpg; cc++)
            for (int t = 0; t < taps; t++)
              acc += (double)weight[((size_t)o * cpg + cc) * taps + t] *
                     (double)samp[(((size_t)(g * cpg + cc)) * taps + t) * Ho * Wo + (size_t)ho * Wo + wo];
          if (bias) acc += bias[o];
          out[(((size_t)b * Cout + o) * Ho + ho) * Wo + wo] = (float)acc;
        }
    }
  }
  free(samp);
}

/* ------------------------------------------------------------------------------------------------------- */
/* a2 backward: grad wrt input / offset from a column gradient, and the reference's two helper weights       */
/*   get_gradient_weight (deform_conv_cuda_kernel.cu:117-143), get_coordinate_weight (:145-188),              */
/*   deformable_col2im_gpu_kernel (:279-335), deformable_col2im_coord_gpu_kernel (:373-436).                  */
/* ------------------------------------------------------------------------------------------------------- */
static float dcn_coordinate_weight(float argmax_h, float argmax_w, int height, int width, const float* im, int data_width, int bp_dir) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int hl = (int)floorf(argmax_h), wl = (int)floorf(argmax_w), hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - argmax_w) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += -1 * (argmax_w - wl) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - argmax_w) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_w - wl) * im[hh * data_width + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - argmax_h) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - argmax_h) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += -1 * (argmax_h - hl) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_h - hl) * im[hh * data_width + wh];
  }
  return weight;
}

/* col [C*taps][B][Ho][Wo] -> grad_im [B,C,H,W] (accumulated in double, written as float), grad_offset [B,dg*2*taps,Ho,Wo] */
void orc_dcn_backward_input(const float* col, const float* im, const float* offset, int B, int C, int H, int W, int kh,
                            int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                            float* grad_im, float* grad_offset) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int taps = kh * kw, cpdg = C / dg;
  double* gi = (double*)calloc((size_t)B * C * H * W, sizeof(double));
  double* go = (double*)calloc((size_t)B * dg * 2 * taps * Ho * Wo, sizeof(double));
  for (int c = 0; c < C; c++)
    for (int t = 0; t < taps; t++)
      for (int b = 0; b < B; b++)
        for (int ho = 0; ho < Ho; ho++)
          for (int wo = 0; wo < Wo; wo++) {
            int i = t / kw, j = t % kw, g = c / cpdg;
            const float* op = offset + ((size_t)b * dg + g) * 2 * taps * Ho * Wo;
            float oh = op[((size_t)(2 * t) * Ho + ho) * Wo + wo], ow = op[((size_t)(2 * t + 1) * Ho + ho) * Wo + wo];
            float h_im = (ho * stride_h - pad_h) + i * dil_h + oh;
            float w_im = (wo * stride_w - pad_w) + j * dil_w + ow;
            float top = col[((((size_t)c * taps + t) * B + b) * Ho + ho) * Wo + wo];
            const float* imp = im + ((size_t)b * C + c) * H * W;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
              int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
              float lh = h_im - hl, lw = w_im - wl;
              for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                  int y = hl + dy, x = wl + dx;
                  if (y < 0 || y >= H || x < 0 || x >= W) continue;
                  float wgt = (dy ? lh : 1 - lh) * (dx ? lw : 1 - lw);
                  gi[(((size_t)b * C + c) * H + y) * W + x] += (double)wgt * top;
                }
            }
            float ih = h_im, iw = w_im;
            if (ih <= -1 || iw <= -1 || ih >= H || iw >= W) { ih = iw = -2; }
            go[(((size_t)b * dg + g) * 2 * taps + 2 * t) * Ho * Wo + (size_t)ho * Wo + wo] +=
                (double)dcn_coordinate_weight(ih, iw, H, W, imp, W, 0) * top;
            go[(((size_t)b * dg + g) * 2 * taps + 2 * t + 1) * Ho * Wo + (size_t)ho * Wo + wo] +=
                (double)dcn_coordinate_weight(ih, iw, H, W, imp, W, 1) * top;
          }
  for (size_t i = 0; i < (size_t)B * C * H * W; i++) grad_im[i] = (float)gi[i];
  for (size_t i = 0; i < (size_t)B * dg * 2 * taps * Ho * Wo; i++) grad_offset[i] = (float)go[i];
  free(gi); free(go);
}

/* ------------------------------------------------------------------------------------------------------- */
/* a2, DCNv2 (modulated) column kernels, restated per image exactly as the host side drives them                 */
/*   (deform_conv_cuda.cpp:490-685 loops over the batch and calls every kernel with batch_size = 1):            */
/*   modulated_deformable_im2col_gpu_kernel (deform_conv_cuda_kernel.cu:570-633): col = bilinear * mask;         */
/*   modulated_deformable_col2im_gpu_kernel (:635-693): grad_im += gradient_weight * col * mask;                 */
/*   modulated_deformable_col2im_coord_gpu_kernel (:695-767): grad_offset = sum_c coordinate_weight * col * mask, */
/*   grad_mask = sum_c col * bilinear (only where the sample lies inside (-1, H) x (-1, W)).                     */
/*   All sums over channels are carried in double here (the reference: float, thread-serial over channels).      */
/* ------------------------------------------------------------------------------------------------------- */
/* col [C*taps][B][Ho][Wo], mask [B, dg*taps, Ho, Wo] */
void orc_dcn_v2_im2col(const float* im, const float* offset, const float* mask, int B, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                       float* col) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int taps = kh * kw, cpdg = C / dg;
  for (int c = 0; c < C; c++)
    for (int b = 0; b < B; b++) {
      const float* imp = im + ((size_t)b * C + c) * H * W;
      const float* op = offset + ((size_t)b * dg + c / cpdg) * 2 * taps * Ho * Wo;
      const float* mp = mask + ((size_t)b * dg + c / cpdg) * taps * Ho * Wo;
      for (int t = 0; t < taps; t++)
        for (int ho = 0; ho < Ho; ho++)
          for (int wo = 0; wo < Wo; wo++) {
            size_t sp = (size_t)ho * Wo + wo;
            float h_im = (ho * stride_h - pad_h) + (t / kw) * dil_h + op[(size_t)(2 * t) * Ho * Wo + sp];
            float w_im = (wo * stride_w - pad_w) + (t % kw) * dil_w + op[(size_t)(2 * t + 1) * Ho * Wo + sp];
            float val = 0;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = dcn_bilinear(imp, W, H, W, h_im, w_im);
            col[(((size_t)c * taps + t) * B + b) * Ho * Wo + sp] = val * mp[(size_t)t * Ho * Wo + sp];
          }
    }
}

/* col [C*taps][B][Ho][Wo] -> grad_im [B,C,H,W], grad_offset [B,dg*2*taps,Ho,Wo], grad_mask [B,dg*taps,Ho,Wo] */
void orc_dcn_v2_backward_input(const float* col, const float* im, const float* offset, const float* mask, int B, int C,
                               int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                               int dil_h, int dil_w, int dg, float* grad_im, float* grad_offset, float* grad_mask) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int taps = kh * kw, cpdg = C / dg;
  size_t nI = (size_t)B * C * H * W, nO = (size_t)B * dg * 2 * taps * Ho * Wo, nM = nO / 2;
  double* gi = (double*)calloc(nI, sizeof(double));
  double* go = (double*)calloc(nO, sizeof(double));
  double* gm = (double*)calloc(nM, sizeof(double));
  for (int c = 0; c < C; c++)
    for (int t = 0; t < taps; t++)
      for (int b = 0; b < B; b++)
        for (int ho = 0; ho < Ho; ho++)
          for (int wo = 0; wo < Wo; wo++) {
            int g = c / cpdg;
            size_t sp = (size_t)ho * Wo + wo;
            const float* op = offset + ((size_t)b * dg + g) * 2 * taps * Ho * Wo;
            float m = mask[(((size_t)b * dg + g) * taps + t) * Ho * Wo + sp];
            float h_im = (ho * stride_h - pad_h) + (t / kw) * dil_h + op[(size_t)(2 * t) * Ho * Wo + sp];
            float w_im = (wo * stride_w - pad_w) + (t % kw) * dil_w + op[(size_t)(2 * t + 1) * Ho * Wo + sp];
            float top = col[(((size_t)c * taps + t) * B + b) * Ho * Wo + sp];
            const float* imp = im + ((size_t)b * C + c) * H * W;
            int inside = (h_im > -1 && w_im > -1 && h_im < H && w_im < W);
            if (inside) {
              /* (:657) cur_top_grad = col * mask, then the bilinear gradient weights of the four neighbours */
              float topm = top * m;
              int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
              float lh = h_im - hl, lw = w_im - wl;
              for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                  int y = hl + dy, x = wl + dx;
                  if (y < 0 || y >= H || x < 0 || x >= W) continue;
                  float wgt = (dy ? lh : 1 - lh) * (dx ? lw : 1 - lw);
                  gi[(((size_t)b * C + c) * H + y) * W + x] += (double)wgt * topm;
                }
              /* (:748-751) mval += col * bilinear */
              gm[(((size_t)b * dg + g) * taps + t) * Ho * Wo + sp] += (double)top * dcn_bilinear(imp, W, H, W, h_im, w_im);
            }
            float ih = h_im, iw = w_im;
            if (!inside) { ih = iw = -2; }
            /* (:753-756) val += weight * col * mask */
            go[(((size_t)b * dg + g) * 2 * taps + 2 * t) * Ho * Wo + sp] +=
                (double)(dcn_coordinate_weight(ih, iw, H, W, imp, W, 0) * top * m);
            go[(((size_t)b * dg + g) * 2 * taps + 2 * t + 1) * Ho * Wo + sp] +=
                (double)(dcn_coordinate_weight(ih, iw, H, W, imp, W, 1) * top * m);
          }
  for (size_t i = 0; i < nI; i++) grad_im[i] = (float)gi[i];
  for (size_t i = 0; i < nO; i++) grad_offset[i] = (float)go[i];
  for (size_t i = 0; i < nM; i++) grad_mask[i] = (float)gm[i];
  free(gi); free(go); free(gm);
}

/* ------------------------------------------------------------------------------------------------------- */
/* a9: box_iou_rotated (cx,cy,w,h,theta[rad])                                                                */
/*   mmdet/ops/box_iou_rotated/src/box_iou_rotated_utils.h:50-341: vertices (:57-76), edge intersections +      */
/*   contained vertices (:78-156), Graham scan with the CUDA branch's exchange sort (:159-271), fan area          */
/*   (:273-288), IoU on centre-shifted boxes (:314-341).                                                         */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } bpt;
static float b_dot(bpt a, bpt b) { return a.x * b.x + a.y * b.y; }
static float b_cross(bpt a, bpt b) { return a.x * b.y - b.x * a.y; }
static bpt b_sub(bpt a, bpt b) { bpt r = {a.x - b.x, a.y - b.y}; return r; }

static void b_vertices(const float* box, bpt* pts) {
  double theta = box[4];
  float c2 = (float)cos(theta) * 0.5f, s2 = (float)sin(theta) * 0.5f;
  pts[0].x = box[0] - s2 * box[3] - c2 * box[2];
  pts[0].y = box[1] + c2 * box[3] - s2 * box[2];
  pts[1].x = box[0] + s2 * box[3] - c2 * box[2];
  pts[1].y = box[1] - c2 * box[3] - s2 * box[2];
  pts[2].x = 2 * box[0] - pts[0].x; pts[2].y = 2 * box[1] - pts[0].y;
  pts[3].x = 2 * box[0] - pts[1].x; pts[3].y = 2 * box[1] - pts[1].y;
}

static float b_intersection(const float* box1, const float* box2) {
  bpt p1[4], p2[4], v1[4], v2[4], in[24], q[24];
  float dist[24];
  b_vertices(box1, p1); b_vertices(box2, p2);
  for (int i = 0; i < 4; i++) { v1[i] = b_sub(p1[(i + 1) % 4], p1[i]); v2[i] = b_sub(p2[(i + 1) % 4], p2[i]); }
  int num = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float det = b_cross(v2[j], v1[i]);
      if (fabs(det) <= 1e-14) continue;
      bpt v12 = b_sub(p2[j], p1[i]);
      float t1 = b_cross(v2[j], v12) / det, t2 = b_cross(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) { in[num].x = p1[i].x + v1[i].x * t1; in[num].y = p1[i].y + v1[i].y * t1; num++; }
    }
  { bpt AB = v2[0], DA = v2[3]; float abab = b_dot(AB, AB), adad = b_dot(DA, DA);
    for (int i = 0; i < 4; i++) { bpt AP = b_sub(p1[i], p2[0]); float apab = b_dot(AP, AB), apad = -b_dot(AP, DA);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) in[num++] = p1[i]; } }
  { bpt AB = v1[0], DA = v1[3]; float abab = b_dot(AB, AB), adad = b_dot(DA, DA);
    for (int i = 0; i < 4; i++) { bpt AP = b_sub(p2[i], p1[0]); float apab = b_dot(AP, AB), apad = -b_dot(AP, DA);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) in[num++] = p2[i]; } }
  if (num <= 2) return 0.0f;
  /* Graham scan, shift_to_zero = true */
  int t = 0;
  for (int i = 1; i < num; i++) if (in[i].y < in[t].y || (in[i].y == in[t].y && in[i].x < in[t].x)) t = i;
  bpt start = in[t];
  for (int i = 0; i < num; i++) q[i] = b_sub(in[i], start);
  { bpt tmp = q[0]; q[0] = q[t]; q[t] = tmp; }
  for (int i = 0; i < num; i++) dist[i] = b_dot(q[i], q[i]);
  for (int i = 1; i < num - 1; i++)
    for (int j = i + 1; j < num; j++) {
      float cp = b_cross(q[i], q[j]);
      if ((cp < -1e-6) || (fabs(cp) < 1e-6 && dist[i] > dist[j])) {
        bpt qt = q[i]; q[i] = q[j]; q[j] = qt; float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
      }
    }
  int k;
  for (k = 1; k < num; k++) if (dist[k] > 1e-8) break;
  if (k == num) return 0.0f;
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num; i++) {
    while (m > 1 && b_cross(b_sub(q[i], q[m - 2]), b_sub(q[m - 1], q[m - 2])) >= 0) m--;
    q[m++] = q[i];
  }
  if (m <= 2) return 0.0f;
  float area = 0;
  for (int i = 1; i < m - 1; i++) area += (float)fabs(b_cross(b_sub(q[i], q[0]), b_sub(q[i + 1], q[0])));
  return (float)(area / 2.0);
}

void orc_box_iou_rotated(const float* b1, int n, const float* b2, int k, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) {
      const float* r1 = b1 + 5 * (size_t)i; const float* r2 = b2 + 5 * (size_t)j;
      double sx = (r1[0] + r2[0]) / 2.0, sy = (r1[1] + r2[1]) / 2.0;
      float a[5] = {(float)(r1[0] - sx), (float)(r1[1] - sy), r1[2], r1[3], r1[4]};
      float b[5] = {(float)(r2[0] - sx), (float)(r2[1] - sy), r2[2], r2[3], r2[4]};
      float area1 = a[2] * a[3], area2 = b[2] * b[3];
      if (area1 < 1e-14 || area2 < 1e-14) { out[(size_t)i * k + j] = 0.f; continue; }
      float inter = b_intersection(a, b);
      out[(size_t)i * k + j] = inter / (area1 + area2 - inter);
    }
}
