// Hungarian-matcher cost matrices on device: the per-image block of
// HungarianMatcher.forward, dino_detector/matching.py:79-98, for all images in one launch.
//   p          = sigmoid(pred_logits)                                                  matching.py:63
//   neg        = (1 - alpha) * p^gamma * (-log(1 - p + 1e-8))                          :82
//   pos        = alpha * (1 - p)^gamma * (-log(p + 1e-8))                              :83
//   cost_class = pos[:, label] - neg[:, label]                                         :86
//   cost_bbox  = L1 distance of the (cx, cy, w, h) boxes (torch.cdist p=1)             :89
//   cost_giou  = -generalized_box_iou(xyxy(pred), xyxy(gt))                            :92-95, utils.py:124-164
//   C          = w_class * cost_class + w_bbox * cost_bbox + w_giou * cost_giou        :98
// The reference then keeps C[:num_queries] of the matrix it built over ALL B*Q predictions (:102), i.e. the rows of
// image 0 for every image of the batch; `rows_from` selects that behaviour (0) or each image's own rows (-1).
// The assignment itself (scipy linear_sum_assignment, :105) stays on the host.
// One thread per (target g, query q) entry; threads of a wave run along q for a fixed g, so the prediction rows are
// read coalesced-by-row (C+4 floats apart) and the output [Q, n_b] row-major is written with stride n_b: tiny work
// (B*Q*n_gt entries), launch-latency-bound.
#include "dod_common.h"
#include "../../include/dinodet.h"

__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ det, int B, int Q, int C,
                                                         const long long* __restrict__ labels,
                                                         const float* __restrict__ gt_boxes,
                                                         const int* __restrict__ gt_offsets, int G,
                                                         float w_class, float w_bbox, float w_giou, float alpha,
                                                         float gamma, int rows_from, float* __restrict__ cost) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)G * Q) return;
  const int g = (int)(idx / Q), q = (int)(idx - (long)g * Q);
  int lo = 0, hi = B;                       // image of target g: gt_offsets[b] <= g < gt_offsets[b+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gt_offsets[mid] <= g) lo = mid; else hi = mid;
  }
  const int b = lo, off = gt_offsets[b], nb = gt_offsets[b + 1] - off, j = g - off;
  const int src = rows_from >= 0 ? rows_from : b;
  const float* row = det + ((size_t)src * Q + q) * (C + 4);
  const long long lab = labels[g];
  float cc = 0.f;
  if (lab >= 0 && lab < C) {
    const float p = 1.0f / (1.0f + expf(-row[lab]));
    const float pg = gamma == 2.0f ? p * p : powf(p, gamma);
    const float qg = gamma == 2.0f ? (1.0f - p) * (1.0f - p) : powf(1.0f - p, gamma);
    const float neg = (1.0f - alpha) * pg * (-logf(1.0f - p + 1e-8f));
    const float pos = alpha * qg * (-logf(p + 1e-8f));
    cc = pos - neg;
  }
  const float cx = row[C], cy = row[C + 1], w = row[C + 2], h = row[C + 3];
  const float* t = gt_boxes + (size_t)g * 4;
  const float tx = t[0], ty = t[1], tw = t[2], th = t[3];
  const float cb = ((fabsf(cx - tx) + fabsf(cy - ty)) + fabsf(w - tw)) + fabsf(h - th);
  // xyxy
  const float ax1 = cx - 0.5f * w, ay1 = cy - 0.5f * h, ax2 = cx + 0.5f * w, ay2 = cy + 0.5f * h;
  const float bx1 = tx - 0.5f * tw, by1 = ty - 0.5f * th, bx2 = tx + 0.5f * tw, by2 = ty + 0.5f * th;
  const float area1 = (ax2 - ax1) * (ay2 - ay1), area2 = (bx2 - bx1) * (by2 - by1);
  const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = iw * ih;
  const float uni = area1 + area2 - inter;
  const float iou = inter / uni;
  const float ew = fmaxf(fmaxf(ax2, bx2) - fminf(ax1, bx1), 0.f), eh = fmaxf(fmaxf(ay2, by2) - fminf(ay1, by1), 0.f);
  const float earea = ew * eh;
  const float giou = iou - (earea - uni) / earea;
  cost[(size_t)off * Q + (size_t)q * nb + j] = w_class * cc + w_bbox * cb + w_giou * (-giou);
}

extern "C" int dod_match_cost(const float* det, int B, int Q, int C, const int64_t* labels, const float* gt_boxes,
                              const int32_t* gt_offsets, int G, float w_class, float w_bbox, float w_giou, float alpha,
                              float gamma, int rows_from, float* cost, void* stream) {
  if (!det || B <= 0 || Q <= 0 || C <= 0 || G < 0 || !gt_offsets || rows_from >= B) return DOD_ERR_INVALID;
  if (G == 0) return DOD_OK;
  if (!labels || !gt_boxes || !cost) return DOD_ERR_INVALID;
  const long total = (long)G * Q;
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, det, B, Q, C,
                     (const long long*)labels, gt_boxes, gt_offsets, G, w_class, w_bbox, w_giou, alpha, gamma, rows_from, cost);
  return hipGetLastError() == hipSuccess ? DOD_OK : DOD_ERR_HIP;
}
