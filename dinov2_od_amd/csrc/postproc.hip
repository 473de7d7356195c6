// Detection post-processing on device: the per-image / per-class / per-query Python loop of
// evaluate_coco (dino_detector/utils.py:195-233) as three small kernels over the packed detections [B, Q, C+4].
//
// Reference semantics, kept exactly:
//   scores = sigmoid(pred_logits)                                   utils.py:198
//   for image i: for class c in 1..C-1 (0 = background, skipped :211-212): for query q in order:
//     keep if scores[i,q,c] > threshold (0.05, :215)
//     box  = cxcywh -> xyxy: x1 = cx - 0.5 w, y1 = cy - 0.5 h, x2 = cx + 0.5 w, y2 = cy + 0.5 h   utils.py:86-87
//     record {image_id, category_id = c, bbox = [x1, y1, x2 - x1, y2 - y1], score}                 utils.py:225-233
// Records come out in the reference's order (image, class, query): the output offset of every (image, class) run is an
// exclusive scan of the per-run counts, and a run is written by ONE thread walking q in order, so no sort and no atomics.
// Lanes run along the class index: for a fixed q the 64 lanes read consecutive floats of one detection row (coalesced).
// HBM-bound integer/compaction work: algorithmic bytes = 2 reads of the logits (count pass + emit pass; the second hits
// L2 for any realistic batch) + 40 B per kept record.
#include "dod_common.h"
#include "../../include/dinodet.h"

// torch.sigmoid in fp32: 1 / (1 + exp(-x))
__device__ __forceinline__ float pp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void pp_count_kernel(const float* __restrict__ det, int B, int Q, int C, float thr,
                                                       unsigned* __restrict__ counts) {
  const int run = blockIdx.x * blockDim.x + threadIdx.x;        // (image, class-1)
  const int nc = C - 1;
  if (run >= B * nc) return;
  const int b = run / nc, c = 1 + run - b * nc;
  const float* p = det + (size_t)b * Q * (C + 4) + c;
  unsigned n = 0;
  for (int q = 0; q < Q; ++q) n += pp_sigmoid(p[(size_t)q * (C + 4)]) > thr ? 1u : 0u;
  counts[run] = n;
}

// exclusive scan of n counts by one workgroup (n <= a few 10^4): offsets[i] = sum_{j<i} counts[j], total[0] = sum
__global__ __launch_bounds__(1024) void pp_scan_kernel(const unsigned* __restrict__ counts, int n,
                                                       unsigned long long* __restrict__ offsets,
                                                       long long* __restrict__ total) {
  __shared__ unsigned long long part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < n ? lo + per : n;
  unsigned long long s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 partial sums
    const unsigned long long v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned long long run = part[tid] - s;       // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
  if (tid == 1023) *total = (long long)part[1023];
}

__global__ __launch_bounds__(256) void pp_emit_kernel(const float* __restrict__ det, int B, int Q, int C, float thr,
                                                      const long long* __restrict__ image_ids,
                                                      const unsigned long long* __restrict__ offsets,
                                                      dod_detection* __restrict__ out, long long max_out) {
  const int run = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = C - 1;
  if (run >= B * nc) return;
  const int b = run / nc, c = 1 + run - b * nc;
  const float* p = det + (size_t)b * Q * (C + 4);
  const long long id = image_ids ? image_ids[b] : (long long)b;
  unsigned long long o = offsets[run];
  for (int q = 0; q < Q; ++q) {
    const float* row = p + (size_t)q * (C + 4);
    const float sc = pp_sigmoid(row[c]);
    if (!(sc > thr)) continue;
    if ((long long)o < max_out) {
      const float cx = row[C], cy = row[C + 1], w = row[C + 2], h = row[C + 3];
      const float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h;      // 0.5 * w is exact: contraction cannot change the result
      const float x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
      dod_detection d;
      d.image_id = id; d.category_id = c; d.query = q;
      d.bbox[0] = x1; d.bbox[1] = y1; d.bbox[2] = x2 - x1; d.bbox[3] = y2 - y1;
      d.score = sc; d.reserved = 0;
      out[o] = d;
    }
    ++o;
  }
}

static inline size_t pp_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t dod_postprocess_workspace_bytes(int B, int Q, int C) {
  (void)Q;
  if (B <= 0 || C <= 1) return 0;
  const size_t runs = (size_t)B * (C - 1);
  return pp_align(runs * sizeof(unsigned)) + pp_align(runs * sizeof(unsigned long long));
}

extern "C" int dod_postprocess(const float* det, int B, int Q, int C, const int64_t* image_ids, float threshold,
                               dod_detection* out, int64_t max_out, int64_t* count, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!det || !count || B <= 0 || Q <= 0 || C <= 1 || max_out < 0 || (max_out > 0 && !out)) return DOD_ERR_INVALID;
  if ((long long)B * (C - 1) > (1 << 24)) return DOD_ERR_INVALID;
  if (!workspace || workspace_bytes < dod_postprocess_workspace_bytes(B, Q, C)) return DOD_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int runs = B * (C - 1);
  unsigned* counts = (unsigned*)workspace;
  unsigned long long* offsets = (unsigned long long*)((char*)workspace + pp_align((size_t)runs * sizeof(unsigned)));
  const int blocks = (runs + 255) / 256;
  hipLaunchKernelGGL(pp_count_kernel, dim3(blocks), dim3(256), 0, s, det, B, Q, C, threshold, counts);
  hipLaunchKernelGGL(pp_scan_kernel, dim3(1), dim3(1024), 0, s, counts, runs, offsets, (long long*)count);
  hipLaunchKernelGGL(pp_emit_kernel, dim3(blocks), dim3(256), 0, s, det, B, Q, C, threshold,
                     (const long long*)image_ids, offsets, out, (long long)max_out);
  return hipGetLastError() == hipSuccess ? DOD_OK : DOD_ERR_HIP;
}
