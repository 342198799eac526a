/* CPU restatement of torchvision 0.11.3 `nms` (cpu/nms_kernel.cpp) -- TEST INFRASTRUCTURE ONLY.
 *
 * torchvision is a pinned third-party dependency of the reference
 * (scripts/init_env.sh:25, call site fcos_utils/fcos.py:635) that is neither vendored in
 * /root/reference nor installed in this image, and the reference has no tests for it:
 * PARITY UNPINNED for this function.  The published algorithm is restated literally:
 *   areas = (x2-x1)*(y2-y1); order = argsort(scores, descending);
 *   for i in order: if suppressed[i] continue; keep i;
 *     for later j: w = max(0, min(x2)-max(x1)), h likewise, inter = w*h,
 *                  ovr = inter / (area_i + area_j - inter); if (ovr > thr) suppressed[j] = 1
 * with ovr in fp32 and thr a DOUBLE (the kernel's iou_threshold parameter is double).
 * Ties in the sort are resolved by ascending index (torch's CPU sort is stable).
 * Build: see oracle/Makefile (-ffp-contract=off: no fused multiply-add).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float score; int64_t idx; } item_t;

static int cmp_desc(const void* a, const void* b) {
  const item_t* x = (const item_t*)a; const item_t* y = (const item_t*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* boxes [n][4] (x1,y1,x2,y2), scores [n]; returns number kept, indices in keep[] (score order). */
int64_t oracle_nms_f32(const float* boxes, const float* scores, int64_t n, double thr, int64_t* keep) {
  if (n <= 0) return 0;
  item_t* order = (item_t*)malloc(sizeof(item_t) * (size_t)n);
  float* areas = (float*)malloc(sizeof(float) * (size_t)n);
  unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
  for (int64_t i = 0; i < n; ++i) {
    order[i].score = scores[i]; order[i].idx = i;
    const float* b = boxes + 4 * i;
    const float w = b[2] - b[0], h = b[3] - b[1];
    areas[i] = w * h;
  }
  qsort(order, (size_t)n, sizeof(item_t), cmp_desc);
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; ++_i) {
    const int64_t i = order[_i].idx;
    if (sup[i]) continue;
    keep[nk++] = i;
    const float ix1 = boxes[4*i], iy1 = boxes[4*i+1], ix2 = boxes[4*i+2], iy2 = boxes[4*i+3];
    const float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      const int64_t j = order[_j].idx;
      if (sup[j]) continue;
      const float xx1 = ix1 > boxes[4*j] ? ix1 : boxes[4*j];
      const float yy1 = iy1 > boxes[4*j+1] ? iy1 : boxes[4*j+1];
      const float xx2 = ix2 < boxes[4*j+2] ? ix2 : boxes[4*j+2];
      const float yy2 = iy2 < boxes[4*j+3] ? iy2 : boxes[4*j+3];
      const float dw = xx2 - xx1, dh = yy2 - yy1;
      const float w = dw > 0.f ? dw : 0.f, h = dh > 0.f ? dh : 0.f;
      const float inter = w * h;
      const float denom = iarea + areas[j];
      const float ovr = inter / (denom - inter);
      if ((double)ovr > thr) sup[j] = 1;
    }
  }
  free(order); free(areas); free(sup);
  return nk;
}
