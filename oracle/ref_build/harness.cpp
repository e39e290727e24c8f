// harness.cpp -- TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own kernels, which
// build_ref.py has placed above this text.  Nothing here computes geometry: the pairwise IoU and the
// suppression loop are the reference's code; only the thin host steps around the NMS kernel (which the
// reference does with thrust/cub, csrc/cuda/nms_iou.cu:286-319) are restated with the C++ standard library.

// reference: odtk::cuda::nms / nms_rotate for ONE image (nms.cu:115-157, nms_iou.cu:286-319):
//   discard scores <= 0, stable descending sort by score (cub radix sort is stable), the NMS kernel,
//   stable re-sort with the zeroed scores, first `detections_per_im` entries, zero padding.
// One emulated thread with num_per_thread = num_detections is the kernel's own schedule serialised:
// inside one m-iteration threads only read scores[m] and write scores[i > m], so the barrier version
// and the serial version produce the same memory.
// out_index[k] = position of output k in the input (-1 for zero-score rows).  Returns the number of outputs.
template <int NB, typename Kernel>
static int run_nms(const float *scores, const float *boxes, const float *classes, int count, int detections_per_im,
                   float *out_scores, float *out_boxes, float *out_classes, int *out_index, Kernel kernel) {
  std::vector<int> indices;
  for (int i = 0; i < count; ++i)
    if (scores[i] > 0.0f) indices.push_back(i);
  int num = static_cast<int>(indices.size());
  std::stable_sort(indices.begin(), indices.end(), [&](int a, int b) { return scores[a] > scores[b]; });
  std::vector<float> sorted(num);
  for (int k = 0; k < num; ++k) sorted[k] = scores[indices[k]];

  threadIdx = {0, 0, 0}; blockIdx = {0, 0, 0}; blockDim = {1, 1, 1}; gridDim = {1, 1, 1};
  kernel(num, indices.data(), sorted.data());

  std::vector<int> order(num);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sorted[a] > sorted[b]; });
  const int n_out = std::min(detections_per_im, num);
  for (int k = 0; k < detections_per_im; ++k) {
    out_scores[k] = 0.0f;
    out_classes[k] = 0.0f;
    for (int c = 0; c < NB; ++c) out_boxes[NB * k + c] = 0.0f;
    out_index[k] = -1;
  }
  for (int k = 0; k < n_out; ++k) {
    const int src = indices[order[k]];
    out_scores[k] = sorted[order[k]];
    out_classes[k] = classes[src];
    for (int c = 0; c < NB; ++c) out_boxes[NB * k + c] = boxes[NB * src + c];
    out_index[k] = sorted[order[k]] > 0.0f ? src : -1;
  }
  return n_out;
}

extern "C" {

// reference: odtk::cuda::iou (nms_iou.cu:377-387) launching iou_cuda_kernel (:324-375) -- with the SAME
// argument order as its launch statement at :385, which passes (num_anchors, num_boxes, anchors, boxes):
// inside the kernel the anchors are "b_box_vals" (subject polygon, padded) and the boxes are "a_box_vals"
// (clipper), and the buffer comes out as [num_anchors][num_boxes], the shape csrc/extensions.cpp:64-66 gives it.
// boxes [num_boxes, 4 corners, xy], anchors [num_anchors, 4 corners, xy].
void odtk_ref_iou(const float *boxes, const float *anchors, float *out, int num_boxes, int num_anchors) {
  threadIdx = {0, 0, 0}; blockIdx = {0, 0, 0}; blockDim = {1, 1, 1}; gridDim = {1, 1, 1};
  odtk::cuda::iou_cuda_kernel(num_anchors, num_boxes, reinterpret_cast<const float2 *>(anchors),
                              reinterpret_cast<const float2 *>(boxes), out);
}

int odtk_ref_nms_rotate(const float *scores, const float *boxes, const float *classes, int count, float nms_thresh,
                        int detections_per_im, float *out_scores, float *out_boxes, float *out_classes,
                        int *out_index) {
  return run_nms<6>(scores, boxes, classes, count, detections_per_im, out_scores, out_boxes, out_classes, out_index,
                    [&](int num, const int *indices, float *sorted) {
                      odtk::cuda::nms_rotate_kernel(num, nms_thresh, num, indices, sorted, classes,
                                                    reinterpret_cast<const float6 *>(boxes));
                    });
}

int odtk_ref_nms(const float *scores, const float *boxes, const float *classes, int count, float nms_thresh,
                 int detections_per_im, float *out_scores, float *out_boxes, float *out_classes, int *out_index) {
  return run_nms<4>(scores, boxes, classes, count, detections_per_im, out_scores, out_boxes, out_classes, out_index,
                    [&](int num, const int *indices, float *sorted) {
                      odtk::cuda::nms_kernel(num, nms_thresh, num, indices, sorted, classes,
                                             reinterpret_cast<const float4 *>(boxes));
                    });
}

// reference: the gather + box lambda of odtk::cuda::decode / decode_rotate applied to `count` flat score indices
// of ONE image (decode.cu:119-159, decode_rotate.cu:114-167).  Which indices survive and in which order is the
// caller's business (threshold / top-k are thrust/cub host code); this is the part that turns an index into
// (score, box, class): index decomposition, delta gather layout, anchor arithmetic, clamps, sin/cos passthrough.
void odtk_ref_decode_gather(const int *indices, int count, int rotated, const float *in_scores, const float *in_boxes,
                            int height, int width, int scale, int num_anchors, int num_classes, const float *anchors,
                            float *out_scores, float *out_boxes, float *out_classes) {
  std::vector<float> a(anchors, anchors + 4 * num_anchors);
  for (int k = 0; k < count; ++k) {
    if (rotated) {
      auto t = odtk::cuda::decode_rotate_gather(indices[k], height, width, scale, num_anchors, num_classes, true, a.data(),
                                                in_scores, in_boxes);
      out_scores[k] = t.score; out_classes[k] = static_cast<float>(t.cls);
      const float v[6] = {t.box.x1, t.box.y1, t.box.x2, t.box.y2, t.box.s, t.box.c};
      for (int c = 0; c < 6; ++c) out_boxes[6 * k + c] = v[c];
    } else {
      auto t = odtk::cuda::decode_gather(indices[k], height, width, scale, num_anchors, num_classes, true, a.data(),
                                         in_scores, in_boxes);
      out_scores[k] = t.score; out_classes[k] = static_cast<float>(t.cls);
      const float v[4] = {t.box.x, t.box.y, t.box.z, t.box.w};
      for (int c = 0; c < 4; ++c) out_boxes[4 * k + c] = v[c];
    }
  }
}

}  // extern "C"
