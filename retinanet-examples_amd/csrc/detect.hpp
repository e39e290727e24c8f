// detect.hpp -- odtk_detect's selection + decode + NMS as ONE launch: select_decode_kernel's body, and in the workgroup that
// wrote an image's LAST candidate list, that image's NMS (nms.hpp: nms_body) -- rotated boxes: the NMS's stage 1 (the first
// round in order, exported for the suppression-matrix launches that follow).
//
// Replaces the launch boundary between reference steps D4-D6 (csrc/cuda/decode.cu:108-167) and N1-N7 (csrc/cuda/nms.cu:115-157)
// that rounds 1-4 kept: `nms_kernel` could not start before the slowest segment of the slowest image had been decoded, and
// then paid a cold start (~2 us dispatch + first-touch latencies).  Now the NMS of image i runs while other images are still
// being selected, in a workgroup that is already resident.
//
// Hand-off (no fence, nobody waits; the same recipe as the tournament's survivor lists, select_decode.hpp "Publish"):
//   writer : every list value is stored write-through (agent-scope atomic store, `sc1`), `s_waitcnt vmcnt(0)`, workgroup
//            barrier, then ONE device-scope ticket per list on the image's counter (SelSeg::lists_done of its level-0 segment,
//            zeroed by the prefilter);
//   reader : the workgroup whose ticket is n_levels - 1 reads every list value with agent-scope loads (nms.hpp: GlobalF32<true>).
// The NMS's own LDS carve-up (NmsLds) starts at the same dynamic-LDS base the selection used: the selection is over by then.
#pragma once

#include "nms.hpp"
#include "select_decode.hpp"

namespace odtk {

// what nms_body needs beyond DecodeArgs (whose outputs ARE its inputs); together with DecodeArgs below 4 KiB of kernel arguments
struct FusedNmsArgs {
  float *out_scores, *out_boxes, *out_classes;   // [batch, ndet], [batch, ndet, NB], [batch, ndet]
  int32_t *out_indices;                          // optional
  // rotated, stage 1 exports (NmsArgs: first_*)
  float *first_box, *first_cls;
  uint32_t *first_n;
  unsigned long long *first_keys;
  uint32_t *first_state;
  uint32_t m_max;
  int ndet;
  float thresh;
  uint32_t flags;
};
static_assert(sizeof(DecodeArgs) + sizeof(FusedNmsArgs) <= 4096, "kernel arguments of detect_kernel must stay below 4 KiB");

// kStage: 0 = the whole (axis-aligned) NMS; 1 = rotated: the first round in order (nms_kernel<6, ., 1>'s work)
template <int NB, typename T, bool kLogits, int kStage>
__global__ __launch_bounds__(kSelThreads) void detect_kernel(const DecodeArgs a, const FusedNmsArgs f) {
  static_assert(kSelThreads == kNmsThreads, "one workgroup runs both bodies");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ uint32_t s_ticket;
  uint32_t b = 0;
  if (!select_decode_body<NB, T, kLogits, kSortCap, true>(a, s_dyn, &b)) return;   // (block-uniform)
  // this workgroup wrote one list of image b (write-through): drain the stores, then the ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = atomicAdd(&a.sel[b].lists_done, 1u);             // (segment of level 0, image b)
  __syncthreads();
  if (s_ticket != static_cast<uint32_t>(a.n_levels) - 1u) return;                   // (block-uniform) another list is still to come
  __syncthreads();
  NmsArgs na{};
  na.scores = a.out_scores;
  na.boxes = a.out_boxes;
  na.classes = a.out_classes;
  na.out_scores = f.out_scores;
  na.out_boxes = f.out_boxes;
  na.out_classes = f.out_classes;
  na.out_indices = f.out_indices;
  na.count = static_cast<uint32_t>(a.n_levels) * static_cast<uint32_t>(a.top_n);
  na.run_len = static_cast<uint32_t>(a.top_n);
  na.run_valid = a.run_valid;
  na.ndet = f.ndet;
  na.thresh = f.thresh;
  na.flags = f.flags;
  na.first_box = f.first_box;
  na.first_cls = f.first_cls;
  na.first_n = f.first_n;
  na.first_keys = f.first_keys;
  na.first_state = f.first_state;
  na.m_max = f.m_max;
  nms_body<NB, false, kStage, true>(na, static_cast<int>(b), a.batch, s_dyn);
}

}  // namespace odtk
